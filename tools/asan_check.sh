#!/bin/bash
# AddressSanitizer job for the HOST side of libq1env: builds the library with -fsanitize=address on the host
# code only (-fno-gpu-sanitize: device code unchanged, same kernels), preloads the ASan runtime into python and runs
#   (1) the CPU ABI tests (symbol table, struct layout, loud failure without a device),
#   (2) when a GPU is present: __graft_entry__.smoke() + the staging-heavy compat calls (step_host small/large, reset_at,
#       reset_many, decode_host, phys.apply float32/float64, get/set state) that exercise the pointer arithmetic of the *_host paths.
# Output: gpurun_out/asan/{build.log,cpu.log,gpu.log}; exit status 0 = no ASan report.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
OUT=gpurun_out/asan
mkdir -p $OUT
SO=q1physrl_amd/libq1env_asan.so
# Runtime: GCC's libasan, not ROCm's compiler-rt build - the latter also intercepts hsa_amd_memory_pool_allocate for DEVICE-side
# ASan and aborts every HIP allocation on a GPU that is not in xnack/ASan mode.  The host instrumentation clang emits needs three
# newer helper symbols GCC 11's runtime lacks; tools/asan_shim.c (its own tiny preloaded library) forwards them to libc.
gcc -O1 -fPIC -shared tools/asan_shim.c -o $OUT/libasan_shim.so > $OUT/build.log 2>&1 || { echo "ASAN SHIM BUILD FAILED"; exit 2; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function \
    -mllvm -amdgpu-kernarg-preload-count=16 -fsanitize=address -fno-gpu-sanitize \
    q1physrl_amd/csrc/q1env_*.hip -o $SO >> $OUT/build.log 2>&1 || { echo "ASAN BUILD FAILED"; tail -5 $OUT/build.log | cut -c1-300; exit 2; }
# (libstdc++ rides along so that the runtime finds the real __cxa_throw when it initialises: torch's lazy device initialisation throws
# and catches a C++ exception, and GCC's ASan aborts with a CHECK if it had no libstdc++ to resolve the interceptor's target in)
RT="$(readlink -f "$(gcc -print-file-name=libasan.so)") $(readlink -f "$(gcc -print-file-name=libstdc++.so.6)") $ROOT/$OUT/libasan_shim.so"
# canary: the same toolchain + runtime must catch a deliberate overflow
/opt/rocm/lib/llvm/bin/clang -O1 -g -fPIC -shared -fsanitize=address tools/asan_canary.c -o $OUT/libcanary.so >> $OUT/build.log 2>&1
rm -f $OUT/canary.*
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=$ROOT/$OUT/canary LD_PRELOAD="$RT" python -c "import ctypes; ctypes.CDLL('$ROOT/$OUT/libcanary.so').q1_asan_canary(0)" > /dev/null 2>&1
if ! grep -q "heap-buffer-overflow" $OUT/canary.* 2>/dev/null; then echo "asan: CANARY NOT DETECTED - the sanitizer setup is not live"; exit 3; fi
echo "asan: canary overflow detected (setup is live)"
export Q1ENV_LIB_PATH=$ROOT/$SO
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:log_path=$ROOT/$OUT/report
rm -f $OUT/report.*
LD_PRELOAD="$RT" python -m pytest tests/test_abi_symbols.py -q -x -p no:cacheprovider > $OUT/cpu.log 2>&1
rc1=$?
rc2=0
if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
    # (dlopen goes through the sanitizer's interceptor, so libtorch's RPATH no longer finds its own lazily loaded libraries)
    TORCH_LIB=$(python -c "import importlib.util, os; print(os.path.join(os.path.dirname(importlib.util.find_spec('torch').origin), 'lib'))")
    LD_LIBRARY_PATH="$TORCH_LIB:${LD_LIBRARY_PATH:-}" LD_PRELOAD="$RT" timeout 600 python tools/asan_gpu_calls.py > $OUT/gpu.log 2>&1
    rc2=$?
fi
n=$(ls $OUT/report.* 2>/dev/null | wc -l)
echo "asan: cpu rc=$rc1 gpu rc=$rc2 reports=$n"
tail -3 $OUT/cpu.log; [ -f $OUT/gpu.log ] && tail -5 $OUT/gpu.log
[ "$rc1" = 0 ] && [ "$rc2" = 0 ] && [ "$n" = 0 ]
