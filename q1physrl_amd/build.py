"""Build libq1env.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m q1physrl_amd.build [--force] [--check]      # or __graft_entry__.build()

The library is several translation units (csrc/q1env_*.hip: env core, policy glue, tick server, resident sampler, learner,
diagnostics), each compiled to an object file (in parallel; only the stale ones) and linked into ONE shared library.  hipcc
cross-compiles gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with the repo snapshot.
-ffp-contract=off is part of the numerics contract (the reference never fuses a multiply-add), not an optimisation knob.
-fvisibility=hidden: only the functions include/q1env.h declares are exported (tests/test_abi_symbols.py).

--check builds a second library, libq1env_check.so, with -DQ1_CHECK: device-side assertions of the hand-rolled hand-off protocols
(the inline-assembly 16-byte sc1 granule stores are read back and compared with the registers they were issued from); used by the
soak tools through Q1ENV_LIB_PATH, never by the product path.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SOURCES = sorted(glob.glob(os.path.join(CSRC, "q1env_*.hip")))
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(os.path.dirname(PKG), "include", "q1env.h")]
DEPS = SOURCES + HEADERS
OBJ_DIR = os.path.join(PKG, "build")
OUT = os.path.join(PKG, "libq1env.so")
OUT_CHECK = os.path.join(PKG, "libq1env_check.so")

COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
                 "-Wall", "-Wno-unused-function",
                 "-mllvm", "-amdgpu-kernarg-preload-count=16"]   # leading scalar kernel arguments arrive in SGPRs (step_kernel's state pointers)
# per translation unit, on top of COMPILE_FLAGS.  q1env_learner.hip: matrix instructions take their C / D operands in the ordinary
# vector registers wherever the allocator can afford it (round 4): the learner's backward kernel spent 19 % of its vector instructions
# moving values between the two register files - transposition results that the very next instruction converts (558 -> 100 moves per
# tile, 2 933 -> 2 501 vector instructions; the 32 768-sample SGD step 106 -> 101 us).  The policy / sampler units keep the compiler's
# default (the resident sampler measured 3 % slower with it).
TU_FLAGS = {"q1env_learner.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
            # the persistent learner: the same register form (its step loop read 220 accumulator registers back per step), and NO atomic
            # optimizer: that pass rewrites a uniform-address atomic into a wave reduction + s_waitcnt vmcnt(0) + readfirstlane, i.e. it
            # makes the barrier poll that is requested early and looked at late (q1learner_persist.hpp, barrier 3) synchronous again
            "q1env_plearner.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]
HIPCC_FLAGS = COMPILE_FLAGS + ["-shared"]     # (kept for tools that compile a single file the old way, e.g. tools/asan_check.sh)


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); libq1env.so cannot be built")


def sources_sha16(flags=None):
    """16 hex digits over everything the library is built from: csrc/*.hip, csrc/*.hpp, include/q1env.h (names + contents, sorted)
    and the compile flags.  Compiled into the library (q1env_build_id) and recorded by the profiling tools next to the counters they
    take, so that bench.py can tell whether profiles/pmc.json describes the library it is running."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(DEPS):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(COMPILE_FLAGS if flags is None else flags).encode())
    h.update(repr(sorted(TU_FLAGS.items())).encode())
    return h.hexdigest()[:16]


BUILD_ID_TU = "q1env_core.hip"        # the translation unit that defines q1env_build_id()


def _obj_of(src, tag):
    return os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + tag + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def is_stale(out=OUT):
    return _stale(out, DEPS)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)


def build_lib(force=False, verbose=False, check=False, extra_flags=(), out=None, tag=None, allow_miscompiled=False):
    """Compile the stale translation units (in parallel) and link.  check=True: the -DQ1_CHECK assertion build (libq1env_check.so).
    out / tag: an experimental variant (extra_flags) built next to the product library, e.g. tools/exp_step_large.py.
    Every library is disassembled and held to q1physrl_amd/isa_check.py's rule before it replaces the previous one (the compiler bug that made
    the persistent learner fault in round 5 is visible in the machine code and nowhere else); allow_miscompiled=True only for the diagnostic
    variant that reproduces it (tools/r6_fault.sh)."""
    if out is None:
        out = OUT_CHECK if check else OUT
    if tag is None:
        tag = "_check" if check else ""
    if out == OUT:
        build_rows_helper(force=force, verbose=verbose)       # the host-side CPython helper rides with the product build
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = COMPILE_FLAGS + (["-DQ1_CHECK=1"] if check else []) + list(extra_flags)
    # the build id (hash of all sources + flags) is compiled into ONE translation unit; its stamp file makes that unit - and the
    # link - stale whenever any source changed, even one that unit does not include
    bid = sources_sha16(flags)
    stamp = os.path.join(OBJ_DIR, "build_id" + tag + ".txt")
    old = open(stamp).read().strip() if os.path.exists(stamp) else None
    if old != bid:
        with open(stamp, "w") as f:
            f.write(bid + "\n")
    if not force and not _stale(out, DEPS + [stamp]):
        return out
    hipcc = hipcc_path()
    jobs = []
    for src in SOURCES:
        obj = _obj_of(src, tag)
        is_id_tu = os.path.basename(src) == BUILD_ID_TU
        if force or _stale(obj, [src] + HEADERS + ([stamp] if is_id_tu else [])):
            jobs.append([hipcc] + flags + TU_FLAGS.get(os.path.basename(src), []) + (['-DQ1_BUILD_ID="' + bid + '"'] if is_id_tu else []) +
                        ["-c", src, "-o", obj])
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    _run([hipcc] + LINK_FLAGS + [_obj_of(s, tag) for s in SOURCES] + ["-o", out + ".tmp"], verbose)
    if not allow_miscompiled:
        from . import isa_check
        try:
            isa_check.check_objects([out + ".tmp"])
        except RuntimeError:
            for stray in glob.glob(out + ".tmp*"):
                os.remove(stray)
            raise
    os.replace(out + ".tmp", out)
    for stray in glob.glob(out + ".*"):            # the offload bundler's per-target intermediates (libq1env.so.0.hipv4-..., ...)
        try:
            os.remove(stray)
        except OSError:
            pass
    return out


ROWS_SRC = os.path.join(CSRC, "q1rows.c")
ROWS_OUT = os.path.join(PKG, "_q1rows.so")


def build_rows_helper(force=False, verbose=False):
    """The CPython helper of the drop-in surface (csrc/q1rows.c: RLlib's list-of-tuples actions -> float64 rows in one C pass), gcc.
    Optional: env.py falls back to its NumPy formulations without it, so a missing compiler or header only warns."""
    if not force and os.path.exists(ROWS_OUT) and os.path.getmtime(ROWS_OUT) >= os.path.getmtime(ROWS_SRC):
        return ROWS_OUT
    import sysconfig
    try:
        import numpy
        cmd = ["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(), ROWS_SRC,
               "-o", ROWS_OUT + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        os.replace(ROWS_OUT + ".tmp", ROWS_OUT)
        return ROWS_OUT
    except Exception as ex:   # noqa: BLE001 - no compiler / no Python headers: the NumPy path stays
        import warnings
        warnings.warn(f"q1physrl_amd: the action-row helper was not built ({ex}); list-of-tuples actions take the NumPy path", RuntimeWarning)
        return None


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True, check="--check" in sys.argv))
