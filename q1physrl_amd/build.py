"""Build libq1env.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m q1physrl_amd.build            # or __graft_entry__.build()

hipcc cross-compiles gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with
the repo snapshot.  -ffp-contract=off is part of the numerics contract (the reference never fuses a
multiply-add), not an optimisation knob.
"""
import os
import glob
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "q1env.hip")
DEPS = [SRC] + sorted(glob.glob(os.path.join(PKG, "csrc", "*.hpp"))) + [os.path.join(os.path.dirname(PKG), "include", "q1env.h")]
OUT = os.path.join(PKG, "libq1env.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function",
               "-mllvm", "-amdgpu-kernarg-preload-count=16"]   # leading scalar kernel arguments arrive in SGPRs (step_kernel's state pointers)


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); libq1env.so cannot be built")


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force=False, verbose=False):
    if not force and not is_stale():
        return OUT
    cmd = [hipcc_path()] + HIPCC_FLAGS + [SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(OUT + ".tmp", OUT)
    for stray in glob.glob(OUT + ".*"):            # the offload bundler's per-target intermediates (libq1env.so.0.hipv4-..., ...)
        try:
            os.remove(stray)
        except OSError:
            pass
    return OUT


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
