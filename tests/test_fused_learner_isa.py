"""CPU: properties of the fused large-minibatch learner step's compiled code that its speed rests on (csrc/q1learner_fused.hpp, the shared-operand
weight-gradient kernel of csrc/q1learner.hpp), checked on the compiler's own assembly (hipcc cross-compiles gfx950 without a GPU):
  * the fused forward + backward kernel fits TWO waves per SIMD (eight waves = eight sample tiles per workgroup) without spilling - the phases are
    fully unrolled and sit within a few registers of the 256 a wave may have: a branch around a store (`if (half == 0)`) or one more live vector
    costs the allocator 80 - 160 registers there (round 6 measured 98 and 163 spilled registers that way);
  * no scratch traffic inside its matrix-instruction phases;
  * the saturation report is one pair of atomics per WORKGROUP - per wave it was 2 x 2 048 same-address atomics per launch, which kept the dispatch
    open 14 us after its last wave had ended;
  * the wave's own h1 vectors are requested before the first store of the backward phase (behind the dZ2 stores they wait for those stores'
    acknowledgements: in-order completion);
  * the weight-gradient kernel's barriers wait for LDS only (a drained vector-memory queue per tile would undo its three-tile prefetch)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FZ = "_ZN7q1learn21learner_fwdbwd_kernelILb%dEEEviPKfPKlS4_NS_5FzNetES5_NS_8LossArgsENS_6BcArgsE"
WGS = "_ZN7q1learn27learner_wgrad_shared_kernelEiNS_5WgNetES0_i"
BWD = "_ZN7q1learn23learner_backward_kernelILb1EEEviPKfPKlS4_NS_6BwdNetES5_iNS_8LossArgsENS_6BcArgsE"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    from q1physrl_amd import build
    out = str(tmp_path_factory.mktemp("fz") / "learner.s")
    src = os.path.join(build.CSRC, "q1env_learner.hip")
    cmd = [build.hipcc_path()] + [f for f in build.COMPILE_FLAGS if f != "-fPIC"] + build.TU_FLAGS["q1env_learner.hip"] + \
        ["-I" + os.path.join(ROOT, "include"), "-I" + build.CSRC, "-S", "--cuda-device-only", "-o", out, src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read()


def _body(asm, name):
    a = asm.index(name + ":")
    return asm[a:asm.index(".Lfunc_end", a)]


def _meta(asm, name, key):
    i = asm.index(".name:           " + name)
    j = asm.rfind("  - .agpr_count", 0, i)                     # a kernel's metadata block: "- .agpr_count", its arguments, .name, the register counts
    k = asm.find("  - .agpr_count", i)
    block = asm[j:k if k >= 0 else i + 2000]
    m = re.search(r"\." + key + r":\s+(\d+)", block)
    assert m, key
    return int(m.group(1))


@pytest.mark.parametrize("dw1", [0, 1])
def test_fused_kernel_fits_two_waves_per_simd_without_spilling(asm, dw1):
    name = FZ % dw1
    assert _meta(asm, name, "vgpr_count") <= 256 and _meta(asm, name, "agpr_count") == 0
    assert _meta(asm, name, "vgpr_spill_count") <= 2            # (today: 0 / 1 - the one written before the forward phase and read back after it)
    body = _body(asm, name)
    mf = 0
    for line in body.split("\n"):
        if "v_mfma" in line:
            mf += 1
        # scratch traffic only outside the matrix-instruction phases of the backward half (matrix instructions 40 .. of ~225 / ~260)
        assert not ("scratch_" in line and mf > 40), line
    assert mf >= 200


@pytest.mark.parametrize("name", [FZ % 0, FZ % 1, BWD])
def test_saturation_report_is_per_workgroup(asm, name):
    body = _body(asm, name)
    atomics = re.findall(r"global_atomic_\w+", body)
    assert sorted(set(atomics)) == ["global_atomic_add", "global_atomic_umax"] and len(atomics) == 2, atomics
    # ... issued behind a workgroup barrier by one thread, i.e. after the per-wave values met in LDS
    last_barrier = max(m.start() for m in re.finditer(r"\ts_barrier", body))
    assert all(m.start() > last_barrier for m in re.finditer(r"global_atomic_", body))


def test_own_h1_vectors_are_requested_before_the_backward_phase_stores(asm):
    body = _body(asm, FZ % 1)
    barriers = [m.start() for m in re.finditer(r"\ts_barrier", body)]
    assert len(barriers) == 4                                   # forward image staged | forward done | backward images staged | statistics rows
    phase_b = body[barriers[2]:barriers[3]]
    first_store = phase_b.index("global_store_dword")
    loads_before = len(re.findall(r"global_load_dwordx4", phase_b[:first_store]))
    assert loads_before >= 16, loads_before                     # the sixteen 16-byte vectors of tanh(H1)
    assert len(re.findall(r"global_load_dwordx4", phase_b[first_store:])) == 0


def test_weight_gradient_kernel_barriers_wait_for_lds_only(asm):
    body = _body(asm, WGS)
    lines = body.split("\n")
    n = 0
    for i, line in enumerate(lines):
        if line.strip().startswith("s_barrier"):
            n += 1
            before = " ".join(x.strip() for x in lines[max(0, i - 4):i])
            assert "vmcnt(0)" not in before, before
    assert n >= 3                                               # one per sample tile of the unrolled ring
    assert _meta(asm, WGS, "vgpr_spill_count") == 0
