"""CPU: properties of the persistent learner's compiled code that its speed rests on (csrc/q1learner_persist.hpp), checked on the compiler's
own assembly (hipcc cross-compiles gfx950 without a GPU) - they depend on per-unit compiler flags (q1physrl_amd/build.py TU_FLAGS) and on
source idioms that a harmless-looking edit can undo:
  * the product kernel keeps everything in registers (no scratch traffic in the step loop);
  * exchanged operands are read by 16-byte device-scope BUFFER loads with the constant part of the address in the instruction's offset field
    (not by per-lane 64-bit pointers hoisted out of the step loop);
  * the barrier polls are plain returning atomics (the atomic optimizer's wave reduction + s_waitcnt vmcnt(0) + readfirstlane would make
    the early poll of barrier 3 synchronous)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    from q1physrl_amd import build
    out = str(tmp_path_factory.mktemp("pl") / "plearner.s")
    src = os.path.join(build.CSRC, "q1env_plearner.hip")
    cmd = [build.hipcc_path()] + [f for f in build.COMPILE_FLAGS if f != "-fPIC"] + build.TU_FLAGS["q1env_plearner.hip"] + \
        ["-I" + os.path.join(ROOT, "include"), "-I" + build.CSRC, "-S", "--cuda-device-only", "-o", out, src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read()


def _kernel_text(asm, prof):
    name = "_ZN4q1pl25persistent_learner_kernelILb%dEEEvNS_4ArgsE" % (1 if prof else 0)
    a = asm.index(name + ":")
    b = asm.index(".Lfunc_end", a)
    return asm[a:b]


def _meta(asm, prof, key):
    name = "_ZN4q1pl25persistent_learner_kernelILb%dEEEvNS_4ArgsE" % (1 if prof else 0)
    i = asm.index(".name:           " + name)
    m = re.search(r"\." + key + r":\s+(\d+)", asm[i:i + 1500])
    return int(m.group(1))


def test_product_kernel_stays_in_registers(asm):
    assert _meta(asm, False, "vgpr_spill_count") <= 4               # (today: two, written in the prologue and read back in the epilogue)
    body = _kernel_text(asm, False)
    loops = [m.start() for m in re.finditer(r"Loop Header: Depth=1", body)]
    assert loops
    assert body.count("scratch_load") + body.count("scratch_store") <= 8     # ... and nothing of the kind inside a step loop:
    for a, b in zip(loops, loops[1:] + [len(body)]):
        span = body[a:b]
        if span.count("v_mfma") >= 40:                           # a step loop (the small loops of prologue / epilogue have no matrix products)
            assert "scratch_" not in span


def test_exchange_loads_are_device_scope_buffer_loads_with_folded_offsets(asm):
    body = _kernel_text(asm, False)
    loads = re.findall(r"buffer_load_dwordx4 [^\n]*", body)
    sc1 = [l for l in loads if l.rstrip().endswith("sc1")]
    assert len(sc1) >= 2 * 60                                   # ~67 exchange loads per step and network specialisation
    folded = [l for l in sc1 if "offset:" in l]
    assert len(folded) >= len(sc1) // 3                         # (s & 3) * 1024 and v * 512 sit in the offset field (a quarter of the pieces has offset 0)
    assert len(set(re.findall(r"buffer_load_dwordx4 \S+ (v\d+),", "\n".join(sc1)))) <= 12     # ... and a handful of address registers serve all of them
    assert not re.search(r"global_load_dwordx4 [^\n]* sc1", body)   # no per-lane 64-bit pointers to exchanged data


def test_barrier_polls_are_plain_returning_atomics(asm):
    body = _kernel_text(asm, False)
    polls = [m.end() for m in re.finditer(r"global_atomic_add v\d+, v\d+, v\d+, s\[\d+:\d+\][^\n]* sc0\n", body)]
    assert len(polls) >= 2 * 5                                  # per network: loop polls of three barriers + the early readings of barriers 1 and 3
    # the optimizer's rewrite broadcasts the (single) returned value to the wave: s_waitcnt vmcnt(0) + v_readfirstlane right behind the atomic
    for e in polls:
        assert "v_readfirstlane_b32" not in "\n".join(body[e:e + 600].split("\n")[:6]), body[e - 80:e + 300]
