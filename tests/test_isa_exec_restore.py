"""CPU: the machine code of every kernel the library ships is held to q1physrl_amd/isa_check.py's rule - no vector instruction in front of the
exec restore that opens a divergent `if`'s join block (or a divergent loop's exit).  That is the code-generation bug behind the persistent
learner's GPU memory fault of round 5 (profiles/r6_fault_rocgdb.txt): register-allocator copies placed where only the `if`'s lanes execute
them.  It depends on the allocator's decisions, i.e. on any edit of a kernel, and a build that has it looks healthy until a lane uses the
garbage - so build_lib() refuses to link such a library and this suite checks the rule itself and the libraries build() produced."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from q1physrl_amd import isa_check  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "r6_faulting_join_block.s")


def test_the_faulting_block_of_round_5_is_flagged():
    v = isa_check.exec_restore_violations(open(GOLDEN).read())
    assert len(v) == 1
    func, label, _line, pend = v[0]
    assert "persistent_learner_kernel" in func and label == ".LBB1_438"
    assert len(pend) == 64 and all(p.startswith("v_accvgpr_write_b32") for p in pend)       # (the v_writelane scalar spills between them are exec-blind)


CLEAN_IF = """
kern:
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mov_b32_e32 v1, 0
	global_store_dword v1, v1, s[0:1]
.LBB0_2:
	v_writelane_b32 v255, s30, 8
	s_mov_b32 s48, s84
	s_or_b64 exec, exec, s[4:5]
	v_accvgpr_write_b32 a0, v1
	s_endpgm
"""


def test_rule_on_synthetic_blocks():
    assert isa_check.exec_restore_violations(CLEAN_IF) == []
    # the same join with one copy moved in front of the restore
    bad = CLEAN_IF.replace("\ts_or_b64 exec, exec, s[4:5]\n\tv_accvgpr_write_b32 a0, v1\n", "\tv_accvgpr_write_b32 a0, v1\n\ts_or_b64 exec, exec, s[4:5]\n")
    v = isa_check.exec_restore_violations(bad)
    assert len(v) == 1 and v[0][1] == ".LBB0_2" and v[0][3] == ["v_accvgpr_write_b32 a0, v1"]
    # if / else: the flow block opens with s_or_saveexec
    els = """
kern:
	s_and_saveexec_b64 s[2:3], vcc
	s_xor_b64 s[2:3], exec, s[2:3]
	s_cbranch_execz .LBB0_2
	v_mov_b32_e32 v0, 1
.LBB0_2:
	v_mov_b32_e32 v9, v8
	s_or_saveexec_b64 s[4:5], s[2:3]
	s_xor_b64 exec, exec, s[4:5]
	s_endpgm
"""
    v = isa_check.exec_restore_violations(els)
    assert len(v) == 1 and v[0][3] == ["v_mov_b32_e32 v9, v8"]
    # divergent loop exit
    loop = """
kern:
.LBB0_1:
	v_add_u32_e32 v0, 1, v0
	s_or_b64 s[6:7], vcc, s[6:7]
	s_andn2_b64 exec, exec, s[6:7]
	s_cbranch_execnz .LBB0_1
; %bb.2:
	ds_read_b32 v3, v2
	s_or_b64 exec, exec, s[6:7]
	s_endpgm
"""
    v = isa_check.exec_restore_violations(loop)
    assert len(v) == 1 and v[0][3] == ["ds_read_b32 v3, v2"]
    assert isa_check.exec_restore_violations(loop.replace("\tds_read_b32 v3, v2\n\ts_or_b64 exec, exec, s[6:7]\n", "\ts_or_b64 exec, exec, s[6:7]\n\tds_read_b32 v3, v2\n")) == []
    # a skip branch INSIDE a divergent region (not preceded by the saveexec) leads to code that belongs under the narrowed mask: not a join
    inner = """
kern:
	s_and_saveexec_b64 s[2:3], s[10:11]
	s_cbranch_execz .LBB0_5
; %bb.1:
	s_and_b64 vcc, exec, s[42:43]
	s_cbranch_vccz .LBB0_3
; %bb.2:
	global_load_dword v34, v131, s[72:73] sc1
	s_cbranch_execz .LBB0_4
	s_branch .LBB0_5
.LBB0_3:
.LBB0_4:
	v_mov_b32_e32 v34, v131
	global_atomic_add v34, v131, v34, s[72:73] sc0
.LBB0_5:
	s_or_b64 exec, exec, s[2:3]
	s_endpgm
"""
    assert isa_check.exec_restore_violations(inner) == []
    # the disassembler's form: labels <L8>, numbered per function
    dis = """
0000000000001e00 <kern_a>:
	s_and_saveexec_b64 s[0:1], vcc                             // 00000000211C: BE80206A
	s_cbranch_execz L1                                         // 000000002120: BF880155
	v_mov_b32_e32 v1, 0                                        // 000000002124: 7E020280
0000000000002678 <L1>:
	v_accvgpr_write_b32 a3, v1                                 // 000000002678: D3D94003 18000101
	s_or_b64 exec, exec, s[0:1]                                // 000000002680: 87FE007E
0000000000003000 <kern_b>:
	s_and_saveexec_b64 s[0:1], vcc                             // 000000003000: BE80206A
	s_cbranch_execz L1                                         // 000000003004: BF880155
0000000000003010 <L1>:
	s_or_b64 exec, exec, s[0:1]                                // 000000003010: 87FE007E
	v_accvgpr_write_b32 a3, v1                                 // 000000003018: D3D94003 18000101
"""
    v = isa_check.exec_restore_violations(dis)
    assert [(x[0], x[1]) for x in v] == [("kern_a", "L1")]


@pytest.mark.parametrize("name", ["libq1env.so", "libq1env_check.so"])
def test_shipped_libraries_are_clean(name):
    """Every kernel of every translation unit, as linked (the product library and the -DQ1_CHECK assertion build of __graft_entry__.build())."""
    so = os.path.join(ROOT, "q1physrl_amd", name)
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    text = isa_check.device_disassembly(so)
    assert text.count("s_and_saveexec_b64") > 200          # (the disassembly is what it claims to be: thousands of divergent regions were looked at)
    v = isa_check.exec_restore_violations(text)
    assert v == [], isa_check.format_violations(v)


def test_build_refuses_a_violating_library(tmp_path, monkeypatch):
    """build_lib() runs the check on what it links, before the library replaces the previous one."""
    from q1physrl_amd import build
    monkeypatch.setattr(isa_check, "device_disassembly", lambda path, workdir=None: open(GOLDEN).read())
    with pytest.raises(RuntimeError, match="exec restore"):
        isa_check.check_objects([os.path.join(build.PKG, "libq1env.so")])
