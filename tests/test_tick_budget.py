"""CPU: the headline kernel's tick stays within its instruction budget (<= 255 VALU instructions per tick per wave).
With one wave per SIMD at 65 536 envs a tick costs one issue slot per instruction, so the instruction count of the
compiler's own assembly IS the performance model; tools/count_tick_insts.py cross-compiles csrc/q1env_core.hip for gfx950 (no GPU
needed), finds rollout_kernel<float, true, FMT_PACKED, false, 1, false>'s tick loop and counts the blocks a tick executes.  The
hardware counterpart, SQ_INSTS_VALU = 236.7 per tick per wave, is in profiles/r4_summary.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_headline_tick_is_within_its_instruction_budget(capsys):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import count_tick_insts
    hot = count_tick_insts.main([])
    out = capsys.readouterr().out
    assert hot is not None, out
    assert 4 <= len(hot["blocks"]) <= 6, hot                    # header, physics, friction, output block (+ at most two the compiler may split off)
    assert 200 <= hot["valu"] <= 240, hot                       # 237 since round 4 (294 in round 3, 337 before); the floor argument per block: profiles/r5_tick_floor.txt
    assert hot["slots"] <= 272, hot                             # 268: VALU + SALU + LDS + VMEM + s_nop + s_waitcnt + branches (round 5: write-through stores, one wait less)
    assert "Occupancy: 4" in out and "ScratchSize: 0" in out    # <= 128 VGPRs (four waves per SIMD for the 262 144-env configs), no spills


def test_two_ahead_instantiation_costs_the_same_tick(capsys):
    """rollout_kernel<..., DEPTH = 2> (launches of >= 32 ticks: the action is requested two ticks ahead, the loop body is two ticks): per
    tick within 3 VALU instructions / 3 slots of the one-ahead form, still four waves per SIMD."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import count_tick_insts
    hot = count_tick_insts.main(["--kernel", "rollout_kernelIfLb1ELi2ELb0ELi1ELb0ELi2EE"])
    out = capsys.readouterr().out
    assert hot is not None, out
    assert 400 <= hot["valu"] <= 2 * 243 and hot["slots"] <= 2 * 275, hot
    assert "Occupancy: 4" in out
