#!/usr/bin/env python3
"""Static instruction count of a kernel's tick loop from the compiler's own assembly (no GPU needed).

    python tools/count_tick_insts.py [--kernel SUBSTR] [--src csrc/q1env_core.hip] [--keep out.s] [-D...]

Compiles the translation unit for gfx950 (device only, the product flags of q1physrl_amd/build.py), takes the FIRST kernel whose
mangled name contains SUBSTR (default: rollout_kernel<float, true, FMT_PACKED, false, 1> = the bench's headline kernel), finds its
innermost loop(s) and prints, per basic block of the loop, the number of VALU / SALU / LDS / VMEM / s_nop / s_waitcnt / branch
instructions.  A lone wave on its SIMD pays one issue slot per instruction whatever its type (tools/ubench_f64.hip), so the sum over
the blocks a tick executes is the tick's cost; rocprofv3's SQ_INSTS_VALU is the measured counterpart (profiles/r4_summary.txt)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(op):
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="rollout_kernelIfLb1ELi2ELb0ELi1ELb0ELi1EE")
    ap.add_argument("--src", default=os.path.join(ROOT, "q1physrl_amd", "csrc", "q1env_core.hip"))
    ap.add_argument("--keep", default=None)
    ap.add_argument("--asm", default=None, help="use this assembly file instead of compiling")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args(argv)
    from q1physrl_amd import build
    if a.asm:
        text = open(a.asm).read()
    else:
        out = a.keep or os.path.join(tempfile.mkdtemp(), "tu.s")
        cmd = [build.hipcc_path()] + build.COMPILE_FLAGS + ["-D" + d for d in a.D] + ["--cuda-device-only", "-S", "-o", out, a.src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
        text = open(out).read()
    lines = text.splitlines()
    start = next((i for i, ln in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(a.kernel) + r"\S*:", ln)), None)
    if start is None:
        sys.exit(f"kernel containing {a.kernel!r} not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    print(lines[start].split(":")[0])
    for ln in lines[end:end + 80]:
        if re.search(r"NumVgprs|NumSgprs|Occupancy|ScratchSize", ln):
            print("   ", ln.strip("; ").strip())
    # basic blocks
    blocks, cur, name, note, targets = [], [], "entry", "", []
    headers = set(re.findall(r"^(\.LBB\d+_\d+):\s*;.*Loop Header", "\n".join(body), flags=re.M))

    def close():
        blocks.append((name, note, cur, any(t in headers for t in targets)))
    for ln in body[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
        if m:
            close()
            name, note, cur, targets = m.group(1), (m.group(2) or ""), [], []
            continue
        m2 = re.match(r"^; %bb\.(\d+):\s*(;.*)?$", ln)
        if m2:
            close()
            name, note, cur, targets = "%bb." + m2.group(1), (m2.group(2) or ""), [], []
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        cur.append(t.split()[0])
        if t.startswith(("s_cbranch", "s_branch")):
            targets.append(t.split()[-1])
    close()
    hot = hot_path(blocks)
    total = collections.Counter()
    for name, note, ops, _back in blocks:
        if "in Loop" not in note and "Inner Loop Header" not in note and "Loop Header" not in note:
            continue
        c = collections.Counter(classify(o) for o in ops)
        total.update(c)
        print(f"  {name:12s} {sum(c.values()):4d}  " + "  ".join(f"{k}={c[k]}" for k in ("valu", "salu", "lds", "vmem", "nop", "wait", "branch") if c[k])
              + "   " + note.strip("; ").strip()[:60])
    print("  all loop blocks:", dict(total), "sum", sum(total.values()))
    if hot:
        print(f"  hot path of the tick loop ({' '.join(hot['blocks'])}): VALU {hot['valu']}  all issue slots {hot['slots']}")
    return hot


def hot_path(blocks):
    """The blocks one tick executes: from the header of the LARGEST loop (by instruction count) to the block that holds the back edge,
    in layout order - the compiler places the blocks a tick normally runs through contiguously and the cold ones (the library sin / cos
    fallback for |yaw| >= 2^20 rad) behind the latch.  Returns {'blocks', 'valu', 'slots'} or None."""
    loops = {}
    for idx, (name, note, ops, _b) in enumerate(blocks):
        m = re.search(r"Loop Header", note)
        if m and name.startswith(".LBB"):
            loops[name] = idx
    best = None
    for header, hidx in loops.items():
        hname = header.replace(".LBB", "BB")
        members = [i for i, (n, note, ops, _b) in enumerate(blocks) if i == hidx or ("Header=" + hname + " ") in (note + " ")]
        if not members:
            continue
        size = sum(len(blocks[i][2]) for i in members)
        if best is None or size > best[0]:
            best = (size, header, hidx, members)
    if best is None:
        return None
    _, header, hidx, members = best
    # layout order from the header to the first member that branches back to a loop header (the latch)
    run = [hidx]
    i = hidx + 1
    while i < len(blocks) and i in members and len(run) < 16:
        run.append(i)
        if blocks[i][3]:                       # this block branches back to the header
            break
        i += 1
    valu = sum(sum(1 for o in blocks[j][2] if classify(o) == "valu") for j in run)
    slots = sum(len(blocks[j][2]) for j in run)
    return {"blocks": [blocks[j][0] for j in run], "valu": valu, "slots": slots}


if __name__ == "__main__":
    main()
