"""Static check of the COMPILED gfx950 code of libq1env.so for one specific miscompilation (round 6; DESIGN.md section 7, "the fault").

What happened.  The persistent learner (csrc/q1learner_persist.hpp) faulted on the GPU in one source form and ran in an equivalent one.
Caught under rocgdb (profiles/r6_fault_rocgdb.txt), the faulting wave was storing through an address whose per-lane offset had been parked
in accumulation registers for the duration of the step loop - and the parking copies (64 x v_accvgpr_write_b32) had been placed by the
register allocator at the TOP of the join block behind a one-lane `if`, IN FRONT of the `s_or_b64 exec, exec, s[..]` that re-enables the
other lanes:

        s_and_saveexec_b64 s[4:5], s[6:7]          ; if (tid == 0 && g == 0)
        s_cbranch_execz .LBB1_438
        ...                                        ;     a.status[2 + ni] = ...
    .LBB1_438:
        v_accvgpr_write_b32 a38, v102              ; <- live-range split copies: executed by the lanes of the `if` only
        ...                                        ;    (no lane at all in 15 of the 16 workgroups)
        s_or_b64 exec, exec, s[4:5]                ; <- the join's exec restore, which must open the block

A vector instruction only writes the lanes EXEC enables, so every other lane read garbage back after the loop.  It is a code-generation
bug of this LLVM (split copies inserted in front of the block prologue), it depends on the register allocator's decisions and therefore
on any edit anywhere in the kernel, and nothing at run time distinguishes a build that has it from one that has not - until a lane
dereferences the garbage.  So it is checked where it can be seen: in the machine code, for every kernel of every translation unit.

The rule.  The join block of a divergent `if` (the target of the `s_cbranch_execz` that directly follows the `s_and_saveexec_b64`) and the
exit of a divergent loop (behind `s_andn2_b64 exec, exec, m ; s_cbranch_execnz`), when their first EXEC-writing instruction WIDENS the mask
(`s_or_b64 exec, exec, x`, `s_or_saveexec_b64`, `s_mov_b64 exec, x`), must not execute a vector instruction before it (v_readlane / v_writelane / v_readfirstlane ignore EXEC and are what the compiler's own
scalar spills use there: allowed).

build_lib() runs this on the objects it links and refuses to produce a library that violates it; tests/test_isa_exec_restore.py runs it on
every translation unit's assembly and on a synthetic positive (the faulting form itself, -DQ1PL_ONE_ROW_VAR, is kept compilable for that).
"""
import os
import re
import subprocess
import tempfile

LLVM_BIN = os.environ.get("Q1_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

_VECTOR = ("v_", "ds_", "global_", "buffer_", "flat_", "scratch_", "image_", "tbuffer_")
_EXEC_BLIND = ("v_writelane", "v_readlane", "v_readfirstlane")
_LABEL = re.compile(r"^(?:[0-9a-fA-F]+\s+)?<?([.\w$]+)>?:\s*(?://.*|;.*)?$")
_FUNC = re.compile(r"^(?:[0-9a-fA-F]+\s+<([\w$.]+)>:|([A-Za-z_][\w$.]*):\s*(?:;.*)?)$")


def _instr(line):
    """(mnemonic, operand string) of an assembly / disassembly line, or None (blank, comment, directive, label)."""
    s = line.split("//")[0].split(";")[0].strip()
    if not s or s.startswith(".") or s.endswith(":") or s.startswith("<"):
        return None
    parts = s.split(None, 1)
    if not re.match(r"^[a-z][a-z0-9_]*$", parts[0]):
        return None
    return parts[0], (parts[1] if len(parts) > 1 else "")


def _widens_exec(op, args):
    a = [x.strip() for x in args.split(",")]
    if op == "s_or_b64" and len(a) == 3 and a[0] == "exec" and "exec" in a[1:]:
        return True
    if op == "s_or_saveexec_b64":
        return True
    if op == "s_mov_b64" and a and a[0] == "exec":
        return True
    return False


def _writes_exec(op, args):
    a = [x.strip() for x in args.split(",")]
    return bool(a) and (a[0] == "exec" or "saveexec" in op)


def exec_restore_violations(text):
    """[(function, label or '<fallthrough>', line number of the exec restore, [vector instructions in front of it])] - empty = clean.
    `text`: the compiler's assembly (-S: labels `.LBB1_438:`) or llvm-objdump -d --symbolize-operands output (labels `<L8>:`)."""
    lines = text.split("\n")
    # labels are numbered per function in the disassembly: resolve a branch target inside its own function
    func_of, labels, cur = [None] * len(lines), {}, None
    for i, l in enumerate(lines):
        s = l.strip()
        m = _LABEL.match(s)
        if m and _instr(l) is None:
            name = m.group(1)
            if not (name.startswith(".L") or re.match(r"^L\d+$", name)):
                cur = name
            else:
                labels[(cur, name)] = i
        func_of[i] = cur
    out = []

    def scan(start, func, what):
        pend = []
        k = start
        while k < len(lines):
            ins = _instr(lines[k])
            if ins is None:
                k += 1
                continue
            op, args = ins
            if _writes_exec(op, args):
                if _widens_exec(op, args) and pend:
                    out.append((func, what, k + 1, pend))
                return
            if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                return
            if op.startswith(_VECTOR) and not op.startswith(_EXEC_BLIND):
                pend.append(lines[k].split("//")[0].strip())
            k += 1

    def prev_instr(i):
        k = i - 1
        while k >= 0:
            ins = _instr(lines[k])
            if ins is not None:
                return k, ins
            if _LABEL.match(lines[k].strip()):
                return None, None                         # (a label in between: another way into the branch)
            k -= 1
        return None, None

    seen = set()
    for i, l in enumerate(lines):
        ins = _instr(l)
        if ins is None:
            continue
        op, args = ins
        if op == "s_cbranch_execz":
            # the branch around a divergent `if`: s_and[n2]_saveexec_b64 sX, cond [; s_xor_b64 sX, exec, sX] ; s_cbranch_execz JOIN
            # (other s_cbranch_execz - the skip branches INSIDE a divergent region - lead to code that belongs under the narrowed mask)
            k, pi = prev_instr(i)
            if pi is not None and pi[0] == "s_xor_b64":
                k, pi = prev_instr(k)
            if pi is None or "saveexec" not in pi[0]:
                continue
            tgt = args.strip().split()[0]
            j = labels.get((func_of[i], tgt))
            if j is not None and (func_of[i], tgt) not in seen:
                seen.add((func_of[i], tgt))
                scan(j + 1, func_of[i], tgt)
        elif op == "s_cbranch_execnz":
            # the back-edge of a divergent loop: s_andn2_b64 exec, exec, sM ; s_cbranch_execnz LOOP ; <exit: s_or_b64 exec, exec, sM>
            k, pi = prev_instr(i)
            if pi is not None and pi[0] == "s_andn2_b64" and pi[1].replace(" ", "").startswith("exec,exec,"):
                scan(i + 1, func_of[i], "<behind the loop back-edge at line %d>" % (i + 1))
    return out


def device_disassembly(path, workdir=None):
    """gfx950 disassembly (llvm-objdump -d --symbolize-operands) of every device code object embedded in `path` (a host object file or
    shared library produced by hipcc), concatenated."""
    own = workdir is None
    tmp = tempfile.mkdtemp(prefix="q1isa_") if own else workdir
    try:
        local = os.path.join(tmp, os.path.basename(path))
        if os.path.abspath(local) != os.path.abspath(path):
            if os.path.lexists(local):
                os.remove(local)
            os.symlink(os.path.abspath(path), local)
        objdump = os.path.join(LLVM_BIN, "llvm-objdump")
        r = subprocess.run([objdump, "--offloading", os.path.basename(local)], cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("llvm-objdump --offloading failed on " + path + ":\n" + r.stderr)
        texts = []
        for name in sorted(os.listdir(tmp)):
            if name.startswith(os.path.basename(local) + ".") and "amdgcn" in name:
                d = subprocess.run([objdump, "-d", "--symbolize-operands", "--no-show-raw-insn", name], cwd=tmp, capture_output=True, text=True)
                if d.returncode != 0:
                    raise RuntimeError("llvm-objdump -d failed on " + name + ":\n" + d.stderr)
                texts.append(d.stdout)
        if not texts:
            raise RuntimeError("no gfx950 code object found in " + path)
        return "\n".join(texts)
    finally:
        if own:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)


def format_violations(v, limit=6):
    rows = []
    for func, label, line, pend in v[:limit]:
        rows.append(f"  {func}: block {label}: {len(pend)} vector instruction(s) in front of the exec restore at line {line}, first: {pend[0]}")
    if len(v) > limit:
        rows.append(f"  ... and {len(v) - limit} more")
    return "\n".join(rows)


def check_objects(paths):
    """Raise RuntimeError naming every violation found in the device code of `paths` (host objects / shared libraries)."""
    bad = []
    for p in paths:
        v = exec_restore_violations(device_disassembly(p))
        if v:
            bad.append(os.path.basename(p) + ":\n" + format_violations(v))
    if bad:
        raise RuntimeError("miscompiled device code: vector instructions in front of a join block's exec restore (q1physrl_amd/isa_check.py):\n" +
                           "\n".join(bad))


if __name__ == "__main__":
    import sys
    rc = 0
    for p in sys.argv[1:]:
        text = open(p).read() if p.endswith((".s", ".S", ".asm")) else device_disassembly(p)
        v = exec_restore_violations(text)
        print(p, "-", len(v), "violation(s)")
        if v:
            print(format_violations(v, 50))
            rc = 1
    sys.exit(rc)
