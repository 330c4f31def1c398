#!/usr/bin/env python
"""tools/isa_check_lds_issue.py -- build-time check of the inline-asm LDS reads (k_*.hip: lds_issue_f64 / lds_issue_i32 / lds_landed).

Those helpers issue `ds_read` from inline assembly so that the ISSUE point holds, and wait for it in a second asm (`s_waitcnt lgkmcnt(0)`).
The compiler does not know that the destination registers are written asynchronously: nothing in the language stops it from copying,
re-using or spilling them between the two statements -- the read would then land in a register that holds something else, silently
(round-4 advisor).  This script compiles the kernels to gfx950 assembly (device only, ~90 s, no GPU needed) and verifies, for EVERY such
read in EVERY kernel:

  * a `s_waitcnt lgkmcnt(0)` asm block follows, and control can neither leave nor enter the code between the two (forward branches
    over exec-masked blocks inside the region are fine: the emissions' exp sits there), and
  * no instruction between the read and that wait names one of the read's destination registers (source or destination).

Exit status 0 = every read verified; 1 = a violation (printed with the kernel's name and the offending lines).

    python tools/isa_check_lds_issue.py                 # compile and check
    python tools/isa_check_lds_issue.py --asm /tmp/k.s  # check an assembly file made with `hipcc -S --cuda-device-only`
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def assemble(path):
    sys.path.insert(0, ROOT)
    from cat_amd.build import assemble as asm   # every kernel family (cat_amd/csrc/k_*.hip), assembled in parallel, one file
    asm(path)


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(lines):
    """-> (number of inline ds_reads verified, list of violation strings)

    The region between a read and its wait may contain forward, exec-masked blocks (s_cbranch_execz over a divergent block): it is
    accepted when every branch inside it targets a label inside it (or the wait itself) and every label inside it is referenced from
    inside it only -- control can then neither leave nor enter the region, and "no instruction of the region names the destination" is
    a statement about every path."""
    bad, nread = [], 0
    # kernel extents and label references
    starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and not l.startswith("\t")]
    kernel_of = {}
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        for i in range(a, b):
            kernel_of[i] = (lines[a].split(":")[0], a, b)
    refs = {}   # (kernel start, label) -> lines that branch to it
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith("s_cbranch") or t.startswith("s_branch"):
            k = kernel_of.get(i, ("?", 0, 0))[1]
            refs.setdefault((k, t.split()[-1]), []).append(i)
    i, n = 0, len(lines)
    while i < n:
        l = lines[i]
        if "#ASMSTART" in l and i + 1 < n and re.match(r"\s*ds_read_b(32|64|128)\s", lines[i + 1]):
            kernel, ks, ke = kernel_of.get(i, ("?", 0, n))
            ins = lines[i + 1].strip()
            dst = regs(ins.split(None, 1)[1].split(",")[0])     # first operand = destination
            nread += 1
            j, wait = i + 2, -1
            while j < ke:
                if "#ASMSTART" in lines[j] and j + 1 < n and "s_waitcnt" in lines[j + 1] and "lgkmcnt(0)" in lines[j + 1]:
                    wait = j
                    break
                j += 1
            if wait < 0:
                bad.append(f"{kernel}: `{ins}` (line {i + 2}): no `s_waitcnt lgkmcnt(0)` asm behind it")
                i += 1
                continue
            labels_in = {}
            for j in range(i + 2, wait):
                m = re.match(r"(\.LBB\d+_\d+):", lines[j].strip())
                if m:
                    labels_in[m.group(1)] = j
            ok = True
            for j in range(i + 2, wait):
                t = lines[j].strip()
                if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
                    continue
                if t.startswith("s_cbranch") or t.startswith("s_branch"):
                    tgt = t.split()[-1]
                    if tgt not in labels_in:   # (a jump backwards to a label INSIDE the region is layout only: blocks placed out of line)
                        bad.append(f"{kernel}: `{ins}` (line {i + 2}): `{t}` (line {j + 1}) leaves the region before the wait (line {wait + 2})")
                        ok = False
                        break
                    continue
                if "s_endpgm" in t or t.startswith("s_setpc") or t.startswith("s_swappc"):
                    bad.append(f"{kernel}: `{ins}` (line {i + 2}): `{t}` before the wait")
                    ok = False
                    break
                if re.match(r"\.LBB\d+_\d+:", t):
                    continue
                t = t.split(";")[0].strip()
                ops = t.split(None, 1)[1] if " " in t else ""
                if regs(ops) & dst:
                    bad.append(f"{kernel}: `{ins}` (line {i + 2}): destination touched before its wait by `{t}` (line {j + 1})")
                    ok = False
                    break
            if ok:
                for lab, at in labels_in.items():
                    outside = [r for r in refs.get((ks, lab), []) if r < i or r > wait]
                    if outside:
                        bad.append(f"{kernel}: `{ins}` (line {i + 2}): label {lab} inside the region is entered from line {outside[0] + 1}")
                        break
        i += 1
    return nread, bad


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--asm", help="assembly file to read instead of compiling")
    ap.add_argument("--keep", help="write the assembly here")
    a = ap.parse_args()
    path = a.asm or a.keep or os.path.join(tempfile.mkdtemp(prefix="isa_"), "crf.s")
    if not a.asm:
        assemble(path)
    nread, bad = check(open(path).read().split("\n"))
    for b in bad:
        print("VIOLATION", b)
    print(f"{nread} inline LDS reads checked, {len(bad)} violations")
    return 1 if bad or nread == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
