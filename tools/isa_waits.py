#!/usr/bin/env python
"""tools/isa_waits.py -- where does a kernel WAIT?  Compiles cat_amd/csrc/k_*.hip to gfx950 assembly (device only, a minute, no GPU
needed) and prints, for every loop of the kernels whose mangled name contains KEY, the sequence of memory instructions, barriers and
s_waitcnt vmcnt(...) in layout order -- runs of the same instruction folded.  What to look for (round 4, DESIGN.md section 2 "Waits the
compiler put where the source meant none"): a `s_waitcnt vmcnt(0)` right BEHIND a group of prefetch loads (vmcnt counts in order: the frame
then waits for what it has just asked for), one at the top of every conditionally executed step (the compiler could not know an earlier step
had waited), scratch_load (a spill reload sits behind vmcnt(0) too).  Out-of-line blocks appear where the compiler laid them out, not
where they execute.

    python tools/isa_waits.py crf_grad_ctc_kernelILi2E          # one kernel
    python tools/isa_waits.py grad_den_kernel --min-lines 300   # every instantiation, loops of at least 300 lines
    python tools/isa_waits.py ctc_pair_kernelILi1E --lgkm       # LDS / scalar waits too
    python tools/isa_waits.py KEY --asm /tmp/k.s                # reuse an assembly file (--keep writes one)
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assemble(path):
    sys.path.insert(0, ROOT)
    from cat_amd.build import assemble as asm   # every kernel family (cat_amd/csrc/k_*.hip), assembled in parallel, one file
    asm(path)


def functions(lines, key):
    out, name = None, None
    for l in lines:
        if out is None:
            if l.startswith("_Z") and ":" in l and key in l.split(":")[0]:
                out, name = [], l.split(":")[0]
        else:
            out.append(l)
            if "s_endpgm" in l:
                yield name, out
                out = None


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("key")
    ap.add_argument("--asm", help="assembly file to read instead of compiling")
    ap.add_argument("--keep", help="write the assembly here")
    ap.add_argument("--min-lines", type=int, default=150, help="skip loops shorter than this (header to next header)")
    ap.add_argument("--lgkm", action="store_true", help="list s_waitcnt lgkmcnt and ds_ instructions as well")
    a = ap.parse_args()
    path = a.asm or a.keep or os.path.join(tempfile.mkdtemp(prefix="isa_"), "crf.s")
    if not a.asm:
        assemble(path)
    lines = open(path).read().split("\n")
    pat = r"s_waitcnt vmcnt|global_load|global_store|global_atomic|buffer_|scratch_|s_barrier"
    if a.lgkm:
        pat += r"|s_waitcnt lgkmcnt|ds_read|ds_write|ds_max|ds_add|s_load"
    pat = re.compile(pat)
    for name, body in functions(lines, a.key):
        print("==", name, f"({len(body)} lines)")
        hdr = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l] + [len(body)]
        for lo, hi in zip(hdr[:-1], hdr[1:]):
            if hi - lo < a.min_lines:
                continue
            items = []
            for l in body[lo:hi]:
                if pat.search(l):
                    t = l.strip().split()
                    items.append(t[0] + (" " + t[1] if "waitcnt" in t[0] else ""))
            folded, prev, n = [], None, 0
            for k in items + [None]:
                if k == prev:
                    n += 1
                    continue
                if prev is not None:
                    folded.append(f"{prev} x{n}" if n > 1 else prev)
                prev, n = k, 1
            print(f"  loop at line {lo} .. {hi}:", " | ".join(folded))
    if a.keep:
        print("assembly kept in", path)


if __name__ == "__main__":
    sys.exit(main())
