#!/usr/bin/env python
"""tools/tail_timeline.py <kernel_trace.csv> -- timeline of the LAST loss call in a rocprofv3 kernel trace:
every kernel's start/end relative to the end of the den recursions, to see what is left after them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
# the last kernel of a call: crf_finalize_kernel, or -- since round 5 folded into it -- the robust grad launch
fin = [i for i, k in enumerate(ks) if "crf_finalize" in k[2]] or [i for i, k in enumerate(ks) if "crf_robust_grad" in k[2]]
if len(fin) < 2: sys.exit("need at least two calls in the trace")
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2          # a call in the middle of the run
a, b = fin[which - 1] + 1, fin[which]
call = ks[a:b + 1]
t0 = call[0][0]
chain_end = max(e for s, e, n in call if any(k in n for k in ("fac_pair", "fac2_pair", "fac_pair2", "res_pair", "den_pair", "batch_frame")))
def short(n):
    n = n.replace("crf::", "").split("(")[0]
    return n[:60]
print(f"call: {len(call)} kernels, {(call[-1][1] - t0) / 1e3:.1f} us from first start to finalize end; den chains end at {(chain_end - t0) / 1e3:.1f} us")
for s, e, n in call:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  rel_chain_end {(s - chain_end) / 1e3:8.1f}  {short(n)}")
nxt = ks[b + 1] if b + 1 < len(ks) else None
if nxt: print(f"next kernel after finalize starts {(nxt[0] - call[-1][1]) / 1e3:.1f} us later: {short(nxt[2])}")
