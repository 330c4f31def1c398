"""tools/timing_probe_batch.py -- in-kernel phase stamps of crf_batch_frame_kernel / crf_batch_persist_kernel in a TIMING build (-DCRF_TIMING):
launch 700 of the large-graph point, 4 workgroups x 4 waves; s_memtime ticks (100 MHz reference clock: 10 ns each)."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctc_crf  # noqa: E402
from cat_amd.ctc_crf import _C  # noqa: E402
from cat_amd.den_lm import synth_den_lm  # noqa: E402
from cat_amd.synth import make_batch  # noqa: E402

dev = torch.device("cuda:0")
B, T, V = 64, 1500, 72
fst = os.path.join(tempfile.mkdtemp(prefix="crfprobe_"), "den_lm.fst")
H, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 32)
g = synth_den_lm(V, H, D, seed=0, path=fst)
ctx = ctc_crf.CRFContext(fst, 0)
logits, labels, lx, ly = make_batch(g, B, T, V, seed=0, ragged=False)
x = torch.tensor(logits, device=dev, requires_grad=True)
crit = ctc_crf.CTC_CRF_LOSS(lamb=0.1)
for _ in range(2):
    x.grad = None
    crit(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly)).backward()
torch.cuda.synchronize()
print("kernel", _C.last_den_kernel(), "S", g["S"], "A", len(g["src"]))
tm = _C.timing_read()
if not tm:
    print("not a timing build")
    sys.exit(0)
names = ["entry->scalars", "scalars->task desc", "desc->chunk0 in LDS", "LDS->emissions in ring", "stream loop", "end (max, atomics)"]
for wsel in range(16):
    base = 14000 + wsel * 16
    r = tm[base:base + 10]
    if not r[0]:
        continue
    d = [r[i + 1] - r[i] for i in range(6)]
    r12 = tm[base:base + 12]
    tail = f" | persistent: own stores drained {r12[7] - r12[6]} | workgroup there {r12[10] - r12[7]} | grid barrier {r12[11] - r12[10]} | frame {r12[11] - r12[0]}" if r12[11] else ""
    print(f"wg {wsel // 4} wave {wsel % 4}: batches {r[8]} bundles {r[9]} | " + " | ".join(f"{n} {v}" for n, v in zip(names, d)) + f" | total {r[6] - r[0]}" + tail)

# persistent launch, frame 700: when did each workgroup start the frame, reach the grid barrier, get released?  (s_memrealtime: 100 MHz, device-wide)
arr = [(tm[5000 + b], tm[2000 + b], tm[3000 + b], int(tm[4000 + b]), b) for b in range(1000) if tm[2000 + b]]
if arr:
    t0 = min(a[0] for a in arr)
    st = sorted(a[0] - t0 for a in arr); av = sorted(a[1] - t0 for a in arr); rv = sorted(a[2] - t0 for a in arr)
    q = lambda v, f: v[min(len(v) - 1, int(len(v) * f))]
    print(f"{len(arr)} workgroups, units of 10 ns after the first frame start: starts median {q(st, .5)} max {st[-1]} | arrivals min {av[0]} p10 {q(av, .1)} median {q(av, .5)} p90 {q(av, .9)} max {av[-1]} | releases min {rv[0]} median {q(rv, .5)} max {rv[-1]}")
    import collections
    by = collections.defaultdict(list)
    for a in arr:
        by[a[3]].append(a)
    for k in sorted(by):
        v = by[k]
        bav = sorted(a[1] - t0 for a in v); brv = sorted(a[2] - t0 for a in v)
        print(f"  XCD {k} ({len(v)} workgroups, blocks = {sorted(set(a[4] & 7 for a in v))} mod 8): arrivals {bav[0]} / {q(bav, .5)} / {bav[-1]}, releases {brv[0]} / {brv[-1]}; latest blocks {[a[4] for a in sorted(v, key=lambda a: -a[1])[:3]]}")
