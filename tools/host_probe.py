#!/usr/bin/env python
"""tools/host_probe.py -- how long the HOST takes to enqueue one step (no sync), vs the GPU time per step.
If enqueue time ~ GPU time, something on the host blocks on the stream; if it is much smaller the host runs ahead."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cat_amd.synth import make_batch
from cat_amd import ctc_crf
from cat_amd.den_lm import synth_den_lm
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(); fst = os.path.join(tmp, "d.fst")
g = synth_den_lm(72, 2048, 24, seed=0, path=fst)
ctx = ctc_crf.CRFContext(fst, 0)
logits, labels, lx, ly = make_batch(g, 64, 1500, 72, seed=0, ragged=False)
x = torch.tensor(logits, device=dev, requires_grad=True)
labels_t, lx_t, ly_t = torch.tensor(labels), torch.tensor(lx), torch.tensor(ly)
crit = ctc_crf.CTC_CRF_LOSS(lamb=0.1)
def step():
    x.grad = None
    t0 = time.perf_counter()
    loss = crit(x, labels_t, lx_t, ly_t)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(5): step()
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.perf_counter()
hs = [step() for _ in range(N)]
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
f = np.array([h[0] for h in hs]) * 1e3; b = np.array([h[1] for h in hs]) * 1e3
print(f"host enqueue: forward {f.mean():.3f} ms (min {f.min():.3f} max {f.max():.3f}), backward {b.mean():.3f} ms; all {N} steps enqueued in {(t1 - t0) * 1e3:.1f} ms, GPU done after {(t2 - t0) * 1e3:.1f} ms = {(t2 - t0) * 1e3 / N:.3f} ms/step")
print("per-step forward enqueue ms:", " ".join(f"{v:.2f}" for v in f[:24]))
