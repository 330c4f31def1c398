#!/usr/bin/env python
"""tools/hwq_probe.py [extra_streams] -- the step time when the process owns extra streams (as with RCCL or a torch
copy stream): HIP maps all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); the loss's four
streams must not share one.  Run with GPU_MAX_HW_QUEUES=4 and 8 to compare."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_crf  # noqa: E402  (sets GPU_MAX_HW_QUEUES=8 unless it is set already)
import numpy as np, torch  # noqa: E402
from cat_amd.den_lm import synth_den_lm  # noqa: E402
from cat_amd.synth import make_batch  # noqa: E402
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(device=dev) for _ in range(extra)]
for s in streams:                      # make them real
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
fst = os.path.join(tempfile.mkdtemp(), "d.fst")
g = synth_den_lm(72, 2048, 24, seed=0, path=fst)
ctx = ctc_crf.CRFContext(fst, 0)
logits, labels, lx, ly = make_batch(g, 64, 1500, 72, seed=0, ragged=False)
x = torch.tensor(logits, device=dev, requires_grad=True)
lab_t, lx_t, ly_t = torch.tensor(labels), torch.tensor(lx), torch.tensor(ly)
crit = ctc_crf.CTC_CRF_LOSS(lamb=0.1)
def step():
    x.grad = None
    crit(x, lab_t, lx_t, ly_t).backward()
for _ in range(4): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} extra streams {extra}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")
