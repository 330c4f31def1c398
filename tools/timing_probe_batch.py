"""tools/timing_probe_batch.py -- in-kernel phase stamps of crf_batch_frame_kernel in a TIMING build (-DCRF_TIMING):
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
g = synth_den_lm(V, 8192, 32, seed=0, path=fst)
ctx = ctc_crf.CRFContext(fst, 0)
logits, labels, lx, ly = make_batch(g, B, T, V, seed=0, ragged=False)
x = torch.tensor(logits, device=dev, requires_grad=True)
crit = ctc_crf.CTC_CRF_LOSS(lamb=0.1)
for _ in range(2):
    x.grad = None
    crit(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly)).backward()
torch.cuda.synchronize()
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
    print(f"wg {wsel // 4} wave {wsel % 4}: batches {r[8]} bundles {r[9]} | " + " | ".join(f"{n} {v}" for n, v in zip(names, d)) + f" | total {r[6] - r[0]}")
