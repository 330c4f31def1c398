#!/usr/bin/env python
"""tools/hwq_probe.py [extra_streams] -- the step time inside a process that looks like a CAT trainer: the HIP runtime,
the allocator and `extra_streams` other streams (RCCL's, a torch copy stream) exist BEFORE `ctc_crf` is imported
(cat/ctc/train.py:118 imports it inside AMTrainer.__init__, after set_device + init_process_group, train.py:48-55), and
those streams keep getting work while the loss runs.  HIP maps all streams of a process onto GPU_MAX_HW_QUEUES hardware
queues (default 4); the loss needs its ONE side stream on another queue than the caller's and finds one by probing.
Run with GPU_MAX_HW_QUEUES unset / =4 / =8 and 0 / 2 / 6 extra streams to compare."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: E402
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(extra)]
bufs = [torch.zeros(1 << 18, device=dev) for _ in range(extra)]
host = torch.zeros(1 << 18).pin_memory()
def poke():                            # a little work on every extra stream: a copy (even ones) or a kernel (odd ones)
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            if i % 2 == 0:
                bufs[i].copy_(host, non_blocking=True)
            else:
                bufs[i].add_(1.0)
poke()
torch.cuda.synchronize()
import ctc_crf  # noqa: E402   (after HIP is up: nothing in the package may depend on the environment at import time)
from cat_amd.den_lm import synth_den_lm  # noqa: E402
from cat_amd.synth import make_batch  # noqa: E402
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
    poke()
for _ in range(4): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} extra streams {extra}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
