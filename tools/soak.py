"""tools/soak.py [calls [histories]] -- the loss at the benchmarked shape (B=64, T=1500, V=72, factored staged schedule), the same two batches
alternating for `calls` calls back to back into a NaN-poisoned workspace, with two extra busy streams in the process; every
result is compared with the first one of its batch (loss and a gradient checksum): an intermittent stage-ordering race would
show as a differing or NaN result.  histories = 3072: the same soak on a graph that takes the factored layout over TWO CUs
per recursion (hand-off through tagged granules every frame; the noise streams then compete for the CUs its workgroups spin on)."""
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

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
B, T, V = 64, 1500, 72
fst = os.path.join(tempfile.mkdtemp(prefix="crfsoak_"), "den_lm.fst")
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
g = synth_den_lm(V, H, 24, seed=0, path=fst)
ctx = ctc_crf.CRFContext(fst, 0)
gh = _C.graph_for(dev)
print("den_lm H=%d: S=%d A=%d, factored geometry %d, kernels: %s" % (H, g["S"], g["A"], _C.graph_stats(gh)["fac_geom"], _C.den_kernels(gh, B, T, V)))
data = []
for seed, ragged in ((0, False), (5, True)):
    lg, lab, lx, ly = make_batch(g, B, T, V, seed=seed, ragged=ragged)
    data.append((torch.tensor(lg, device=dev), torch.tensor(lab), torch.tensor(lx), torch.tensor(ly)))
_C.set_debug_poison(True)
noise = [torch.cuda.Stream(), torch.cuda.Stream()]
junk = torch.randn(4096, 4096, device=dev)
res = [[], []]
for i in range(calls):
    if i % 8 == 0:
        for s in noise:                      # other work in the process: a matmul and a copy on their own streams
            with torch.cuda.stream(s):
                junk2 = junk @ junk
                junk3 = junk.clone()
    x, lab, lx, ly = data[i & 1]
    loss, grad, _ = _C.loss_fwd_bwd(x, lab, lx, ly, 1.0 / B, 1.1 / B, gh, True)
    res[i & 1].append((loss, grad.double().sum(), grad.abs().max()))
torch.cuda.synchronize()
bad = 0
for k in range(2):
    l0, s0, m0 = [v.item() for v in res[k][0]]
    for (l, s, m) in res[k][1:]:
        l, s, m = l.item(), s.item(), m.item()
        if not (abs(l - l0) <= 1e-6 * abs(l0) and abs(s - s0) <= 1e-6 * max(1.0, abs(s0)) and abs(m - m0) <= 1e-6 * m0):
            bad += 1
    print(f"batch {k}: loss {l0:.6f}, grad sum {s0:.6e}, max |grad| {m0:.4e}, {len(res[k])} calls")
print("soak:", "OK" if bad == 0 else f"{bad} DIFFERING RESULTS")
sys.exit(0 if bad == 0 else 1)
