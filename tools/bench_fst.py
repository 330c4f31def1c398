#!/usr/bin/env python
"""tools/bench_fst.py -- the loss on an ESTIMATED den_lm (cat_amd.den_lm.prep_den_lm on a synthetic corpus drawn
from a sparse second-order source) instead of bench.py's random T o LM graph: which kernel family the graph
compiler picks for a graph with the in-degree profile of a real n-gram LM, and how fast it is.
usage: python tools/bench_fst.py [sentences] [num_extra_lm_states] [B] [T]"""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cat_amd import ctc_crf, den_lm
from oracle import fst_io

nsent = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 250
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
V = 72
rng = np.random.default_rng(0)
trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
seqs = []
for _ in range(nsent):
    L, s, a, b = int(rng.integers(10, 40)), [], 0, 0
    for _ in range(L):
        c = 1 + int(rng.choice(V - 1, p=trans[a, b])); s.append(c); a, b = b, c
    seqs.append(s)
fst = os.path.join(tempfile.mkdtemp(), "den_lm.fst")
g = den_lm.prep_den_lm(seqs, V, fst, 4, 3, extra, selection=os.environ.get("DEN_LM_SELECTION", "count"))   # ("count": the graphs of rounds 1 - 5' records; "likelihood": the default rule since round 6)
ctx = ctc_crf.CRFContext(fst, 0)
st = ctc_crf._C.graph_stats(ctc_crf._C.graph_for(torch.device("cuda", 0)))
kind = f"factored (geometry {st['fac_geom']})" if st["fac"] else f"resident K={st['res_K']}" if st["res_K"] else "streaming"
print(f"den_lm from {nsent} sentences: S={g['S']} A={g['A']} max in/out degree {st['max_in_deg']}/{st['max_out_deg']} -> {kind} kernels")
gr = fst_io.read_fst(fst)
labels, ly = [], []
for b in range(B):
    lab = den_lm.random_labels_from_graph(gr, T // 6, np.random.default_rng(b))
    labels.append(lab); ly.append(len(lab))
x = torch.log_softmax(torch.randn(B, T, V, device="cuda") * 2.0, -1).requires_grad_(True)
lab_t = torch.tensor(np.concatenate(labels), dtype=torch.int32)
lx_t, ly_t = torch.full((B,), T, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32)
crit = ctc_crf.CTC_CRF_LOSS(lamb=0.1)
def step():
    x.grad = None
    crit(x, lab_t, lx_t, ly_t).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"B={B} T={T}: {dt * 1e3:.3f} ms/step, {B / dt:.0f} utt/s")
ctc_crf._C.profile_enable(True)
step()
print("kernels (ms):", {k: round(v, 3) for k, v in ctc_crf._C.profile_read().items() if v >= 0})
ctc_crf._C.profile_enable(False)
