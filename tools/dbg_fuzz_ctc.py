"""tools/dbg_fuzz_ctc.py seed -- the numerator of a fuzz case under the library's switches: which path goes wrong where?"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_crf, oracle
from tests.test_gpu_fuzz import _case
from tests.util import make_batch, small_synth, rel_err, crf_env
C = ctc_crf._C
seed = int(sys.argv[1])
V, H, d, B, T, sigma, lamb, mode, frac = _case(seed)
g, p = small_synth(tempfile.mkdtemp(), V, H, d, seed)
logits, labels, lx, ly = make_batch(g, B, T, V, seed=seed, ragged=True, scale=sigma, label_frac=frac, min_len=0)
rng = np.random.default_rng(seed)
if B >= 3 and seed % 3 == 0:
    lab = [list(labels[sum(ly[:i]):sum(ly[:i + 1])]) for i in range(B)]
    k = int(rng.integers(1, B)); lx = np.array(lx); lx[k] = seed % 2; lab[k] = lab[k][:int(lx[k])]
    ly = np.array([len(x) for x in lab], dtype=np.int32); labels = np.array([v for x in lab for v in x], dtype=np.int32)
print(f"seed {seed}: V={V} B={B} T={T} sigma={sigma} lx={list(map(int, lx))} ly={list(map(int, ly))}")
ref = oracle.ctc(logits, labels, lx, ly)
og = -np.asarray(ref[0])
x = torch.tensor(logits, device="cuda:0")
for name, env in (("default", {}), ("robust_ctc=1 (all log-domain)", {"CRF_ROBUST_CTC": 1}), ("robust=0 (no fallback)", {"CRF_ROBUST": 0}), ("ctc_tilt=0", {"CRF_CTC_TILT": 0})):
    with crf_env(**env):
        _, gc, ex = C.loss_fwd_bwd(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 0.0, 1.0, None, True)
        fb = C.last_fallback_counts(torch.cuda.current_stream().cuda_stream)
    gc = gc.cpu().numpy()
    print(f"-- {name}: fallback {fb}")
    for b in range(B):
        if lx[b] == 0: continue
        dmax = np.abs(gc[b] - og[b]).max(-1)
        bad = np.argwhere(~(dmax <= 1e-4)).ravel()
        fin = np.isfinite(gc[b]).all()
        rs = gc[b][:lx[b]].sum(-1)     # each frame's posteriors sum to -1 (c_ctc = 1 -> grad = -gamma)
        print(f"   b={b} finite {fin} err {rel_err(gc[b], og[b]):.2e} bad frames {bad[:12].tolist()} ({len(bad)})  row sums min {rs.min():.4f} max {rs.max():.4f}")
