"""tools/dbg_fuzz_den.py seed -- the denominator of a fuzz case: default path vs everything through the log-domain fallback (robust=1)."""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_crf, oracle
from oracle import fst_io
from tests.test_gpu_fuzz import CAMPAIGN, _case
from tests.test_gpu_parity import _mode
from tests.util import make_batch, small_synth, rel_err, post_err, crf_env
C = ctc_crf._C
seed = int(sys.argv[1])
V, H, d, B, T, sigma, lamb, mode, frac = _case(seed)
g, p = small_synth(tempfile.mkdtemp(), V, H, d, seed + CAMPAIGN)
logits, labels, lx, ly = make_batch(g, B, T, V, seed=seed + CAMPAIGN, ragged=True, scale=sigma, label_frac=frac, min_len=0)
rng = np.random.default_rng(seed + CAMPAIGN)
if B >= 3 and seed % 3 == 0:
    lab = [list(labels[sum(ly[:i]):sum(ly[:i + 1])]) for i in range(B)]
    k = int(rng.integers(1, B)); lx = np.array(lx); lx[k] = seed % 2
print(f"seed {seed}: V={V} H={H} d={d} B={B} T={T} sigma={sigma} mode={mode} lx={list(map(int, lx))}")
den = oracle.den(fst_io.read_fst(p), logits, lx)
og = np.asarray(den[0])
for name, env in (("default", {}), ("robust=1", {"CRF_ROBUST": 1}), ("robust=0", {"CRF_ROBUST": 0})):
    with _mode(mode), crf_env(**env):
        ctx = ctc_crf.CRFContext(p, 0)
        x = torch.tensor(logits, device="cuda:0")
        _, gd, ex = C.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, C.graph_for(x.device), True)
        fb = C.last_fallback_counts(torch.cuda.current_stream().cuda_stream)
        gd = gd.cpu().numpy()
        del ctx
    print(f"-- {name}: fallback {fb}")
    for b in range(B):
        if lx[b] == 0: continue
        dmax = np.abs(gd[b] - og[b]).max(-1)
        bad = np.argwhere(~(dmax <= 1e-4)).ravel()
        print(f"   b={b} err {rel_err(gd[b], og[b]):.2e} bad frames {bad[:12].tolist()} ({len(bad)}) max frame err {dmax.max():.3e}")
        n = int(lx[b]); o64 = og[b, :n].astype(np.float64); m = o64 >= 1e-3
        if m.any():
            r = np.where(m, np.abs(gd[b, :n] - o64) / np.maximum(o64, 1e-30), 0.0)
            t_, v_ = np.unravel_index(np.argmax(r), r.shape)
            worst = np.sort(r.max(-1))[::-1][:6]
            print(f"      entry-wise (>= 1e-3): worst {r.max():.3e} at t={t_} v={v_}: ours {gd[b, t_, v_]:.6e} oracle {o64[t_, v_]:.6e}; frames over 5e-5: {(r.max(-1) > 5e-5).sum()} of {n}; worst frames {np.round(worst, 6).tolist()}")
            print(f"      costs: ours alpha {float(ex['costs_alpha'][b]):.6f} beta {float(ex['costs_beta'][b]):.6f} oracle {den[1][b]:.6f}")
if len(sys.argv) > 3:
    b, t = int(sys.argv[2]), int(sys.argv[3])
    with _mode(mode):
        ctx = ctc_crf.CRFContext(p, 0)
        x = torch.tensor(logits, device="cuda:0")
        _, gd, ex = C.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, C.graph_for(x.device), True)
        gd = gd.cpu().numpy()
    for tt in (t - 1, t, t + 1):
        d = logits[b, tt] - logits[b, tt].max()
        ours, orc = gd[b, tt], og[b, tt]
        idx = np.argsort(-np.maximum(ours, orc))[:6]
        print(f"frame {tt}: top labels by posterior:", [(int(v), f"d={d[v]:.1f}", f"ours={ours[v]:.4f}", f"oracle={orc[v]:.4f}") for v in idx], "sum ours", ours.sum(), "oracle", orc.sum())
