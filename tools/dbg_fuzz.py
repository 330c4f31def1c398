"""tools/dbg_fuzz.py seed... -- a failing case of tests/test_gpu_fuzz.py taken apart: den-only and numerator-only costs / gradients vs the oracle."""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_crf, oracle
from oracle import fst_io
from tests.test_gpu_fuzz import _case
from tests.test_gpu_parity import _mode
from tests.util import make_batch, small_synth, rel_err
C = ctc_crf._C
for seed in map(int, sys.argv[1:]):
    V, H, d, B, T, sigma, lamb, mode, frac = _case(seed)
    td = tempfile.mkdtemp()
    g, p = small_synth(td, V, H, d, seed)
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=seed, ragged=True, scale=sigma, label_frac=frac, min_len=0)
    rng = np.random.default_rng(seed)
    if B >= 3 and seed % 3 == 0:
        lab = [list(labels[sum(ly[:i]):sum(ly[:i + 1])]) for i in range(B)]
        k = int(rng.integers(1, B)); lx = np.array(lx); lx[k] = seed % 2; lab[k] = lab[k][:int(lx[k])]
        ly = np.array([len(x) for x in lab], dtype=np.int32); labels = np.array([v for x in lab for v in x], dtype=np.int32)
    print(f"==== seed {seed}: V={V} H={H} d={d} B={B} T={T} sigma={sigma} lamb={lamb} mode={mode} frac={frac} lx={list(lx)} ly={list(ly)}")
    gr = fst_io.read_fst(p)
    den = oracle.den(gr, logits, lx)
    ctc = oracle.ctc(logits, labels, lx, ly) if hasattr(oracle, "ctc") else None
    with _mode(mode):
        ctx = ctc_crf.CRFContext(p, 0)
        x = torch.tensor(logits, device="cuda:0")
        _, gd, ex = C.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, C.graph_for(x.device), True)
        fb = C.last_fallback_counts(torch.cuda.current_stream().cuda_stream)
        gd = gd.cpu().numpy(); ca = ex["costs_alpha"].cpu().numpy(); cb = ex["costs_beta"].cpu().numpy()
        print("  den: kernel", C.last_den_kernel(), "fallback", fb)
        for b in range(B):
            e = rel_err(gd[b], np.asarray(den[0])[b]) if lx[b] > 0 else 0.0
            bad = np.argwhere(np.abs(gd[b] - np.asarray(den[0])[b]).max(-1) > 1e-4 * max(1e-30, np.abs(np.asarray(den[0])[b]).max()))[:, 0]
            print(f"   b={b} lx={lx[b]} alpha {ca[b]:.4f} beta {cb[b]:.4f} oracle {np.asarray(den[1]).ravel()[b]:.4f}  grad err {e:.2e}  bad frames {bad[:10].tolist()} ({len(bad)})")
        _, gc, ex = C.loss_fwd_bwd(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 0.0, 1.0, None, True)
        fb = C.last_fallback_counts(torch.cuda.current_stream().cuda_stream)
        gc = gc.cpu().numpy(); cc = ex["costs_ctc"].cpu().numpy()
        print("  ctc: fallback", fb, "invalid", ex["invalid"].cpu().numpy().tolist())
        if ctc is not None:
            for b in range(B):
                og = -np.asarray(ctc[0])[b]
                e = rel_err(gc[b], og) if lx[b] > 0 and np.abs(og).max() > 0 else 0.0
                bad = np.argwhere(np.abs(gc[b] - og).max(-1) > 1e-4 * max(1e-30, np.abs(og).max()))[:, 0]
                print(f"   b={b} ly={ly[b]} cost {cc[b]:.4f} oracle {np.asarray(ctc[1]).ravel()[b]:.4f} valid {np.asarray(ctc[2]).ravel()[b]} grad err {e:.2e} bad frames {bad[:10].tolist()} ({len(bad)})")
        del ctx
