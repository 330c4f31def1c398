import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import ctc_crf, oracle
from oracle import fst_io
from tests.test_gpu_fuzz import CAMPAIGN, _case
from tests.test_gpu_parity import _mode
from tests.util import make_batch, small_synth, rel_err
seed = int(sys.argv[1])
V, H, d, B, T, sigma, lamb, mode, frac = _case(300 + seed)
g, p = small_synth(tempfile.mkdtemp(), V, H, d, seed + CAMPAIGN)
_, labels, lx, ly = make_batch(g, B, T, V, seed=seed + CAMPAIGN, ragged=True, scale=1.0, label_frac=frac, min_len=0)
rng = np.random.default_rng(7000 + seed + CAMPAIGN)
raw = (rng.normal(size=(B, T, V)) * sigma).astype(np.float32)
x64 = raw.astype(np.float64); m = x64.max(-1, keepdims=True); lse = m + np.log(np.exp(x64 - m).sum(-1, keepdims=True))
logp = (x64 - lse).astype(np.float32)
size_average = bool(seed % 2)
print(dict(V=V, H=H, d=d, B=B, T=T, sigma=sigma, lamb=lamb, mode=mode, frac=frac, lx=list(map(int, lx)), ly=list(map(int, ly))))
ref = oracle.ctc_crf(fst_io.read_fst(p), logp, labels, lx, ly, lamb=lamb, size_average=size_average)
gl = ref["grad"].astype(np.float64)
gx = gl - np.exp(x64 - lse) * gl.sum(-1, keepdims=True)
tl, tx, ty = torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32)
for md in (mode, "factored", "streaming"):
    with _mode(md):
        ctx = ctc_crf.CRFContext(p, 0)
        xf = torch.tensor(raw, device="cuda:0", requires_grad=True)
        lf = ctc_crf.CTC_CRF_LOSS(lamb=lamb, size_average=size_average, fuse_log_softmax=True)(xf, tl, tx, ty); lf.backward()
        gf = xf.grad.cpu().numpy()
        xl = torch.tensor(logp, device="cuda:0", requires_grad=True)
        l2 = ctc_crf.CTC_CRF_LOSS(lamb=lamb, size_average=size_average)(xl, tl, tx, ty); l2.backward()
        g2 = xl.grad.cpu().numpy()
        del ctx
    print(f"mode {md}: loss fused {lf.item():.6f} plain {l2.item():.6f} oracle {ref['loss']:.6f}")
    for b in range(B):
        if lx[b] == 0: continue
        e1, e2 = rel_err(gf[b], gx[b]), rel_err(g2[b], ref['grad'][b])
        dd = np.abs(gf[b] - gx[b]); t_, v_ = np.unravel_index(np.argmax(dd), dd.shape)
        print(f"   b={b} lx={lx[b]} fused-vs-chain-rule {e1:.2e}  plain-vs-oracle {e2:.2e}; worst fused entry t={t_} v={v_}: ours {gf[b,t_,v_]:.6e} ref {gx[b,t_,v_]:.6e} (max |ref| {np.abs(gx[b]).max():.3e}); sum_v g ours-plain {g2[b,t_].sum():.6e} oracle {gl[b,t_].sum():.6e}")
# the two posterior matrices separately (plain log-probs), with the fallback counts and per-frame sums
C = ctc_crf._C
gden_ref = oracle.den(fst_io.read_fst(p), logp, lx)[0]
gctc_ref, cc_ref, valid = oracle.ctc(logp, labels, lx, ly)
for md in (mode, "streaming"):
    for sw in ({}, {"robust_ctc": 1}, {"ctc_tilt": 0}):
        with _mode(md), C.debug_opts(**sw):
            ctx = ctc_crf.CRFContext(p, 0)
            x = torch.tensor(logp, device="cuda:0")
            st = torch.cuda.current_stream().cuda_stream
            _, gden, _ = C.loss_fwd_bwd(x, None, tx, None, 1.0, 0.0, C.graph_for(x.device), True); fb1 = C.last_fallback_counts(st)
            _, gctc, ex = C.loss_fwd_bwd(x, tl, tx, ty, 0.0, -1.0, None, True); fb2 = C.last_fallback_counts(st)
            gden, gctc = gden.cpu().numpy(), gctc.cpu().numpy()
            del ctx
        print(f"mode {md} {sw}: fallback den-call {fb1} ctc-call {fb2}")
        for b in range(B):
            n = int(lx[b])
            if n == 0: continue
            sd, sc = gden[b, :n].sum(-1), gctc[b, :n].sum(-1)
            ed = np.abs(gden[b, :n] - gden_ref[b, :n]).max(); ec = np.abs(gctc[b, :n] - gctc_ref[b, :n]).max()
            tw = int(np.argmax(np.abs(sc - 1)))
            print(f"   b={b}: gamma_den max err {ed:.2e} row sums in [{sd.min():.7f}, {sd.max():.7f}]; gamma_ctc max err {ec:.2e} row sums in [{sc.min():.7f}, {sc.max():.7f}] worst frame {tw}; cost ctc ours {float(ex['costs_ctc'][b]):.4f} oracle {cc_ref[b]:.4f}")
