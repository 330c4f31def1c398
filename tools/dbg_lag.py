"""tools/dbg_lag.py nats -- where does the lagged-scale window test go non-finite?  den-only / ctc-only / combined gradients."""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_crf, oracle
from oracle import fst_io
from cat_amd.den_lm import synth_den_lm
from cat_amd.synth import make_batch
nats = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
p = os.path.join(tempfile.mkdtemp(), "g.fst")
g = synth_den_lm(9, 24, 5, 7, path=p)
B, T, V = 3, 40, 10
rng = np.random.default_rng(22)
x = rng.normal(size=(B, T, V)) * 2.0
x[0, :, 9] += nats
x[1, 10:25, 9] += nats
m = x.max(-1, keepdims=True)
logits = (x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))).astype(np.float32)
_, labels, lx, ly = make_batch(g, B, T, 9, seed=3, ragged=True)
lx[:] = [40, 36, 31]
print("switches", ctc_crf._C.build_switches(), "nats", nats)
ctx = ctc_crf.CRFContext(p, 0)
C = ctc_crf._C
xt = torch.tensor(logits, device="cuda:0")
for name, cd, cc in (("den only", 1.0, 0.0), ("ctc only", 0.0, 1.0), ("both", 1.0, 1.1)):
    loss, grad, ex = C.loss_fwd_bwd(xt, torch.tensor(labels, dtype=torch.int32) if cc else None, torch.tensor(lx, dtype=torch.int32),
                                    torch.tensor(ly, dtype=torch.int32) if cc else None, cd, cc, C.graph_for(xt.device) if cd else None, True)
    gr = grad.cpu().numpy()
    bad = np.argwhere(~np.isfinite(gr).all(-1))
    print(name, "loss", float(loss.item()), "fallback", C.last_fallback_counts(torch.cuda.current_stream().cuda_stream), "non-finite (b,t):", bad[:12].tolist(), "n", len(bad))
    if cd and not cc:
        den = oracle.den(fst_io.read_fst(p), logits, lx)
        print("   costs_alpha", ex["costs_alpha"].cpu().numpy(), "oracle", np.asarray(den[1]).ravel(), "beta", ex["costs_beta"].cpu().numpy())
        fin = np.isfinite(gr).all(-1)
        print("   max |gamma - oracle| over finite frames", np.abs(gr - np.asarray(den[0]))[fin].max())
