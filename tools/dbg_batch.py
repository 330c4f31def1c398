"""tools/dbg_batch.py H d [B T] -- the utterance-minor kernels on a benchmark-generator den_lm (V = 72): denominator costs and gradient of the
persistent launch and of the per-frame launches against the fp64 oracle, fallback counts, kernel taken (robust fallback OFF: a wrong fast
result must show, not be repaired)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_crf, oracle
from oracle import fst_io
from cat_amd.den_lm import synth_den_lm
from tests.util import make_batch, rel_err, crf_env
C = ctc_crf._C
H, d = int(sys.argv[1]), int(sys.argv[2])
B, T = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 24)
V = 72
p = os.path.join(tempfile.mkdtemp(), "g.fst")
g = synth_den_lm(V, H, d, seed=0, path=p)
logits, labels, lx, ly = make_batch(g, B, T, V, seed=1, ragged=False)
t0 = time.time()
den = oracle.den(fst_io.read_fst(p), logits[:4], lx[:4])
print(f"S={g['S']} A={len(g['src'])} oracle (4 utterances) {time.time() - t0:.1f} s")
for name, env in (("persistent", {"CRF_BAT_PERSIST": 1}), ("per frame", {"CRF_BAT_PERSIST": 0})):
    with crf_env(CRF_ROBUST=0, **env):
        ctx = ctc_crf.CRFContext(p, 0)
        x = torch.tensor(logits, device="cuda:0")
        for rep in range(2):
            _, gd, ex = C.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, C.graph_for(x.device), True)
            torch.cuda.synchronize()
        fb = C.last_fallback_counts(torch.cuda.current_stream().cuda_stream)
        gd = gd.cpu().numpy()
        print(f"-- {name}: kernel {C.last_den_kernel()} fallback counts {fb} costs alpha {ex['costs_alpha'][:4].cpu().numpy()} beta {ex['costs_beta'][:4].cpu().numpy()} oracle {den[1][:4]}")
        for b in range(4):
            print(f"   b={b} gradient err {rel_err(gd[b], np.asarray(den[0])[b]):.2e}")
        del ctx
