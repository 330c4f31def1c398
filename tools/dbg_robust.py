import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctc_crf, oracle
from oracle import fst_io
from tests.util import small_synth, make_batch
from tests.test_gpu_parity import _mode
g, p = small_synth(pathlib.Path(tempfile.mkdtemp()), 9, 24, 5, 7)
B, T, V = 3, 40, 10
rng = np.random.default_rng(21); x = rng.normal(size=(B, T, V)) * 2.0
x[0, :, 9] += 300; x[1, 10:25, 9] += 300
m = x.max(-1, keepdims=True); logits = (x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))).astype(np.float32)
_, labels, lx, ly = make_batch(g, B, T, 9, seed=3, ragged=True); lx[:] = [40, 36, 31]
ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
print("oracle", ref["loss"], ref["costs_den"], ref["costs_ctc"])
core = ctc_crf._C
for mode in ["factored", "resident", "streaming", "batch"]:
    for rob in ["", "0", "1"]:
        core.debug_set("robust", None if rob == "" else int(rob))
        with _mode(mode):
            ctx = ctc_crf.CRFContext(p, 0)
            st = core.graph_stats(core.graph_for(torch.device("cuda", 0)))
            xx = torch.tensor(logits, device="cuda:0")
            loss, grad, ex = core.loss_fwd_bwd(xx, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 1 / 3, 1.1 / 3, core.graph_for(xx.device), True)
            print(mode, "robust=", rob or "auto", "fac", st["fac"], "K", st["res_K"], "loss", float(loss), "ca", ex["costs_alpha"].cpu().numpy(), "cb", ex["costs_beta"].cpu().numpy(),
                  "cc", ex["costs_ctc"].cpu().numpy(), "grad err", float(np.abs(grad.cpu().numpy() - ref["grad"]).max()))
            del ctx
