"""Seeded fuzz of the hot path against the fp64 oracle: shapes, vocabulary sizes at the kernels' boundaries (63 / 64 / 65, 128 / 129,
256 / 257), graph sizes, ragged lengths incl. empty utterances, label densities, lamb, size_average, and -- what found a silent bug in
round 5 -- the SCALE of the network output: from diffuse (sigma 0.5) to posteriors hundreds of nats apart (sigma 40), where the
rescaled fp32 recursions, the fp64 numerator chains, the grad kernels' descaling and the log-domain fallbacks each have a range that
must meet the next one's.  The reference has one arithmetic for all of it (log-domain fp32: den_calculate.cu:29-35,
gpu_ctc_kernels.h:87-458); here every case is checked in the kernel family a seeded choice picks, through the C ABI, against
oracle/crf_oracle.c in fp64: loss and every utterance's gradient within 1e-4, no NaN / inf anywhere."""
import numpy as np
import pytest

import oracle
from oracle import fst_io
from tests.test_gpu_parity import MODES, TOL, run_hip
from tests.util import make_batch, rel_err, small_synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crf():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctc_crf
    return ctc_crf


VS = [2, 3, 9, 24, 40, 63, 64, 65, 72, 128, 129, 200, 256, 257]
SIGMAS = [0.5, 2.0, 8.0, 20.0, 40.0]


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    V = int(VS[seed % len(VS)])
    H = int(max(V - 1, rng.integers(V - 1, 4 * V + 8)))
    d = int(rng.integers(1, min(V - 1, 12) + 1))
    B = int(rng.integers(1, 10))
    T = int(rng.integers(1, 90)) if seed < 112 else int(rng.integers(90, 400))
    sigma = float(SIGMAS[(seed // 3) % len(SIGMAS)])
    lamb = float([0.0, 0.01, 0.1, 1.0][seed % 4])
    mode = MODES[(seed * 7 + seed // 5) % len(MODES)]
    frac = int([2, 3, 6, 12][(seed // 2) % 4])
    return V, H, d, B, T, sigma, lamb, mode, frac


@pytest.mark.parametrize("seed", range(168))
def test_fuzz_vs_oracle(crf, tmp_path, seed):
    V, H, d, B, T, sigma, lamb, mode, frac = _case(seed)
    g, p = small_synth(tmp_path, V, H, d, seed)
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=seed, ragged=True, scale=sigma, label_frac=frac, min_len=0)
    rng = np.random.default_rng(seed)
    if B >= 3 and seed % 3 == 0:                        # an empty utterance / a one-frame utterance somewhere in the batch
        lab = [list(labels[sum(ly[:i]):sum(ly[:i + 1])]) for i in range(B)]
        k = int(rng.integers(1, B))
        lx = np.array(lx); lx[k] = seed % 2
        lab[k] = lab[k][:int(lx[k])]
        ly = np.array([len(x) for x in lab], dtype=np.int32)
        labels = np.array([v for x in lab for v in x], dtype=np.int32)
    size_average = bool(seed % 2)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=lamb, size_average=size_average)
    if not np.isfinite(ref["loss"]):
        pytest.skip("the oracle itself is not finite for this draw (an utterance without a valid alignment)")
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=lamb, size_average=size_average, mode=mode)
    what = dict(V=V, H=H, d=d, B=B, T=T, sigma=sigma, lamb=lamb, mode=mode, frac=frac, lx=list(map(int, lx)))
    assert np.isfinite(loss) and np.isfinite(grad).all(), what
    assert abs(loss - ref["loss"]) <= TOL * max(1.0, abs(ref["loss"])), (loss, ref["loss"], what)
    # 1e-4 in EVERY regime since round 5 -- also where the fast kernels cannot go (network outputs a hundred nats apart per frame, path
    # scores thousands of nats apart): forward / backward consistency checks, per-frame mass checks and the emission-weighted lost-term
    # bound hand such utterances to the log-domain fp64 fallbacks, which carry any input (1.5e-6 at sigma 40, T 300).
    tol = TOL
    for b in range(B):
        if lx[b] > 0:
            assert rel_err(grad[b], ref["grad"][b]) <= tol, (b, what)
        assert np.all(grad[b, lx[b]:] == 0.0), (b, what)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_fused_log_softmax_vs_oracle(crf, tmp_path, seed):
    """The same fuzz through `CTC_CRF_LOSS(fuse_log_softmax=True)` on RAW fp32 network outputs: numpy fp64 log_softmax, the
    oracle's loss and d loss / d log_probs, the chain rule d/dx = g - softmax(x) * sum_v g (cat/ctc/train.py:174-186 + autograd)."""
    import torch
    from tests.test_gpu_parity import _mode
    V, H, d, B, T, sigma, lamb, mode, frac = _case(300 + seed)
    g, p = small_synth(tmp_path, V, H, d, seed)
    _, labels, lx, ly = make_batch(g, B, T, V, seed=seed, ragged=True, scale=1.0, label_frac=frac, min_len=0)
    rng = np.random.default_rng(7000 + seed)
    raw = (rng.normal(size=(B, T, V)) * sigma).astype(np.float32)
    x64 = raw.astype(np.float64)
    m = x64.max(-1, keepdims=True)
    lse = m + np.log(np.exp(x64 - m).sum(-1, keepdims=True))
    logp = (x64 - lse).astype(np.float32)
    size_average = bool(seed % 2)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logp, labels, lx, ly, lamb=lamb, size_average=size_average)
    if not np.isfinite(ref["loss"]):
        pytest.skip("the oracle itself is not finite for this draw")
    gl = ref["grad"].astype(np.float64)
    gx = gl - np.exp(x64 - lse) * gl.sum(-1, keepdims=True)
    with _mode(mode):
        ctx = crf.CRFContext(p, 0)
        xf = torch.tensor(raw, device="cuda:0", requires_grad=True)
        lf = crf.CTC_CRF_LOSS(lamb=lamb, size_average=size_average, fuse_log_softmax=True)(
            xf, torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32))
        lf.backward()
        loss, grad = float(lf.item()), xf.grad.cpu().numpy()
        del ctx
    what = dict(V=V, H=H, d=d, B=B, T=T, sigma=sigma, lamb=lamb, mode=mode, frac=frac, lx=list(map(int, lx)))
    assert np.isfinite(loss) and np.isfinite(grad).all(), what
    assert abs(loss - ref["loss"]) <= TOL * max(1.0, abs(ref["loss"])), (loss, ref["loss"], what)
    for b in range(B):
        if lx[b] > 0 and np.abs(gx[b]).max() > 0:
            assert rel_err(grad[b], gx[b]) <= 2 * TOL, (b, what)     # (the oracle's own d loss / d log_probs is fp32: two roundings meet in the chain rule)


@pytest.mark.parametrize("H,d,B,T,sigma,kern", [(2304, 24, 3, 60, 2.0, "crf_fac2_pair_kernel"), (3072, 24, 2, 40, 8.0, "crf_fac2_pair_kernel"),
                                               (6144, 24, 2, 30, 2.0, "crf_batch_persist_kernel"), (6144, 24, 2, 30, 20.0, "crf_batch_persist_kernel"),
                                               (2048, 24, 4, 90, 20.0, "crf_fac_pair_kernel"), (2048, 24, 3, 200, 40.0, "crf_fac_pair_kernel")])
def test_fuzz_graphs_of_the_benchmark_size_class(crf, tmp_path, H, d, B, T, sigma, kern):
    """... and on den_lm of the benchmark generator's size classes, each on the kernels it takes BY ITSELF (one CU per recursion, two CUs, the
    utterance-minor kernels -- since round 6 all frames in one persistent launch), with peaked outputs."""
    import torch
    V = 72
    g, p = small_synth(tmp_path, V, H, d, 0)
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=H + T, ragged=True, scale=sigma)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    ctx = crf.CRFContext(p, 0)
    x = torch.tensor(logits, device="cuda:0", requires_grad=True)
    loss = crf.CTC_CRF_LOSS(lamb=0.1)(x, torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32))
    loss.backward()
    k = crf._C.last_den_kernel()
    grad = x.grad.cpu().numpy()
    del ctx
    assert k.startswith(kern), k
    assert np.isfinite(grad).all() and abs(loss.item() - ref["loss"]) <= TOL * abs(ref["loss"])
    for b in range(B):
        assert rel_err(grad[b], ref["grad"][b]) <= TOL, b
