"""Seeded fuzz of the hot path against the fp64 oracle: shapes, vocabulary sizes at the kernels' boundaries (63 / 64 / 65, 128 / 129,
256 / 257), graph sizes, ragged lengths incl. empty utterances, label densities, lamb, size_average, and -- what found a silent bug in
round 5 -- the SCALE of the network output: from diffuse (sigma 0.5) to posteriors hundreds of nats apart (sigma 40), where the
rescaled fp32 recursions, the fp64 numerator chains, the grad kernels' descaling and the log-domain fallbacks each have a range that
must meet the next one's.  The reference has one arithmetic for all of it (log-domain fp32: den_calculate.cu:29-35,
gpu_ctc_kernels.h:87-458); here every case is checked in the kernel family a seeded choice picks, through the C ABI, against
oracle/crf_oracle.c in fp64: loss and every utterance's gradient within 1e-4, no NaN / inf anywhere.
Round 6: both posterior matrices entry-wise; long utterances on the metric graph's size class; den_lm graphs ESTIMATED from text; the group width of the
utterance-minor kernels; a constructed case for the one blind spot of the forward / backward check; and CAMPAIGNS -- the same cases re-drawn with fresh seeds
(CRF_FUZZ_CAMPAIGN, tools/gpu_fuzz_campaign.sh): 36 of them in round 6 found two more precision holes (profiles/round6_fuzz_campaigns.txt)."""
import os

import numpy as np
import pytest

import oracle
from oracle import fst_io
from tests.test_gpu_parity import MODES, TOL, _mode, run_hip
from tests.util import crf_env, make_batch, post_err, rel_err, small_synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crf():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctc_crf
    return ctc_crf


# An EXTENDED campaign draws fresh cases with the same coverage structure: CRF_FUZZ_CAMPAIGN=n shifts every random draw (graph, batch, shape) by n * 100 000
# (tools/gpu_fuzz_campaign.sh; the suite itself runs campaign 0).
CAMPAIGN = int(os.environ.get("CRF_FUZZ_CAMPAIGN", "0")) * 100000

VS = [2, 3, 9, 24, 40, 63, 64, 65, 72, 128, 129, 200, 256, 257]
SIGMAS = [0.5, 2.0, 8.0, 20.0, 40.0]


def _case(seed):
    rng = np.random.default_rng(1000 + seed + CAMPAIGN)
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
    g, p = small_synth(tmp_path, V, H, d, seed + CAMPAIGN)
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=seed + CAMPAIGN, ragged=True, scale=sigma, label_frac=frac, min_len=0)
    rng = np.random.default_rng(seed + CAMPAIGN)
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
    # (utterance-minor modes: the group width drawn too -- 8 / 16 are what these batch sizes take by themselves, 32 / 64 what B > 16 / 32 does)
    ul = int([8, 16, 32, 64][(seed // 11) % 4]) if mode.startswith("batch") else 0
    with crf_env(CRF_BAT_UL=ul):
        loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=lamb, size_average=size_average, mode=mode)
    what = dict(V=V, H=H, d=d, B=B, T=T, sigma=sigma, lamb=lamb, mode=mode, frac=frac, lx=list(map(int, lx)), ul=ul)
    assert np.isfinite(loss) and np.isfinite(grad).all(), what
    assert abs(loss - ref["loss"]) <= TOL * max(1.0, abs(ref["loss"])), (loss, ref["loss"], what)
    # 1e-4 in EVERY regime since round 5 -- also where the fast kernels cannot go (network outputs a hundred nats apart per frame, path
    # scores thousands of nats apart): forward / backward consistency checks, per-frame mass checks and the emission-weighted lost-term
    # bound hand such utterances to the log-domain fp64 fallbacks, which carry any input (1.5e-6 at sigma 40, T 300).
    tol = TOL
    floor = 0.05 * (1.0 / B if size_average else 1.0)     # (tests/util.py rel_err: an utterance whose whole gradient is the cancellation of two equal posteriors)
    for b in range(B):
        if lx[b] > 0:
            assert rel_err(grad[b], ref["grad"][b], floor) <= tol, (b, what)
        assert np.all(grad[b, lx[b]:] == 0.0), (b, what)
    # ENTRY-WISE on the two posterior matrices (round 6): the combined gradient is a difference of two O(1) posteriors and can only
    # be judged norm-wise (tests/util.py rel_err); gamma_den and gamma_ctc themselves are non-negative, no cancellation: every entry >= 1e-3 within 1e-4
    # relative of the oracle's, in the same kernel family
    import torch
    core = crf._C
    gden_ref = oracle.den(fst_io.read_fst(p), logits, lx)[0]
    gctc_ref, _, valid = oracle.ctc(logits, labels, lx, ly)
    with _mode(mode), crf_env(CRF_BAT_UL=ul):
        ctx = crf.CRFContext(p, 0)
        x = torch.tensor(logits, device="cuda:0")
        tl, tx, ty = torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32)
        gden = core.loss_fwd_bwd(x, None, tx, None, 1.0, 0.0, core.graph_for(x.device), True)[1].cpu().numpy()
        gctc = core.loss_fwd_bwd(x, tl, tx, ty, 0.0, -1.0, None, True)[1].cpu().numpy()
        del ctx
    for b in range(B):
        n = int(lx[b])
        if n > 0:
            assert post_err(gden[b, :n], gden_ref[b, :n]) <= TOL, ("gamma_den", b, what)
            if valid[b]:
                assert post_err(gctc[b, :n], gctc_ref[b, :n]) <= TOL, ("gamma_ctc", b, what)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_fused_log_softmax_vs_oracle(crf, tmp_path, seed):
    """The same fuzz through `CTC_CRF_LOSS(fuse_log_softmax=True)` on RAW fp32 network outputs: numpy fp64 log_softmax, the
    oracle's loss and d loss / d log_probs, the chain rule d/dx = g - softmax(x) * sum_v g (cat/ctc/train.py:174-186 + autograd)."""
    import torch
    from tests.test_gpu_parity import _mode
    V, H, d, B, T, sigma, lamb, mode, frac = _case(300 + seed)
    g, p = small_synth(tmp_path, V, H, d, seed + CAMPAIGN)
    _, labels, lx, ly = make_batch(g, B, T, V, seed=seed + CAMPAIGN, ragged=True, scale=1.0, label_frac=frac, min_len=0)
    rng = np.random.default_rng(7000 + seed + CAMPAIGN)
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
            assert rel_err(grad[b], gx[b], 0.05 * (1.0 / B if size_average else 1.0)) <= 2 * TOL, (b, what)     # (the oracle's own d loss / d log_probs is fp32: two roundings meet in the chain rule)


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


@pytest.fixture(scope="module")
def size_class_graphs(tmp_path_factory):
    """The benchmark generator's graphs the long-utterance fuzz runs on: H = 2048 (the metric graph: one CU per recursion, 1024 threads) and
    H = 3072 (two CUs per recursion), built once."""
    d = tmp_path_factory.mktemp("fuzz_graphs")
    out = {}
    for H in (2048, 3072):
        g, p = small_synth(d, 72, H, 24, 0)
        out[H] = (g, p)
    return out


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_long_utterances_metric_size_class(crf, size_class_graphs, seed):
    """40 more seeds where the first 168 are thin: T = 400 ... 1 500 on the metric graph's size class (even seeds: S = 4 097, one CU per
    recursion, the staged schedule with the one-launch grad pass from T >= 256 on) and on the two-CU layout (odd seeds: S = 6 145), network outputs from
    diffuse to a hundred nats apart, ragged pairs, each on the kernels the graph takes by itself; loss and every utterance's gradient within 1e-4 of the fp64 oracle."""
    import torch
    rng = np.random.default_rng(5000 + seed + CAMPAIGN)
    H = 2048 if seed % 2 == 0 else 3072
    g, p = size_class_graphs[H]
    B = int(rng.integers(1, 3))
    T = int(rng.integers(400, 1501))
    sigma = float(SIGMAS[seed % len(SIGMAS)])
    lamb = float([0.0, 0.01, 0.1, 1.0][(seed // 2) % 4])
    logits, labels, lx, ly = make_batch(g, B, T, 72, seed=seed + CAMPAIGN, ragged=True, scale=sigma, label_frac=int([3, 6, 12][seed % 3]))
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=lamb)
    if not np.isfinite(ref["loss"]):
        pytest.skip("the oracle itself is not finite for this draw")
    ctx = crf.CRFContext(p, 0)
    x = torch.tensor(logits, device="cuda:0", requires_grad=True)
    loss = crf.CTC_CRF_LOSS(lamb=lamb)(x, torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32))
    loss.backward()
    k = crf._C.last_den_kernel()
    grad = x.grad.cpu().numpy()
    del ctx
    what = dict(H=H, B=B, T=T, sigma=sigma, lamb=lamb, lx=list(map(int, lx)), kernel=k)
    assert k.startswith("crf_fac_pair_kernel" if H == 2048 else "crf_fac2_pair_kernel"), what
    assert np.isfinite(loss.item()) and np.isfinite(grad).all(), what
    assert abs(loss.item() - ref["loss"]) <= TOL * max(1.0, abs(ref["loss"])), (loss.item(), ref["loss"], what)
    for b in range(B):
        assert rel_err(grad[b, :lx[b]], ref["grad"][b, :lx[b]], 0.05 / B) <= TOL, (b, what)
        assert np.all(grad[b, lx[b]:] == 0.0), (b, what)


@pytest.mark.parametrize("seed", range(33))
def test_fuzz_estimated_den_lm(crf, tmp_path, seed):
    """The fuzz on den_lm graphs ESTIMATED from text (cat_amd.den_lm.prep_den_lm: the graphs CAT trains on -- back-off arcs, in-degrees from 1 to hundreds,
    probabilities as weights) instead of the hashed synthetic T o LM: vocabulary 4 ... 48, corpus size, n-gram order 2 ... 4, no-prune order, number of extra
    states and BOTH selection rules drawn per seed; every kernel family in rotation (a forced family the graph does not fit falls to the next one by itself);
    output scale sigma 0.5 ... 40; label sequences taken from the corpus, so that the numerator's paths exist in the LM.  Loss, every utterance's gradient and
    both posterior matrices entry-wise within 1e-4 of the fp64 oracle."""
    import torch
    from cat_amd import den_lm
    rng = np.random.default_rng(9000 + seed + CAMPAIGN)
    V = int([4, 6, 9, 16, 24, 33, 48][seed % 7])
    order = int([2, 3, 4][(seed // 7) % 3])
    noprune = int(min(order, [1, 2, 3][seed % 3]))
    extra = int(rng.integers(0, 60))
    nsent = int(rng.integers(20, 600))
    conc = float([0.05, 0.3, 2.0][(seed // 2) % 3])                 # sparse ... dense second-order source
    trans = rng.dirichlet(np.ones(V - 1) * conc, size=(V, V))
    seqs = []
    for _ in range(nsent):
        sq, a, b_ = [], 0, 0
        for _ in range(int(rng.integers(3, 25))):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b_])); sq.append(c); a, b_ = b_, c
        seqs.append(sq)
    p = str(tmp_path / "est.fst")
    den_lm.prep_den_lm(seqs, V, p, order, noprune, extra, selection=["likelihood", "count"][seed % 2])
    g = fst_io.read_fst(p)
    B = int(rng.integers(1, 7))
    T = int(rng.integers(4, 160))
    sigma = float(SIGMAS[(seed // 3) % len(SIGMAS)])
    lamb = float([0.0, 0.01, 0.1, 1.0][seed % 4])
    mode = MODES[(seed * 5 + seed // 3) % len(MODES)]
    logits = (rng.normal(size=(B, T, V)) * sigma).astype(np.float32)
    x64 = logits.astype(np.float64)
    logits = (x64 - (x64.max(-1, keepdims=True) + np.log(np.exp(x64 - x64.max(-1, keepdims=True)).sum(-1, keepdims=True)))).astype(np.float32)
    lx = np.sort(rng.integers(1, T + 1, size=B))[::-1].astype(np.int32); lx[0] = T
    labels, ly = [], []
    for b in range(B):
        sq = seqs[int(rng.integers(0, nsent))]
        lab = sq[:max(1, min(len(sq), int(lx[b]) // int([2, 3, 6][seed % 3])))]
        labels += lab; ly.append(len(lab))
    labels, ly = np.array(labels, dtype=np.int32), np.array(ly, dtype=np.int32)
    ref = oracle.ctc_crf(g, logits, labels, lx, ly, lamb=lamb)
    if not np.isfinite(ref["loss"]):
        pytest.skip("the oracle itself is not finite for this draw (an utterance without a valid alignment)")
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=lamb, mode=mode)
    what = dict(V=V, order=order, noprune=noprune, extra=extra, nsent=nsent, S=int(g["S"]), A=len(g["src"]), B=B, T=T, sigma=sigma, lamb=lamb, mode=mode, lx=list(map(int, lx)), ly=list(map(int, ly)))
    assert np.isfinite(loss) and np.isfinite(grad).all(), what
    assert abs(loss - ref["loss"]) <= TOL * max(1.0, abs(ref["loss"])), (loss, ref["loss"], what)
    for b in range(B):
        assert rel_err(grad[b, :lx[b]], ref["grad"][b, :lx[b]], 0.05 / B) <= TOL, (b, what)
        assert np.all(grad[b, lx[b]:] == 0.0), (b, what)
    core = crf._C
    gden_ref = oracle.den(g, logits, lx)[0]
    gctc_ref, _, valid = oracle.ctc(logits, labels, lx, ly)
    with _mode(mode):
        ctx = crf.CRFContext(p, 0)
        x = torch.tensor(logits, device="cuda:0")
        tl, tx, ty = torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32)
        gden = core.loss_fwd_bwd(x, None, tx, None, 1.0, 0.0, core.graph_for(x.device), True)[1].cpu().numpy()
        gctc = core.loss_fwd_bwd(x, tl, tx, ty, 0.0, -1.0, None, True)[1].cpu().numpy()
        del ctx
    for b in range(B):
        n = int(lx[b])
        assert post_err(gden[b, :n], gden_ref[b, :n]) <= TOL, ("gamma_den", b, what)
        if valid[b]:
            assert post_err(gctc[b, :n], gctc_ref[b, :n]) <= TOL, ("gamma_ctc", b, what)


@pytest.mark.parametrize("mode", MODES)
def test_dominant_path_dropped_in_both_directions(crf, tmp_path, mode):
    """The residual blind spot DESIGN section 5 admitted, CONSTRUCTED: the forward / backward consistency check compares two log Z, so a
    path that BOTH directions drop leaves them agreeing on the same wrong value.  Three disjoint chains over tokens 1, 2, 3 (an LM of three sentences "1 1 1 ...",
    "2 2 2 ...", "3 3 3 ..."), 30 frames in three segments of 10, log-probs per frame:
                      head            middle          tail            total
        chain A (1)    0               -16             -16             -320      leads the forward recursion: after the head B is 120, C 160 nats behind
        chain B (2)   -12               0              -12             -240      DOMINANT by 80 nats -- and 2^-173 behind the frame's best after the head in the
        chain C (3)   -16              -16               0             -320      forward direction, after the tail in the backward one: flushed from the scaled fp32
    vectors in both (below even the denormals).  Forward log Z = -320 (A alone), backward log Z = -320 (C alone): crf_den_check_kernel sees nothing.  What catches
    it is the grad pass's frame mass: q lives on A's pairs, b on C's, every product q * b of a frame is zero -- not a normal float -- the utterance is flagged
    and the log-domain fallback redoes it (log Z = -240 + ..., posteriors on token 2).  Two paths cannot produce the hole at all (dropping B forward AND backward
    with a survivor in each direction needs B behind by > 101 nats at both ends and ahead in total: the middle would have to make up > 202 nats while staying within 101
    of the survivor on either side); with three, the survivors of the two directions are different chains, i.e. different states, and the frame mass is exactly zero.
    The test pins that: every kernel family returns the oracle's loss and gradient, and the denominator fallback has fired."""
    import math
    import torch
    from cat_amd import den_lm
    V, Tn = 4, 30
    L3, L2 = math.log(1 / 3), math.log(0.5)
    lm = dict(num_states=4, start=0, tok_in=[-1, 1, 2, 3], arcs=[[(1, 1, L3), (2, 2, L3), (3, 3, L3)], [(1, 1, L2)], [(2, 2, L2)], [(3, 3, L2)]],
              final=[-math.inf, L2, L2, L2])
    p = str(tmp_path / "three_chains.fst")
    den_lm.compose_ctc_topo(lm, V, p)
    lp = np.full((1, Tn, V), -60.0, dtype=np.float32)                   # (the blank far away: paths with blanks do not matter)
    for t in range(Tn):
        seg = t // 10
        lp[0, t, 1] = 0.0 if seg == 0 else -16.0
        lp[0, t, 2] = 0.0 if seg == 1 else -12.0
        lp[0, t, 3] = 0.0 if seg == 2 else -16.0
    labels, lx, ly = np.array([2], dtype=np.int32), np.array([Tn], dtype=np.int32), np.array([1], dtype=np.int32)
    ref = oracle.ctc_crf(fst_io.read_fst(p), lp, labels, lx, ly, lamb=0.1)
    cden = oracle.den(fst_io.read_fst(p), lp, lx)[1][0]
    assert abs(cden - (-240.0 + L3 + L2)) < 1e-6                        # the dominant chain alone: the others are e^-80 of it
    with _mode(mode):
        ctx = crf.CRFContext(p, 0)
        x = torch.tensor(lp, device="cuda:0", requires_grad=True)
        loss = crf.CTC_CRF_LOSS(lamb=0.1)(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly))
        loss.backward()
        nden, _ = crf._C.last_fallback_counts(torch.cuda.current_stream().cuda_stream)
        grad = x.grad.cpu().numpy()
        del ctx
    assert abs(loss.item() - ref["loss"]) <= TOL * abs(ref["loss"]), (loss.item(), ref["loss"])
    assert rel_err(grad[0], ref["grad"][0]) <= TOL
    assert nden == 1, "the fast kernels cannot carry this input: the utterance must have gone through the log-domain fallback"
