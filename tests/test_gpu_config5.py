"""BASELINE.json config #5's OWN graph (``synth_den_lm(5000, 32768, 64, seed 0)``: S = 65 537 states -- the first graph whose
state ids exceed 16 bits -- A = 4 292 272 arcs, V = 5000) against the fp64 oracle, on the utterance-minor kernels that produced the
config-#5 bench lines (profiles/round2_r2y2_point_c5.json: B = 8 per GPU, groups of 8; round2_r2z4_*: B = 64 on one GPU, two
groups of 32), on the factored and on the plain arc streams, with the workspace poisoned before every call; the exponent
bookkeeping of the same kernels over T = 3000 frames (6 000 rescales) on the S = 16 385 graph; and a recipe-shaped point
(V = 217 classes, lamb = 0.01: egs/aishell/exp/ctc-crf-cuside/config.json:12-13,29) on an estimated n-gram den_lm.
Reference semantics: src/ctc_crf/gpu_den/den_calculate.cu:75-103,189-227 (any graph, same two kernels)."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import fst_io
from tests.util import crf_env, make_batch, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def crf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctc_crf
    return ctc_crf


@pytest.fixture(scope="module")
def c5_graph(tmp_path_factory):
    from cat_amd.den_lm import synth_den_lm
    p = os.path.join(str(tmp_path_factory.mktemp("c5")), "den_lm_c5.fst")
    g = synth_den_lm(5000, 32768, 64, 0, path=p)
    assert g["S"] == 65537 and len(g["src"]) == 4292272
    return g, p, fst_io.read_fst(p)


@pytest.fixture(scope="module")
def c5_ctx(crf, c5_graph):
    """ONE graph for the four config-#5 cases (compiling 4.3 M arcs takes tens of seconds).  The factored rows of the arc
    streams are built with the graph; `bat_no_fac` per call then selects the plain streams of the same graph."""
    ctx = crf.CRFContext(c5_graph[1], 0)
    h = crf._C.graph_for(torch.device("cuda", 0))
    st = crf._C.graph_stats(h)
    assert st["S"] == 65537 and st["A"] == 4292272 and st["res_K"] == 0 and st["fac"] == 0
    yield ctx
    del ctx


_REF = {}   # oracle results shared by the two parameters of a test (a minute of CPU work each)


def _oracle_once(key, fn):
    if key not in _REF:
        _REF[key] = fn()
    return _REF[key]


def _slice(logits, labels, lx, ly, idx):
    off = np.concatenate([[0], np.cumsum(ly)])
    lab = np.concatenate([labels[off[i]:off[i + 1]] for i in idx]) if len(idx) else np.zeros(0, np.int32)
    return logits[idx], lab.astype(np.int32), lx[idx], ly[idx]


def _run(crf, p, logits, labels, lx, ly, lamb, poison=True):
    """One call through the C ABI (costs of both directions wanted), plus the numerator posteriors alone."""
    core = crf._C
    B = logits.shape[0]
    s = 1.0 / B
    core.set_debug_poison(poison)
    try:
        x = torch.tensor(logits, device="cuda:0")
        h = core.graph_for(x.device)
        loss, grad, ex = core.loss_fwd_bwd(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), s, s * (1 + lamb), h, True)
        torch.cuda.synchronize()
    finally:
        core.set_debug_poison(False)
    return float(loss.item()), grad.cpu().numpy() * B, {k: v.double().cpu().numpy() for k, v in ex.items() if k.startswith("costs")}, int(ex["invalid"].sum().item())


def _check(ref, idx, grad, costs, lamb):
    for j, b in enumerate(idx):
        assert abs(costs["costs_alpha"][b] - ref["costs_den"][j]) <= TOL * abs(ref["costs_den"][j]), (b, costs["costs_alpha"][b], ref["costs_den"][j])
        assert abs(costs["costs_beta"][b] - ref["costs_den"][j]) <= TOL * abs(ref["costs_den"][j]), b
        assert abs(costs["costs_ctc"][b] - ref["costs_ctc"][j]) <= TOL * abs(ref["costs_ctc"][j]), b
        e = rel_err(grad[b], ref["grad"][j])
        print(f"utterance {b}: gradient error vs fp64 oracle {e:.2e}")
        assert e <= TOL, (b, e)


@pytest.mark.parametrize("plain", [0, 1])
def test_config5_graph_b8_groups_of_8(crf, c5_graph, c5_ctx, plain):
    """(i) B = 8, ragged, T ~ 150: the UL = 8 path of the 143 ms line (per-GPU share of config #5 on 8 GPUs); EVERY utterance's loss
    terms and full [T, V] gradient against the fp64 oracle; factored arc streams (default) and plain ones."""
    g, p, gref = c5_graph
    B, T, V, lamb = 8, 150, 5000, 0.1
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=3, ragged=True)
    ref = _oracle_once("b8", lambda: oracle.ctc_crf(gref, logits, labels, lx, ly, lamb=lamb, size_average=False, threads=8))
    with crf_env(CRF_BAT_NO_FAC=plain):
        assert crf._C.den_kernels(crf._C.graph_for(torch.device("cuda", 0)), B, T, V) == "batch"
        loss, grad, costs, ninv = _run(crf, p, logits, labels, lx, ly, lamb)
    assert ninv == 0 and np.isfinite(loss)
    assert abs(loss - ref["loss"] / B) <= TOL * abs(ref["loss"] / B)
    _check(ref, np.arange(B), grad, costs, lamb)
    for b in range(B):
        assert np.all(grad[b, lx[b]:] == 0.0)


@pytest.mark.parametrize("plain", [0, 1])
def test_config5_graph_b64_two_groups_of_32(crf, c5_graph, c5_ctx, plain):
    """(ii) B = 64 on one GPU: two groups of 32 utterances (the 271 ms line); the first and last utterance of each group
    ({0, 31, 32, 63}) against the fp64 oracle, every other utterance through size-independent properties (forward logZ =
    backward logZ, posterior rows sum to one, rows past lx exactly zero)."""
    g, p, gref = c5_graph
    B, T, V, lamb = 64, 96, 5000, 0.1
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=5, ragged=True)
    idx = np.array([0, 31, 32, 63])
    ref = _oracle_once("b64", lambda: oracle.ctc_crf(gref, *_slice(logits, labels, lx, ly, idx), lamb=lamb, size_average=False, threads=4))
    with crf_env(CRF_BAT_NO_FAC=plain):
        loss, grad, costs, ninv = _run(crf, p, logits, labels, lx, ly, lamb)
        # numerator posteriors alone (c_den = 0): gamma_den = grad + (1 + lamb) gamma_ctc
        x = torch.tensor(logits, device="cuda:0")
        gctc = crf._C.loss_fwd_bwd(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 0.0, -1.0, None, True)[1].cpu().numpy()
    assert ninv == 0 and np.isfinite(loss)
    _check(ref, idx, grad, costs, lamb)
    assert np.allclose(costs["costs_alpha"], costs["costs_beta"], rtol=3e-5, atol=0)
    gden = grad + (1 + lamb) * gctc
    assert gden.min() >= -2e-5
    for b in range(B):
        n = int(lx[b])
        assert np.allclose(gden[b, :n].sum(-1), 1.0, atol=3e-4), b
        assert np.all(grad[b, n:] == 0.0)


def test_utterance_minor_kernels_3000_frames(crf, tmp_path):
    """T = 3000 on the S = 16 385 / A = 545 k graph (utterance-minor kernels): two exact power-of-two rescales per frame and
    utterance, exponents carried per utterance over 3 000 frames; costs of every utterance and the full gradient of one against
    the fp64 oracle."""
    from cat_amd.den_lm import synth_den_lm
    p = os.path.join(str(tmp_path), "large.fst")
    g = synth_den_lm(72, 8192, 32, 0, path=p)
    assert g["S"] == 16385
    B, T, V, lamb = 8, 3000, 72, 0.1
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=1, ragged=True)
    gref = fst_io.read_fst(p)
    idx = np.array([0, 5])
    ref = oracle.ctc_crf(gref, *_slice(logits, labels, lx, ly, idx), lamb=lamb, size_average=False, threads=2)
    dn, ca_all, cb_all = oracle.den(gref, logits[[2, 7]], lx[[2, 7]])
    ctx = crf.CRFContext(p, 0)
    assert crf._C.den_kernels(crf._C.graph_for(torch.device("cuda", 0)), B, T, V) == "batch"
    loss, grad, costs, ninv = _run(crf, p, logits, labels, lx, ly, lamb)
    del ctx
    assert ninv == 0 and np.isfinite(loss)
    _check(ref, idx, grad, costs, lamb)
    for j, b in enumerate([2, 7]):
        assert abs(costs["costs_alpha"][b] - ca_all[j]) <= TOL * abs(ca_all[j])
        assert abs(costs["costs_beta"][b] - ca_all[j]) <= TOL * abs(ca_all[j])
    assert np.allclose(costs["costs_alpha"], costs["costs_beta"], rtol=3e-5, atol=0)


def test_recipe_shaped_point_v217(crf, tmp_path):
    """AISHELL-shaped call: V = 217 output units (egs/aishell/exp/ctc-crf-cuside/config.json:29), lamb = 0.01 (:12-13), an estimated
    4-gram den_lm over those units (cat_amd.den_lm.prep_den_lm, the tool-chain of cat/utils/tool/prep_den_lm.sh), through
    CTC_CRF_LOSS; loss and gradient against the fp64 oracle in the default kernel family and on the generic layout."""
    from cat_amd import den_lm
    rng = np.random.default_rng(17)
    V = 217
    # a synthetic corpus with a Zipf-like unit distribution and local structure (bigram chains), 1 500 "sentences"
    succ = [rng.permutation(np.arange(1, V))[:6] for _ in range(V)]
    sents = []
    for _ in range(1500):
        n = int(rng.integers(4, 18))
        s = [int(rng.integers(1, V))]
        while len(s) < n:
            s.append(int(rng.choice(succ[s[-1]])) if rng.random() < 0.8 else int(rng.integers(1, V)))
        sents.append(s)
    p = os.path.join(str(tmp_path), "den_lm_v217.fst")
    den_lm.prep_den_lm(sents, V, p, ngram_order=4, no_prune_ngram_order=2, num_extra_states=300, selection="count")
    gref = fst_io.read_fst(p)
    B, T, lamb = 6, 120, 0.01
    logits = np.log(np.random.default_rng(3).dirichlet(np.ones(V) * 0.3, size=(B, T)).astype(np.float64) + 1e-12)
    logits = (logits - np.log(np.exp(logits).sum(-1, keepdims=True))).astype(np.float32)
    lx = np.array([120, 117, 100, 96, 80, 61], dtype=np.int32)
    ly = np.array([len(sents[i]) for i in range(B)], dtype=np.int32)       # whole training transcripts: accepted by the den_lm
    labels = np.concatenate([np.array(sents[i]) for i in range(B)]).astype(np.int32)
    ref = oracle.ctc_crf(gref, logits, labels, lx, ly, lamb=lamb)
    for mode in ({}, {"CRF_NO_FACTORED": 1}):
        with crf_env(**mode):
            ctx = crf.CRFContext(p, 0)
            x = torch.tensor(logits, device="cuda:0", requires_grad=True)
            loss = crf.CTC_CRF_LOSS(lamb=lamb)(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly))
            loss.backward()
            got, grad = float(loss.item()), x.grad.cpu().numpy()
            del ctx
        assert abs(got - ref["loss"]) <= TOL * abs(ref["loss"]), (mode, got, ref["loss"])
        assert rel_err(grad, ref["grad"]) <= TOL, mode
