"""GPU parity tests proper: the HIP path (through the C ABI, via the reference-shaped Python surface)
against the CPU oracle, the committed golden vectors, the reference's own denominator kernels when
oracle/_ref is present, and size-independent invariants at BASELINE sizes.

Tolerance: BASELINE.json north_star states loss and grad within 1e-4 relative."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import fst_io
from tests.util import crf_env, graph_to_file, log_softmax_np, make_batch, post_err, rel_err, small_synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def crf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctc_crf
    return ctc_crf


MODES = ["factored", "factored_768", "factored_rcl", "factored_k2", "factored_k2_1024", "factored_pair2", "factored_pair2_512", "resident", "streaming", "batch", "batch_frames"]


_env = crf_env   # (debug switches of the library, tests/util.py)


class _mode(crf_env):
    """The denominator has four kernel families: factored register-resident (one CU per recursion; needs the
    T o LM structure, else it falls back to the next), generic register-resident, utterance-minor ("batch") and streaming
    (fallbacks for graphs that do not fit registers).  no_factored / no_resident are read when a graph is created."""

    def __init__(self, mode):
        self.mode = mode
        kw = dict(
            CRF_NO_RESIDENT=mode in ("streaming", "batch", "batch_frames"),
            CRF_NO_FACTORED=not mode.startswith("factored"),
            # "factored_rcl": the factored kernels' 768-thread variant with the row constants in an LDS table (what graphs with more
            # than three slices of rows per wave take by themselves), forced for every graph with the T o LM structure
            CRF_FAC_RCL=mode == "factored_rcl",
            CRF_FAC_NO_RCL=mode == "factored_rc",      # row constants in registers even for long rows
            # "factored_k2": the factored kernels over TWO compute units per recursion (what T o LM graphs of 120 k - 240 k arcs
            # take by themselves), forced for every graph with the structure
            # ("factored_k2_1024": the same on the 1024-thread geometry -- built in round 4, slower, on request only)
            CRF_FAC_K2=mode in ("factored_k2", "factored_k2_1024"),
            # "factored_pair2": the factored kernels with TWO utterances per workgroup (what batches above CUs / 4 utterances take by
            # themselves), forced for any batch; 0 otherwise, so that the other modes test the one-utterance kernels at any batch size
            # ("factored_pair2_512", round 5: the same kernel on its OWN layout -- 512 threads x 30 chunks, built beside the planner's
            # 1024-thread main layout; what batches above 3/4 of the device take by themselves)
            CRF_FAC_PAIR2=mode in ("factored_pair2", "factored_pair2_512"),
            # "factored": the planner's own order (1024 threads x 15 chunks first since round 3); "factored_768": the 768-thread
            # geometries first (row constants in registers where the rows allow), which is also what the two-utterance kernels need
            CRF_FAC_THREADS=768 if mode in ("factored_768", "factored_pair2") else 1024 if mode == "factored_k2_1024" else 0)
        # "batch": the utterance-minor kernels, what graphs that fit no register-resident layout take by default -- since round 6 all
        # frames in ONE persistent launch with a grid barrier per frame; "batch_frames": the same frame body as one launch per frame
        # (rounds 2 - 5; the fallback when the grid is not co-resident); "streaming": the persistent one-workgroup-per-utterance
        # fallback (no_batch is read per call)
        if mode in ("streaming", "batch", "batch_frames"):
            kw["CRF_NO_BATCH"] = mode == "streaming"
            kw["CRF_BAT_PERSIST"] = mode != "batch_frames"
        super().__init__(**kw)


def run_hip(crf, den_lm, logits, labels, lx, ly, lamb=0.1, size_average=True, mode="factored"):
    with _mode(mode):   # (CRF_NO_RESIDENT / CRF_NO_FACTORED are read when the graph is created, CRF_NO_BATCH per call)
        ctx = crf.CRFContext(den_lm, 0)
        st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
        if mode in ("streaming", "batch", "batch_frames"):
            assert st["res_K"] == 0 and st["fac"] == 0
        if mode == "resident":
            assert st["fac"] == 0
        if mode == "factored_768" and st["fac"]:
            assert st["fac_geom"] != 4
        if mode == "factored_rcl" and st["fac"]:
            assert st["fac_geom"] in (1, 2)                      # (2: neither 768-thread geometry took the graph)
        if mode == "factored_k2" and st["fac"]:
            assert st["fac_geom"] in (3, 2)
        if mode == "factored_k2_1024" and st["fac"]:
            assert st["fac_geom"] in (5, 2)
        x = torch.tensor(logits, device="cuda:0", requires_grad=True)
        crit = crf.CTC_CRF_LOSS(lamb=lamb, size_average=size_average)
        loss = crit(x, torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32),
                    torch.tensor(ly, dtype=torch.int32))
        loss.backward()
        out = float(loss.item()), x.grad.detach().cpu().numpy()
        del ctx
    return out


@pytest.mark.parametrize("mode", MODES)
def test_fixture_kat(crf, golden_dir, mode):
    """Exactly src/ctc_crf/test/main.py:15-35 (the reference's only test), with the value pinned."""
    k = json.load(open(os.path.join(golden_dir, "kat_fixture.json")))
    logits = np.log(np.array(k["probs"], dtype=np.float32))[None]
    loss, grad = run_hip(crf, os.path.join(golden_dir, "den_lm_fixture.fst"), logits, k["labels"], [5], [3], lamb=k["lamb"], mode=mode)
    assert abs(loss - k["loss"]) <= TOL * abs(k["loss"])
    assert rel_err(grad[0], np.array(k["grad"])) <= TOL


def test_fixture_costs_and_gpu_den_gpu_ctc(crf, golden_dir):
    """The _C-level mirrors (binding.cpp:65-117 signatures) against the brute-force KAT."""
    k = json.load(open(os.path.join(golden_dir, "kat_fixture.json")))
    core = crf._C
    ctx = crf.CRFContext(os.path.join(golden_dir, "den_lm_fixture.fst"), 0)
    logits = torch.tensor(np.log(np.array(k["probs"], dtype=np.float32))[None], device="cuda:0")
    gd = torch.zeros_like(logits)
    ca, cb = torch.zeros(1, device="cuda:0"), torch.zeros(1, device="cuda:0")
    core.gpu_den(logits, gd, torch.tensor([5], dtype=torch.int32).cuda(), ca, cb)
    assert abs(ca.item() - k["logZ_den"]) <= TOL * abs(k["logZ_den"])
    assert abs(cb.item() - k["logZ_den"]) <= TOL * abs(k["logZ_den"])
    assert rel_err(gd[0].cpu().numpy(), np.array(k["gamma_den"])) <= TOL
    assert post_err(gd[0].cpu().numpy(), np.array(k["gamma_den"])) <= TOL
    act = logits.transpose(0, 1).contiguous()
    gc = torch.zeros_like(act)
    cc = torch.zeros(1)
    core.gpu_ctc(act, gc, torch.tensor(k["labels"], dtype=torch.int32), torch.tensor([3], dtype=torch.int32),
                 torch.tensor([5], dtype=torch.int32), 1, cc, 0)
    assert abs(cc.item() - k["logp_ctc"]) <= TOL * abs(k["logp_ctc"])
    assert rel_err(gc.transpose(0, 1)[0].cpu().numpy(), np.array(k["gamma_ctc"])) <= TOL
    del ctx


@pytest.mark.parametrize("force_batch", [0, 1])
def test_random_golden(crf, golden_dir, force_batch):
    """Random tiny GENERAL graphs (states entered with several labels, states nobody enters) against the brute-force
    enumerator's values; force_batch: the same through the utterance-minor kernels, whose arc streams take only the rows
    with one entering pair -- these graphs exercise the per-row path for the others inside the same launches."""
    cases = json.load(open(os.path.join(golden_dir, "kat_random.json")))
    core = crf._C
    with _env(CRF_FORCE_BATCH=force_batch):
        for c in cases:
            ctx = crf.CRFContext(os.path.join(golden_dir, c["fst"]), 0)
            lg = torch.tensor(np.array(c["logits"], dtype=np.float32)[None], device="cuda:0")
            T = lg.shape[1]
            gd = torch.zeros_like(lg)
            ca, cb = torch.zeros(1, device="cuda:0"), torch.zeros(1, device="cuda:0")
            core.gpu_den(lg, gd, torch.tensor([T], dtype=torch.int32).cuda(), ca, cb)
            assert abs(ca.item() - c["logZ_den"]) <= TOL * max(1.0, abs(c["logZ_den"])), c["fst"]
            assert abs(cb.item() - c["logZ_den"]) <= TOL * max(1.0, abs(c["logZ_den"])), c["fst"]
            assert rel_err(gd[0].cpu().numpy(), np.array(c["gamma_den"])) <= TOL, c["fst"]
            del ctx


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("seed,B,T,vocab,hist,fan", [(0, 3, 20, 8, 16, 4), (1, 5, 37, 12, 40, 6), (2, 2, 64, 72, 128, 16)])
def test_synth_vs_oracle_ragged(crf, tmp_path, seed, B, T, vocab, hist, fan, mode):
    g, p = small_synth(tmp_path, vocab, hist, fan, seed)
    logits, labels, lx, ly = make_batch(g, B, T, vocab, seed=seed, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode=mode)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL
    for b in range(B):  # rows past lx[b] are exactly zero (copy_grad returns early, den_calculate.cu:239)
        assert np.all(grad[b, lx[b]:] == 0.0)


@pytest.mark.parametrize("B,ul", [(20, 8), (41, 8), (33, 8), (64, 16), (40, 16), (70, 32), (70, 64), (41, -8), (64, -16), (70, -32), (70, -64)])
def test_batch_kernels_utterance_groups(crf, tmp_path, B, ul):
    """Utterance-minor kernels with several utterance groups: 3 groups (6 combos on 8 XCDs, uneven), 6 groups (12 combos:
    two per XCD on four of them), 3 / 3 / 2 groups of 16 / 32 / 64 with padding utterances in the last one -- the
    (group, direction) -> XCD / row-chunk decode of crf_batch_frame_kernel (fewer, exactly and more than eight combos: B=41:
    12, B=33: 10, B=64 / UL=16: 8) and the group-major vectors."""
    g, p = small_synth(tmp_path, 12, 40, 6, 5)
    logits, labels, lx, ly = make_batch(g, B, 31, 12, seed=B, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    # the graph is T o LM: FACTORED streams (U entries, folded tail rows, fused backward rows; tasks of at most 8 / 8 / 4 / 2
    # bundles for groups of 64 / 32 / 16 / 8); ul < 0: the same groups on the plain streams (CRF_BAT_NO_FAC=1)
    with _env(CRF_BAT_UL=abs(ul), CRF_BAT_NO_FAC=1 if ul < 0 else 0):
        loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode="batch")
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL
    for b in range(B):
        assert rel_err(grad[b], ref["grad"][b]) <= TOL, b
        assert np.all(grad[b, lx[b]:] == 0.0)


def test_default_kernel_for_a_batch_beyond_the_device(crf, tmp_path):
    """Which kernel a batch with 2 B workgroups > CUs takes BY DEFAULT (round-3 advisor: the docs said "the two-utterance kernel" while
    the planner's first geometry, 1024 threads, had none).  Pinned -- round 5: on the planner's own layout the two-utterance kernel on
    its SECOND layout (512 threads x 30 chunks, built beside the 1024-thread one; `no_facp` = the one-utterance kernel in rounds of
    the device, as in rounds 3 - 4); with the 768-thread geometries (`fac_threads` = 768) the two-utterance kernel on the main layout;
    all within 1e-4 of the oracle."""
    g, p = small_synth(tmp_path, 24, 96, 8, 13)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    B, T, V = ncu // 2 + 3, 21, 24
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=5, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    for thr, nop, want in ((0, 0, ",512,30,"), (0, 1, "crf_fac_pair_kernel<"), (768, 0, ",768,")):
        with _env(CRF_FAC_THREADS=thr, CRF_NO_FACP=nop):
            ctx = crf.CRFContext(p, 0)
            x = torch.tensor(logits, device="cuda:0", requires_grad=True)
            loss = crf.CTC_CRF_LOSS(lamb=0.1)(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly))
            loss.backward()
            kern = crf._C.last_den_kernel()
            assert (kern.startswith(want) if want.startswith("crf") else kern.startswith("crf_fac_pair2_kernel<") and want in kern), (thr, nop, kern)
            assert abs(loss.item() - ref["loss"]) <= TOL * abs(ref["loss"]) and rel_err(x.grad.cpu().numpy(), ref["grad"]) <= TOL
            del ctx


@pytest.mark.parametrize("geom", ["factored_pair2", "pair2_rcl", "pair2_512"])
@pytest.mark.parametrize("B", [1, 2, 7, 16])
def test_two_utterances_per_workgroup(crf, tmp_path, geom, B):
    """The two-utterance kernels (float2 state vectors, one gather for both utterances) do, per utterance, the arithmetic of the
    one-utterance kernels in the same order (the costs come out bit-identical; the row epilogues' multiply-adds are contracted
    differently by the compiler, so the rows agree to rounding) -- on ragged pairs (the shorter utterance's sums are taken when it
    ends, its rows go to a dump row afterwards), with an empty utterance, a one-frame utterance, an odd batch (the last pair has one
    utterance), with the row constants in registers and in the LDS table; and within 1e-4 of the fp64 oracle."""
    g, p = small_synth(tmp_path, 24, 96, 8, 13)
    T, V = 53, 24
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=B, ragged=True)
    lx = np.array(lx); ly = np.array(ly)
    lab = [list(labels[sum(ly[:i]):sum(ly[:i + 1])]) for i in range(B)]
    if B >= 7:
        lx[2], lab[2] = 0, []                               # an empty utterance (partner of utterance 3)
        lx[5], lab[5] = 1, lab[5][:1]                       # a one-frame utterance (partner of utterance 4, full length: ends 52 frames earlier)
    ly = np.array([len(x) for x in lab], dtype=np.int32)
    labels = np.array([v for x in lab for v in x], dtype=np.int32)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1, size_average=False)
    outs = {}
    for pair in (0, 1):
        # (pair2_512: the planner's 1024-thread main layout for the one-utterance run, the second layout -- 512 threads x 30 chunks --
        # for the two-utterance run: other rows, other lanes, the same sums)
        with _mode("factored_rcl" if geom == "pair2_rcl" else "factored" if geom == "pair2_512" else "factored_768"), _env(CRF_FAC_PAIR2=pair):
            ctx = crf.CRFContext(p, 0)
            x = torch.tensor(logits, device="cuda:0")
            crf._C.set_debug_poison(True)
            try:
                loss, grad, ex = crf._C.loss_fwd_bwd(x, torch.tensor(labels), torch.tensor(lx.astype(np.int32)), torch.tensor(ly), 1.0, 1.1,
                                                     crf._C.graph_for(x.device), True)
                torch.cuda.synchronize()
            finally:
                crf._C.set_debug_poison(False)
            kern = crf._C.last_den_kernel()
            assert kern.startswith("crf_fac_pair2_kernel") == bool(pair), kern
            if pair:
                assert ("<false,512,30," in kern or "<true,512,30," in kern) == (geom == "pair2_512"), kern
            outs[pair] = (float(loss.item()), grad.cpu().numpy(), {k: v.cpu().numpy() for k, v in ex.items()})
            del ctx
    (l0, g0, e0), (l1, g1, e1) = outs[0], outs[1]
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    for k in ("costs_alpha", "costs_beta", "costs_ctc"):
        assert np.allclose(e0[k], e1[k], rtol=1e-6, atol=0), k
    assert rel_err(g1, g0) <= (1e-5 if geom == "pair2_512" else 2e-6)   # (pair2_512: another layout sums a row's arcs in another order)
    assert np.allclose(e1["costs_alpha"], e1["costs_beta"], rtol=3e-5, atol=0)
    assert abs(l1 - ref["loss"]) <= TOL * abs(ref["loss"])
    for b in range(B):
        if lx[b] > 0:
            assert rel_err(g1[b], ref["grad"][b]) <= TOL, b


@pytest.mark.parametrize("B", [16, 21, 70])
def test_factored_layout_over_two_cus(crf, tmp_path, B):
    """The factored kernels with TWO compute units per recursion (fac_geom 3; forced here, graphs of 120 k - 240 k arcs take it by
    themselves): with 16 utterances the two CUs of a recursion are 8 block ids apart -- one XCD, plain stores through the shared
    L2 -- with 21 the last five recursions' CUs are neighbours (write-through hand-off), 70 utterances are two launches (every
    workgroup of a launch must be resident: at most CUs / 4 utterances each); ragged lengths incl. an empty and a
    one-frame utterance, against the fp64 oracle.  The per-utterance costs of both directions must agree (logZ from the forward
    vector on CU 0, from the backward rows of both CUs)."""
    g, p = small_synth(tmp_path, 24, 96, 8, 13)
    logits, labels, lx, ly = make_batch(g, B, 47, 24, seed=B, ragged=True)
    lx = np.array(lx); ly = np.array(ly); labels = list(labels)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode="factored_k2")
    with _mode("factored_k2"):
        ctx = crf.CRFContext(p, 0)
        st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
        del ctx
    assert st["fac"] == 1 and st["fac_geom"] in (3, 5)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL
    for b in range(B):
        assert rel_err(grad[b], ref["grad"][b]) <= TOL, b


@pytest.mark.parametrize("ul", [8, 16, 32, 64])
def test_batch_kernels_factored_streams_estimated_graph(crf, tmp_path, ul):
    """Factored streams of the utterance-minor kernels on an ESTIMATED den_lm (long rows, couples beside plain states, states
    the couple detection leaves alone) against the fp64 oracle; the host check of the factored rows on the same graph."""
    from cat_amd import den_lm
    V = 40
    rng = np.random.default_rng(7)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(1500):
        L, sq, a, b = int(rng.integers(8, 30)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    p = str(tmp_path / "den_est.fst")
    den_lm.prep_den_lm(seqs, V, p, 4, 3, 150, selection="count")
    g = fst_io.read_fst(p)
    hh = crf._C.compile_graph_host_only(p)
    chk = crf._C.debug_facbatch_check(hh)
    assert chk["NU"] >= g["S"] // 2 - 2 and chk["fwd_records"] < 0.6 * g["A"] and chk["bwd_records"] < 0.6 * g["A"]
    B, T = 5, 60
    logits = rng.normal(size=(B, T, V)).astype(np.float32) * 2.0
    logits = logits - np.log(np.exp(logits).sum(-1, keepdims=True))
    lx = np.array([60, 47, 33, 60, 1], dtype=np.int32)
    labels, ly = [], []
    for b in range(B):
        lab = seqs[b][:max(1, int(lx[b]) // 6)]
        labels += lab; ly.append(len(lab))
    ref = oracle.ctc_crf(g, logits, np.array(labels, dtype=np.int32), lx, np.array(ly, dtype=np.int32), lamb=0.1)
    with _env(CRF_BAT_UL=ul):
        loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode="batch")
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_fused_log_softmax(crf, tmp_path, mode, dtype):
    """``CTC_CRF_LOSS(fuse_log_softmax=True)(netout, ...)`` against the unfused path behind torch's own
    log_softmax (cat/ctc/train.py:174-186), loss and d loss / d netout, for fp32 / bf16 / fp16 network outputs (the
    unfused reference sees the same rounded values, upcast), ragged lengths, one invalid utterance (L + repeats > T)."""
    g, fst = small_synth(tmp_path, 24, 96, 8, 3)
    _, labels, lx, ly = make_batch(g, 4, 48, 24, seed=11, ragged=True)
    lx = np.array(lx); ly = np.array(ly); labels = list(labels)
    ly0 = int(ly[0])
    lx[0] = max(1, ly0 - 1)                                   # utterance 0: more labels than frames -> invalid numerator
    rng = np.random.default_rng(5)
    raw = torch.tensor(rng.normal(size=(4, 48, 24)) * 3.0, dtype=torch.float32).to(getattr(torch, dtype))
    lab_t, lx_t, ly_t = (torch.tensor(a, dtype=torch.int32) for a in (labels, lx, ly))
    with _mode(mode):
        ctx = crf.CRFContext(fst, 0)
        xf = raw.cuda().requires_grad_(True)
        lf = crf.CTC_CRF_LOSS(lamb=0.1, fuse_log_softmax=True)(xf, lab_t, lx_t, ly_t)
        lf.backward()
        xu = raw.float().cuda().requires_grad_(True)
        lu = crf.CTC_CRF_LOSS(lamb=0.1)(torch.log_softmax(xu, -1), lab_t, lx_t, ly_t)
        lu.backward()
    assert xf.grad.dtype == raw.dtype
    assert abs(lf.item() - lu.item()) <= TOL * abs(lu.item())
    tol = TOL if dtype == "float32" else 1e-2                 # the fused gradient is rounded to the input's dtype at the end
    assert rel_err(xf.grad.float().cpu().numpy(), xu.grad.cpu().numpy()) <= tol
    # rows sum to zero (softmax Jacobian) and are zero beyond the utterance's length
    gsum = xf.grad.float().sum(-1).abs().max().item()
    assert gsum <= (1e-5 if dtype == "float32" else 2e-2)
    for b in range(4):
        assert xf.grad[b, int(lx[b]):].abs().max().item() == 0.0 if int(lx[b]) < 48 else True
    f2 = crf.ctc_crf_loss(xf.detach(), lab_t, lx_t, ly_t, fst, fuse_log_softmax=True)
    assert abs(f2.item() - lf.item()) <= 1e-6 * abs(lf.item())
    del ctx


@pytest.mark.parametrize("mode,V", [("factored", 40), ("factored_rc", 40), ("factored_k2", 40), ("resident", 40), ("streaming", 40), ("factored", 150), ("factored_rc", 150), ("factored_k2", 150), ("factored_pair2", 40), ("factored_pair2", 150)])
def test_estimated_den_lm_with_long_rows(crf, tmp_path, mode, V):
    """A den_lm ESTIMATED from text (cat_amd.den_lm.prep_den_lm) has the in-degree profile of a real
    n-gram LM: the low-order history states are entered from hundreds of states.  The factored layout cuts such rows
    into pieces on adjacent lanes (butterfly sum in the kernel); all kernel families against the fp64 oracle."""
    from cat_amd import den_lm
    rng = np.random.default_rng(7)                            # (V = 150: states with more than 80 OUT-arcs too: backward rows in pieces)
    trans = rng.dirichlet(np.ones(V - 1) * (0.05 if V < 100 else 1.0), size=(V, V))
    seqs = []
    for _ in range(1500):
        L, sq, a, b = int(rng.integers(8, 30)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    p = str(tmp_path / "den_est.fst")
    den_lm.prep_den_lm(seqs, V, p, 4, 3, 150, selection="count")
    g = fst_io.read_fst(p)
    indeg = np.bincount(g["dst"], minlength=g["S"]).max()
    assert indeg > 100                                        # longer than one lane's 80 arcs
    B, T = 3, 60
    logits = rng.normal(size=(B, T, V)).astype(np.float32) * 2.0
    logits = logits - np.log(np.exp(logits).sum(-1, keepdims=True))
    lx = np.array([60, 47, 33], dtype=np.int32)
    labels, ly = [], []
    for b in range(B):
        lab = seqs[b][:max(1, int(lx[b]) // 6)]
        labels += lab; ly.append(len(lab))
    ref = oracle.ctc_crf(g, logits, np.array(labels, dtype=np.int32), lx, np.array(ly, dtype=np.int32), lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode=mode)
    if mode.startswith("factored"):
        with _mode(mode):
            ctx = crf.CRFContext(p, 0)
        st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
        # rows that are still longer than a lane's 80 arcs after the factorisation: the LDS-table variant by default
        assert st["fac"] == 1 and (st["fac_geom"] == 0 if mode == "factored_rc" else st["fac_geom"] == 3 if mode == "factored_k2" else st["fac_geom"] in (0, 1, 4))
        del ctx
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


@pytest.mark.parametrize("V,nsent", [(40, 1500), (72, 6000)])
def test_estimated_den_lm_default_rule(crf, tmp_path, V, nsent):
    """A den_lm made by the DEFAULT rule of `prep_den_lm` since round 6 -- Kaldi's published selection by log-likelihood, every seen history of
    fewer than --no-prune-ngram-order tokens kept (tests/test_den_lm_tools.py::test_estimator_selects_states_by_likelihood_like_kaldi) -- i.e.
    with the shape real den_lm files have: about one LM state per seen bigram, in-degrees of tens instead of hundreds.  On the kernel family
    the graph takes BY ITSELF, against the fp64 oracle."""
    from cat_amd import den_lm
    rng = np.random.default_rng(17)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(nsent):
        L, sq, a, b = int(rng.integers(8, 30)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    p = str(tmp_path / "den_kaldi_rule.fst")
    den_lm.prep_den_lm(seqs, V, p, 4, 3, 250)
    g = fst_io.read_fst(p)
    B, T = 4, 72
    logits = rng.normal(size=(B, T, V)).astype(np.float32) * 2.0
    logits = logits - np.log(np.exp(logits).sum(-1, keepdims=True))
    lx = np.array([72, 60, 47, 33], dtype=np.int32)
    labels, ly = [], []
    for b in range(B):
        lab = seqs[b][:max(1, int(lx[b]) // 6)]
        labels += lab; ly.append(len(lab))
    ref = oracle.ctc_crf(g, logits, np.array(labels, dtype=np.int32), lx, np.array(ly, dtype=np.int32), lamb=0.1)
    ctx = crf.CRFContext(p, 0)
    x = torch.tensor(logits, device="cuda:0", requires_grad=True)
    loss = crf.CTC_CRF_LOSS(lamb=0.1)(x, torch.tensor(labels, dtype=torch.int32), torch.tensor(lx), torch.tensor(ly, dtype=torch.int32))
    loss.backward()
    st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
    print(f"V={V}: S={g['S']} A={len(g['src'])} max in-degree {np.bincount(g['dst'], minlength=g['S']).max()} kernel {crf._C.last_den_kernel()} fac={st['fac']} geom={st.get('fac_geom')}")
    del ctx
    assert abs(loss.item() - ref["loss"]) <= TOL * abs(ref["loss"])
    for b in range(B):
        assert rel_err(x.grad[b].cpu().numpy(), ref["grad"][b]) <= TOL, b


def test_estimated_den_lm_with_many_rows(crf, tmp_path):
    """The den_lm of tools/bench_fst.py's largest point (40 000 sentences, 72 tokens: S = 6 836, 100 k arcs, in-degree up to
    970): 71 slices of forward rows -- more than the three per wave whose constants fit registers -- so the compiler
    picks, by itself, the 768-thread kernels with the row constants in an LDS table (fac_geom 1; round 1 and most of round 2
    ran this graph on the 512-thread geometry).  Ragged batch against the fp64 oracle."""
    from cat_amd import den_lm
    V = 72
    rng = np.random.default_rng(0)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(40000):
        L, sq, a, b = int(rng.integers(10, 40)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    p = str(tmp_path / "den_est_big.fst")
    den_lm.prep_den_lm(seqs, V, p, 4, 3, 2000, selection="count")
    g = fst_io.read_fst(p)
    assert g["S"] > 6000
    B, T = 4, 48
    logits = rng.normal(size=(B, T, V)).astype(np.float32) * 2.0
    logits = logits - np.log(np.exp(logits).sum(-1, keepdims=True))
    lx = np.array([48, 48, 31, 2], dtype=np.int32)
    labels, ly = [], []
    for b in range(B):
        lab = seqs[b][:max(1, int(lx[b]) // 6)]
        labels += lab; ly.append(len(lab))
    ref = oracle.ctc_crf(g, logits, np.array(labels, dtype=np.int32), lx, np.array(ly, dtype=np.int32), lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1)
    ctx = crf.CRFContext(p, 0)
    st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
    del ctx
    assert st["fac"] == 1 and st["fac_geom"] in (1, 4)            # a table geometry: 1024 threads, or 768 where that does not take the graph
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


@pytest.mark.parametrize("push", [False, True])
def test_rewritten_den_lm_vs_oracle(crf, tmp_path, push):
    """A den_lm as OpenFst tools leave it (cat/utils/tool/prep_den_lm.sh:48-49): states re-numbered, arcs of a state in
    another order, optionally weights pushed along the arcs.  Same loss and gradient as the ORIGINAL graph's oracle (the
    rewrites preserve every path weight; pushing only up to fp32 rounding of the new weights), and the rewritten graph
    keeps the factored layout (a pushed one after the compiler has re-gauged it, fst_graph.cpp regauge_pushed)."""
    from tests.util import transform_graph
    g, p = small_synth(tmp_path, 24, 96, 8, 13)
    q = os.path.join(str(tmp_path), "rewritten.fst")
    g2 = transform_graph(g, q, seed=3, renumber=True, reorder=True, push=push)
    B, T, V = 4, 60, 24
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=6, ragged=True)
    ref0 = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    ref = oracle.ctc_crf(g2, logits, labels, lx, ly, lamb=0.1)
    assert abs(ref["loss"] - ref0["loss"]) <= (1e-4 if push else 1e-6) * abs(ref0["loss"])   # the rewrite itself
    loss, grad = run_hip(crf, q, logits, labels, lx, ly, lamb=0.1)
    with _mode("factored"):
        ctx = crf.CRFContext(q, 0)
    st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
    assert st["fac"] == 1 and st["regauged"] == (1 if push else 0)   # (pushed: the compiler's own potentials bring the structure back)
    del ctx
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


def test_wide_rows_grad_kernel(crf, tmp_path):
    """More than 2 560 forward rows: the per-frame rows (Q, BP) exceed 5 120 floats and the denominator half of the
    grad pass runs with 512-thread workgroups; factored layout, against the fp64 oracle (short T, ragged)."""
    g, p = small_synth(tmp_path, 48, 3000, 8, 21)
    B, T, V = 3, 40, 48
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=4, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    with _mode("factored"):
        ctx = crf.CRFContext(p, 0)
    st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
    assert st["fac"] == 1 and st["fac_fwd_slots"] > 0
    del ctx
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode="factored")
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


def test_factored_schedules(crf, tmp_path):
    """The factored den kernels run everything else BESIDE them while 2B workgroups leave half of the CUs free
    (B = 3 here and in the tests above) and fall back to 'numerator after denominator' for larger batches
    (B = 160 > CUs/4 on any current part); den-only calls (gpu_den) take a third path."""
    g, p = small_synth(tmp_path, 12, 40, 6, 7)
    B, T, V = 160, 24, 12
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=11, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode="factored")
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL
    with _mode("factored"):
        ctx = crf.CRFContext(p, 0)
    assert crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))["fac"] == 1
    lg = torch.tensor(logits[:5], device="cuda:0")
    gd, ca, cb = torch.zeros_like(lg), torch.zeros(5, device="cuda:0"), torch.zeros(5, device="cuda:0")
    crf._C.gpu_den(lg, gd, torch.tensor(lx[:5]).cuda(), ca, cb)
    den = oracle.den(fst_io.read_fst(p), logits[:5], lx[:5])
    assert np.allclose(ca.cpu().numpy(), np.asarray(den[1]).ravel(), rtol=TOL, atol=0)
    assert np.allclose(cb.cpu().numpy(), np.asarray(den[2]).ravel(), rtol=TOL, atol=0)
    assert post_err(gd.cpu().numpy(), np.asarray(den[0])) <= TOL
    del ctx


@pytest.mark.parametrize("mode", MODES)
def test_edge_cases(crf, tmp_path, mode):
    """repeats, L = 0, L + repeats == T (no slack), lx < T, and an invalid utterance (L + repeats > T)."""
    g, p = small_synth(tmp_path, 6, 8, 3, 3)
    rng = np.random.default_rng(5)
    B, T, V = 5, 9, 6
    x = rng.normal(size=(B, T, V)) * 2
    logits = (x - np.log(np.exp(x).sum(-1, keepdims=True))).astype(np.float32)
    labs = [[1, 1, 2], [], [3, 3, 3, 3, 3], [2, 4, 2, 4], [1, 1, 1, 1, 1, 1]]
    lx = np.array([9, 7, 9, 4, 9], dtype=np.int32)  # utt 2: L+rep = 9 == T; utt 4: 6+5 = 11 > 9 invalid
    ly = np.array([len(l) for l in labs], dtype=np.int32)
    labels = np.array([v for l in labs for v in l], dtype=np.int32)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.3, size_average=False)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.3, size_average=False, mode=mode)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


def _peaked(B, T, V, scale, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(B, T, V)) * scale
    x = x - x.max(-1, keepdims=True)
    return (x - np.log(np.exp(x).sum(-1, keepdims=True))).astype(np.float32)


def test_peaked_inputs(crf, tmp_path):
    """Near one-hot posteriors (log-probs down to about -100 below the row max): stresses the
    linear-domain rescaling that replaces the reference's per-arc log-add (den_calculate.cu:29-35)."""
    g, p = small_synth(tmp_path, 10, 24, 5, 7)
    B, T, V = 3, 50, 10
    logits = _peaked(B, T, V, 20.0, 11)
    _, labels, lx, ly = make_batch(g, B, T, V, seed=3, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1)
    assert np.isfinite(loss)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


@pytest.mark.parametrize("mode", MODES)
def test_robust_fallback_forced(crf, tmp_path, mode):
    """CRF_ROBUST=1: every utterance is redone by the robust (per-frame log-shifted) denominator kernels after the fast
    ones; ordinary inputs, all three fast families in front, ragged lengths, against the fp64 oracle."""
    g, p = small_synth(tmp_path, 12, 40, 6, 5)
    B, T, V = 4, 37, 12
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=8, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    with _env(CRF_ROBUST=1):
        loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode=mode)
        with _mode(mode):
            ctx = crf.CRFContext(p, 0)
            x = torch.tensor(logits, device="cuda:0")
            _, gd, ex = crf._C.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, crf._C.graph_for(x.device), True)
            del ctx
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL
    den = oracle.den(fst_io.read_fst(p), logits, lx)
    assert np.allclose(ex["costs_alpha"].cpu().numpy(), np.asarray(den[1]).ravel(), rtol=TOL, atol=0)
    assert np.allclose(ex["costs_beta"].cpu().numpy(), np.asarray(den[2]).ravel(), rtol=TOL, atol=0)
    assert post_err(gd.cpu().numpy(), np.asarray(den[0])) <= TOL


@pytest.mark.parametrize("nats,mode", [(10.0, "factored"), (25.0, "factored"), (45.0, "factored"), (80.0, "factored"), (120.0, "factored")] + [(110.0, m) for m in MODES])
def test_single_frame_shrink_window(crf, tmp_path, nats, mode):
    """A label the den_lm forbids `nats` above everything else, in every frame of utterance 0 and in a stretch of utterance 1: every
    frame shrinks the recursions' vectors by e^-nats.  Up to ~131 nats the scaled-fp32 recursions carry that themselves -- each frame is
    rescaled from the maximum of its OWN source vector, so the rows the grad pass multiplies always sit at 2^20 -- and no utterance
    may take the fallback (`crf_last_fallback_counts`).  A build with the lagged scale (CRF_X_LAG=1: the scale of frame t+1 chosen before
    the size of its vector is known; measured and not adopted) lets the rows carry the frame's growth and hands such utterances
    to the log-shifted fallback from 28 nats on -- its first thresholds gave NaN gradients at 45 nats here.  Either way the result is
    the fp64 oracle's (the reference's log-domain arithmetic, den_calculate.cu:29-35, has no such window).
    110 nats, every kernel family: what this test FOUND in round 5 -- the streaming grad kernels took 2^-64 out of e' * sum(q * b) before
    normalising the frame, and between ~100 and 131 nats that product left the fp32 range although recursions, loss and flags were fine:
    the gradient of rounds 1 - 4 was 90 % off there, silently (kGradDescale, crf_kernels.hip)."""
    g, p = small_synth(tmp_path, 9, 24, 5, 7)             # tokens 1..8; label 9 exists only in the network output
    B, T, V = 3, 40, 10
    rng = np.random.default_rng(22)
    x = rng.normal(size=(B, T, V)) * 2.0
    x[0, :, 9] += nats
    x[1, 10:25, 9] += nats
    m = x.max(-1, keepdims=True)
    logits = (x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))).astype(np.float32)
    _, labels, lx, ly = make_batch(g, B, T, 9, seed=3, ragged=True)
    lx[:] = [40, 36, 31]
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode=mode)
    nden, nnum = crf._C.last_fallback_counts(torch.cuda.current_stream().cuda_stream)
    assert np.isfinite(loss) and np.isfinite(grad).all()
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    for b in range(B):
        assert rel_err(grad[b], ref["grad"][b]) <= TOL
    assert nnum == 0
    if not crf._C.build_switches().get("LAG") or not mode.startswith("factored") or "k2" in mode or "pair2" in mode:
        assert nden == 0 or nats >= 120.0, (nats, nden)     # (120 nats +- the inputs' spread: some frames beyond what e' can hold -- the fallback may take them)
    else:
        assert nden == (2 if nats >= 45.0 else 0) or nats == 25.0, (nats, nden)   # (25 nats: at the lagged rule's threshold)


@pytest.mark.parametrize("mode", MODES)
def test_underflow_fallback_forbidden_argmax(crf, tmp_path, mode):
    """The network is certain (300 nats) of a label the den_lm does not contain, in every frame of utterance 0, in a
    stretch of frames of utterance 1, never in utterance 2: every live (state, label) of such a frame lies 300 nats
    below the row maximum, the scaled-fp32 recursions lose all their mass there, and the reference's log-domain
    arithmetic (den_calculate.cu:29-35) does not.  The fast kernels flag such utterances and the robust kernels redo
    them: finite, and within 1e-4 of the fp64 oracle -- never -inf / NaN where the oracle is finite."""
    g, p = small_synth(tmp_path, 9, 24, 5, 7)             # tokens 1..8; label 9 exists only in the network output
    B, T, V = 3, 40, 10
    rng = np.random.default_rng(21)
    x = rng.normal(size=(B, T, V)) * 2.0
    x[0, :, 9] += 300.0
    x[1, 10:25, 9] += 300.0
    m = x.max(-1, keepdims=True)
    logits = (x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))).astype(np.float32)
    _, labels, lx, ly = make_batch(g, B, T, 9, seed=3, ragged=True)
    lx[:] = [40, 36, 31]
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    assert np.isfinite(ref["loss"]) and ref["costs_den"].min() < -2000
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode=mode)
    assert np.isfinite(loss) and np.isfinite(grad).all()
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL
    for b in range(B):
        assert rel_err(grad[b], ref["grad"][b]) <= TOL


@pytest.mark.parametrize("mode", ["factored", "resident", "batch"])
@pytest.mark.parametrize("fused", [False, True])
def test_numerator_fallback_forced(crf, tmp_path, mode, fused):
    """robust_ctc = 1: EVERY utterance's numerator is redone by the log-domain kernels (crf_robust_ctc_kernel + the fix pass)
    behind the grad pass, which then treats it as absent; ordinary inputs, ragged lengths, an empty label sequence and an invalid
    one (L + repeats > T) riding along; loss, per-utterance costs and the gradient against the fp64 oracle -- also behind the fused
    log_softmax, whose softmax term depends on whether the numerator counts."""
    g, p = small_synth(tmp_path, 12, 40, 6, 5)
    B, T, V = 5, 37, 12
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=8, ragged=True)
    lab = [list(labels[sum(ly[:i]):sum(ly[:i + 1])]) for i in range(B)]
    lab[3] = []                                            # no labels: the all-blank path
    lab[4] = [3] * 25                                      # 25 labels + 24 repeats > lx: invalid (gpu_ctc.h:166-174)
    ly = np.array([len(x) for x in lab], dtype=np.int32)
    labels = np.array([v for x in lab for v in x], dtype=np.int32)
    assert ly[4] * 2 - 1 > lx[4]
    with _env(CRF_ROBUST_CTC=1), _mode(mode):
        ctx = crf.CRFContext(p, 0)
        if fused:
            raw = torch.tensor(np.random.default_rng(3).normal(size=(B, T, V)) * 2.0, dtype=torch.float32)
            logp = raw.log_softmax(-1).numpy()
            x = raw.cuda().requires_grad_(True)
        else:
            logp = logits
            x = torch.tensor(logits, device="cuda:0", requires_grad=True)
        loss = crf.CTC_CRF_LOSS(lamb=0.1, fuse_log_softmax=fused)(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly))
        loss.backward()
        got, grad = float(loss.item()), x.grad.cpu().numpy()
        _, gc, ex = crf._C.loss_fwd_bwd(torch.tensor(logp, device="cuda:0"), torch.tensor(labels), torch.tensor(lx), torch.tensor(ly),
                                        0.0, -1.0, None, True)
        del ctx
    ref = oracle.ctc_crf(fst_io.read_fst(p), logp, labels, lx, ly, lamb=0.1)
    gref, cref, valid = oracle.ctc(logp, labels, lx, ly)
    assert list(valid) == [1, 1, 1, 1, 0] and list(ex["invalid"].cpu().numpy()) == [0, 0, 0, 0, 1]
    assert np.allclose(ex["costs_ctc"].cpu().numpy()[:4], cref[:4], rtol=TOL, atol=0)
    assert rel_err(gc.cpu().numpy(), gref) <= TOL
    assert abs(got - ref["loss"]) <= TOL * abs(ref["loss"])
    gexp = ref["grad"].astype(np.float64)
    if fused:                                              # chain rule of log_softmax
        gexp = gexp - np.exp(logp.astype(np.float64)) * gexp.sum(-1, keepdims=True)
    assert rel_err(grad, gexp) <= TOL


def test_numerator_fallback_on_the_third_stream(crf, tmp_path):
    """The staged schedule moves the log-domain numerator chains to the context's third stream -- beside the grad stages instead of
    in front of them, the marked frames' posteriors subtracted behind the last stage -- once a recent call of the context needed
    them (a pinned host word the chains write, read without a sync).  Four calls with every utterance marked (robust_ctc = 1): the
    first runs them in front of the stages, the following ones on the third stream; each against the fp64 oracle; then the switch
    `aux_stream` 0 / 1 forces either form."""
    g, p = small_synth(tmp_path, 12, 40, 6, 5)
    B, T, V = 6, 96, 12
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=11, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    core = crf._C

    def call():
        x = torch.tensor(logits, device="cuda:0", requires_grad=True)
        loss = crf.CTC_CRF_LOSS(lamb=0.1)(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly))
        streams = core.last_call_streams()
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.item()) - ref["loss"]) <= TOL * abs(ref["loss"])
        assert rel_err(x.grad.cpu().numpy(), ref["grad"]) <= TOL
        return streams

    with _env(CRF_ROBUST_CTC=1), _mode("factored"):
        ctx = crf.CRFContext(p, 0)
        seen = [call() for _ in range(4)]
        if seen[0] >= 2:                                       # (a context without a side stream runs everything in order)
            assert seen[0] in (2, 3) and seen[-1] == 3, seen   # (earlier calls of this process may have marked utterances already)
            with _env(CRF_AUX_STREAM=0):
                assert call() == 2
            with _env(CRF_AUX_STREAM=1):
                assert call() == 3
        del ctx
    with _mode("factored"):                                    # nothing marked, nothing remembered beyond 16 calls: two streams
        ctx = crf.CRFContext(p, 0)
        seen = [call() for _ in range(18)]
        assert seen[-1] <= 2, seen
        del ctx


@pytest.mark.parametrize("switches", [{}, {"CRF_ROBUST": 0}, {"CRF_CTC_TILT": 0}], ids=["default", "tilt_alone", "fallback_alone"])
@pytest.mark.parametrize("T,L,sigma", [(2400, 400, 2.0), (3000, 500, 2.0), (3000, 500, 1.0), (3000, 60, 1.0)])
def test_numerator_long_utterances_with_many_labels(crf, T, L, sigma, switches):
    """T = 3000 frames, L = 500 labels (BASELINE config #5's utterance shape; ly = lx // 6 as bench.py draws them) on inputs that do
    not follow the labels: in the middle of such an utterance the posterior mass lies hundreds of nats below (max alpha)(max beta) --
    beyond what the chains' per-frame rescaling keeps in fp64 (round 3 finding: 0 * inf = NaN gradients from frame ~800 on).  Two
    devices, each sufficient here: the chains are TILTED (ctc_rho: the free mass of either chain moves at the speed the labels need;
    `tilt_alone` = fallback kernels switched off: no frame may be marked), and frames that still leave the range are marked by the grad
    pass and redone by the log-domain kernels (`fallback_alone` = plain chains, as before the tilt).  A short utterance rides along."""
    rng = np.random.default_rng(T)
    V = 72
    x = torch.tensor(rng.normal(size=(2, T, V)) * sigma, dtype=torch.float32).log_softmax(-1)
    ly = np.array([L, 40], dtype=np.int32)
    lx = np.array([T, 300], dtype=np.int32)
    labels = rng.integers(1, V, size=int(ly.sum())).astype(np.int32)
    gref, cref, valid = oracle.ctc(x.numpy(), labels, lx, ly)
    assert valid.all() and np.isfinite(gref).all()
    with _env(**switches):
        _, g, ex = crf._C.loss_fwd_bwd(x.cuda(), torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 0.0, -1.0, None, True)
    g = g.cpu().numpy()
    assert np.isfinite(g).all()
    assert np.allclose(ex["costs_ctc"].cpu().numpy(), cref, rtol=TOL, atol=0) and int(ex["invalid"].sum().item()) == 0
    for b in range(2):
        assert rel_err(g[b], gref[b]) <= TOL, b
        assert np.allclose(g[b, :lx[b]].sum(-1), 1.0, atol=1e-4)      # every frame's posteriors sum to one


@pytest.mark.parametrize("boost", [6.0, 3.0, 1.5])
def test_numerator_tilt_with_an_off_diagonal_alignment(crf, boost):
    """The tilt of the numerator chains assumes nothing about the inputs -- any rho is exact -- but it spends range where the
    alignment leaves the straight line from (0, 0) to (T, 2L + 1).  Here every label sits in the first 40 % of a long utterance
    (then blanks to the end: up to ~450 states off the diagonal) and the network output favours that alignment by `boost` nats over
    N(0, 1) noise: strongly (weak tilt), moderately, barely (nearly the full tilt against a far-off alignment: what does not fit the
    range is marked and redone in the log domain).  Loss term and gradient against the fp64 oracle."""
    rng = np.random.default_rng(int(boost * 10))
    T, L, V = 2400, 400, 72
    labels = rng.integers(1, V, size=L).astype(np.int32)
    ali = np.zeros(T, dtype=np.int64)                       # label, blank, label, blank, ... from frame 0; blanks after frame 2L
    ali[0:2 * L:2] = labels
    raw = rng.normal(size=(1, T, V))
    raw[0, np.arange(T), ali] += boost
    x = torch.tensor(raw, dtype=torch.float32).log_softmax(-1)
    lx, ly = np.array([T], dtype=np.int32), np.array([L], dtype=np.int32)
    gref, cref, valid = oracle.ctc(x.numpy(), labels, lx, ly)
    assert valid.all() and np.isfinite(gref).all()
    _, g, ex = crf._C.loss_fwd_bwd(x.cuda(), torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 0.0, -1.0, None, True)
    g = g.cpu().numpy()
    assert np.isfinite(g).all() and int(ex["invalid"].sum().item()) == 0
    assert np.allclose(ex["costs_ctc"].cpu().numpy(), cref, rtol=TOL, atol=0)
    assert rel_err(g[0], gref[0]) <= TOL
    assert np.allclose(g[0].sum(-1), 1.0, atol=1e-4)


def test_numerator_extreme_range(crf):
    """Forced alignments through labels ~400 nats below the row max: the numerator runs in fp64
    (range e^+-700) exactly so that this matches the log-domain reference semantics."""
    B, T, V = 3, 40, 9
    logits = _peaked(B, T, V, 80.0, 5)
    rng = np.random.default_rng(6)
    ly = np.array([4, 7, 0], dtype=np.int32)
    lx = np.array([40, 33, 21], dtype=np.int32)
    labels = rng.integers(1, V, size=int(ly.sum())).astype(np.int32)
    gref, cref, valid = oracle.ctc(logits, labels, lx, ly)
    assert valid.all() and cref.min() < -300
    core = crf._C
    _, g, ex = core.loss_fwd_bwd(torch.tensor(logits, device="cuda:0"), torch.tensor(labels), torch.tensor(lx),
                                 torch.tensor(ly), 0.0, -1.0, None, True)
    assert np.allclose(ex["costs_ctc"].cpu().numpy(), cref, rtol=TOL, atol=0)
    assert int(ex["invalid"].sum().item()) == 0
    assert rel_err(g.cpu().numpy(), gref) <= TOL


def test_warp_ctc_loss_vs_torch(crf):
    """WARP_CTC_LOSS (reference __init__.py:128-144) against torch's own CPU ctc_loss."""
    rng = np.random.default_rng(2)
    B, T, V = 4, 30, 11
    x = torch.tensor(rng.normal(size=(B, T, V)), dtype=torch.float32).log_softmax(-1)
    ly = torch.tensor([5, 0, 9, 3], dtype=torch.int32)
    lx = torch.tensor([30, 12, 25, 30], dtype=torch.int32)
    labels = torch.tensor(rng.integers(1, V, size=int(ly.sum())), dtype=torch.int32)
    xg = x.cuda().requires_grad_(True)
    loss = crf.WARP_CTC_LOSS(size_average=False)(xg, labels, lx, ly)
    loss.backward()
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(xr.transpose(0, 1), labels.long(), lx.long(), ly.long(), blank=0, reduction="sum")
    ref.backward()
    assert abs(loss.item() - ref.item()) <= TOL * abs(ref.item())
    # torch returns d/d(logits-before-log_softmax) = softmax - gamma ; ours is -gamma on the log-probs
    mask = (torch.arange(T)[None, :] < lx[:, None]).float()[..., None]
    gamma_ref = (x.exp() * mask - xr.grad)
    assert rel_err(-xg.grad.cpu().numpy(), gamma_ref.numpy()) <= 5e-4


@pytest.mark.parametrize("L,T", [(200, 500), (300, 700), (700, 1500), (1400, 3000), (2047, 4200)])
def test_ctc_label_length_variants(crf, L, T):
    """The numerator kernels are specialised on ctc states per thread (1, 2, 4, 8 <- ceil((2L+1)/512)):
    every variant against torch's CPU ctc_loss, with one short utterance riding along in the same batch."""
    rng = np.random.default_rng(L)
    V = 37
    x = torch.tensor(rng.normal(size=(2, T, V)) * 1.5, dtype=torch.float32).log_softmax(-1)
    ly = torch.tensor([L, 7], dtype=torch.int32)
    lx = torch.tensor([T, T // 3], dtype=torch.int32)
    labels = torch.tensor(rng.integers(1, V, size=int(ly.sum())), dtype=torch.int32)
    xg = x.cuda().requires_grad_(True)
    loss = crf.WARP_CTC_LOSS(size_average=False)(xg, labels, lx, ly)
    loss.backward()
    xr = x.clone().double().requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(xr.transpose(0, 1), labels.long(), lx.long(), ly.long(), blank=0, reduction="sum")
    ref.backward()
    assert abs(loss.item() - ref.item()) <= TOL * abs(ref.item())
    mask = (torch.arange(T)[None, :] < lx[:, None]).double()[..., None]
    gamma_ref = (x.double().exp() * mask - xr.grad)
    assert rel_err(-xg.grad.cpu().numpy(), gamma_ref.numpy()) <= 5e-4


def _default_graph(tmp_path_factory):
    d = tmp_path_factory.mktemp("denlm")
    p = os.path.join(str(d), "den_lm_v72.fst")
    from cat_amd.den_lm import synth_den_lm
    return synth_den_lm(72, 2048, 24, 0, path=p), p


@pytest.fixture(scope="module")
def default_graph(tmp_path_factory):
    return _default_graph(tmp_path_factory)


@pytest.mark.parametrize("mode", MODES)
def test_config2_slice_vs_oracle(crf, default_graph, mode):
    """BASELINE config #2 graph and shapes (V=72, S=4097, T=500), on a 3-utterance slice the fp64
    oracle finishes in seconds.  In resident mode this graph runs on K=2 CUs per recursion, i.e. it
    exercises the per-frame all-gather through L2."""
    g, p = default_graph
    logits, labels, lx, ly = make_batch(g, 3, 500, 72, seed=0, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode=mode)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    e = rel_err(grad, ref["grad"])
    print(f"config-2 slice: loss rel err {abs(loss - ref['loss']) / abs(ref['loss']):.2e}, grad err {e:.2e}")
    assert e <= TOL
    # the two posterior matrices separately, entry by entry
    core = crf._C
    with _mode(mode):
        ctx = crf.CRFContext(p, 0)
        x = torch.tensor(logits, device="cuda:0")
        _, gd, _ = core.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, core.graph_for(x.device), True)
        _, gc, _ = core.loss_fwd_bwd(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 0.0, -1.0, None, True)
    gdo, _, _ = oracle.den(fst_io.read_fst(p), logits, lx)
    gco, _, _ = oracle.ctc(logits, labels, lx, ly)
    pe_d, pe_c = post_err(gd.cpu().numpy(), gdo), post_err(gc.cpu().numpy(), gco)
    print(f"posterior entry-wise rel err: den {pe_d:.2e}, ctc {pe_c:.2e}")
    assert pe_d <= TOL and pe_c <= TOL
    del ctx


def test_full_size_invariants(crf, default_graph):
    """B=32, T=500, V=72 (config #2, full) -- size-independent properties:
    both posterior matrices sum to 1 per frame for t < lx and are 0 after; logZ from the forward
    and from the backward recursion agree; per-utterance results do not depend on batch position."""
    g, p = default_graph
    B, T, V = 32, 500, 72
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=1, ragged=True)
    core = crf._C
    ctx = crf.CRFContext(p, 0)
    x = torch.tensor(logits, device="cuda:0")
    _, gden, ex = core.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, core.graph_for(x.device), True)
    gden = gden.cpu().numpy()
    ca, cb = ex["costs_alpha"].cpu().numpy(), ex["costs_beta"].cpu().numpy()
    assert np.allclose(ca, cb, rtol=2e-5, atol=0)
    _, gctc, ex2 = core.loss_fwd_bwd(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly), 0.0, -1.0, None, True)
    gctc = gctc.cpu().numpy()
    assert int(ex2["invalid"].sum().item()) == 0
    for b in range(B):
        n = int(lx[b])
        assert np.allclose(gden[b, :n].sum(-1), 1.0, atol=2e-4)
        assert np.allclose(gctc[b, :n].sum(-1), 1.0, atol=2e-4)
        assert np.all(gden[b, n:] == 0) and np.all(gctc[b, n:] == 0)
    assert gden.min() >= 0 and gctc.min() >= 0
    # permutation invariance: utterance 5 alone == utterance 5 inside the batch
    _, g5, ex5 = core.loss_fwd_bwd(x[5:6].contiguous(), None, torch.tensor(lx[5:6]), None, 1.0, 0.0, core.graph_for(x.device), True)
    assert np.allclose(g5.cpu().numpy()[0], gden[5], rtol=1e-5, atol=1e-7)
    del ctx


@pytest.mark.parametrize("path", ["batch", "streaming"])
def test_large_graph_global_vectors(crf, tmp_path, path):
    """A den_lm whose state vectors exceed a CU's LDS (S = 20 001 states, ~200 k arcs): no register-resident layout
    applies.  Default: the utterance-minor kernels (state vectors [state][utterance] in global memory, one launch per
    frame); CRF_NO_BATCH=1: the persistent streaming kernels with their vectors in L2."""
    from cat_amd.den_lm import synth_den_lm
    p = os.path.join(str(tmp_path), "big.fst")
    g = synth_den_lm(72, 10000, 8, seed=3, path=p)
    logits, labels, lx, ly = make_batch(g, 2, 40, 72, seed=9, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    with _env(CRF_NO_BATCH=1 if path == "streaming" else 0):
        loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1)
    st = crf._C.compile_graph_host_only(p)
    assert crf._C.graph_stats(st)["S"] == 20001
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


def test_resident_layout_beyond_8k_states(crf, tmp_path):
    """A graph with 8 k < S <= 16 k states on the generic register-resident layout (K CUs per recursion exchanging the state
    vector every frame): its gather vector needs the 64 KiB state-vector buffers (16-bit LDS byte offsets: <= 16 384 entries)."""
    g, p = small_synth(tmp_path, 72, 4500, 8, 2)
    assert 8192 < g["S"] <= 16384
    B, T, V = 3, 40, 72
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=2, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    with _mode("resident"):
        ctx = crf.CRFContext(p, 0)
        h = crf._C.graph_for(torch.device("cuda", 0))
        st = crf._C.graph_stats(h)
        assert st["res_K"] >= 1 and crf._C.den_kernels(h, B, T, V) == "resident", st
        del ctx
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1, mode="resident")
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


def test_large_vocab(crf, tmp_path):
    """V = 5000 output units (BASELINE config #5's vocabulary) on a small BPE-like graph."""
    from cat_amd.den_lm import synth_den_lm
    p = os.path.join(str(tmp_path), "v5000.fst")
    g = synth_den_lm(5000, 5000, 3, seed=2, path=p)
    logits, labels, lx, ly = make_batch(g, 2, 30, 5000, seed=4, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    loss, grad = run_hip(crf, p, logits, labels, lx, ly, lamb=0.1)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_err(grad, ref["grad"]) <= TOL


def _ref_den(p, logits, lx):
    """The reference's own kernels (den_calculate.cu, compiled for gfx950 into oracle/_ref)."""
    so = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libden_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libden_ref.so not built (needs /root/reference at build time)")
    lib = ctypes.CDLL(so)
    B, T, V = logits.shape
    gpus = (ctypes.c_int * 1)(0)
    lib.Init(os.fsencode(p), 1, gpus)
    S = ctypes.c_int.in_dll(lib, "DEN_NUM_STATES").value
    x = torch.tensor(logits, device="cuda:0")
    lxd = torch.tensor(lx, dtype=torch.int32, device="cuda:0")
    alpha = torch.empty((T + 1) * B * S, device="cuda:0")
    beta = torch.empty(2 * B * S, device="cuda:0")
    gs = torch.empty(32 * B * V, device="cuda:0")
    grad = torch.zeros(B, T, V, device="cuda:0")
    ca, cb = torch.zeros(B, device="cuda:0"), torch.zeros(B, device="cuda:0")
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib.compute_alpha(vp(alpha.data_ptr()), vp(x.data_ptr()), B, T, S, V, vp(lxd.data_ptr()), vp(ca.data_ptr()), st)
    torch.cuda.synchronize()
    # alpha_lld_kernal (den_calculate.cu:122-161) reduces its last 64 partial sums through shared memory
    # with no barrier and no volatile (:150-154, "warp-synchronous"); built with hipcc -O2 for wave64 the
    # values are kept in registers and logZ comes out wrong by O(1) nats.  The forward recursion itself is
    # fine, so logZ is re-reduced here from the reference's OWN alpha buffer (row lx[b] already has
    # end_weight added by alpha_last_kernel) and handed to compute_beta_and_grad, which takes it as input.
    ca_kernel = ca.clone()
    al = alpha.view(B, T + 1, S)
    ca = torch.stack([torch.logsumexp(al[b, int(lx[b])], 0) for b in range(B)]).contiguous()
    lib.compute_beta_and_grad(vp(beta.data_ptr()), vp(alpha.data_ptr()), vp(x.data_ptr()), vp(ca.data_ptr()),
                              vp(gs.data_ptr()), vp(grad.data_ptr()), B, T, S, V, vp(lxd.data_ptr()), vp(cb.data_ptr()), st)
    print("reference alpha_lld_kernal logZ:", ca_kernel.cpu().numpy(), "re-reduced from its alpha:", ca.cpu().numpy())
    torch.cuda.synchronize()
    out = grad.cpu().numpy(), ca.cpu().numpy(), al.cpu().numpy().astype(np.float64)
    lib.Release(1, gpus)
    return out


@pytest.mark.parametrize("hist,fan,B,T,tol_ref,tol_alpha", [(256, 16, 4, 120, 2e-2, 1e-5), (2048, 24, 3, 500, 1e-1, 3e-5)])
def test_denominator_vs_reference_kernels(crf, tmp_path, hist, fan, B, T, tol_ref, tol_alpha):
    """gpu_den of this repo vs the REFERENCE'S OWN CUDA kernels built for gfx950 (oracle/Makefile `ref`),
    both judged against the fp64 oracle: the reference's fp32 log-domain arithmetic drifts with T
    (each alpha is rounded at ulp(|alpha|) ~ 3e-5 per frame), so its GRADIENT is held to 2e-2 at T = 120 and 1e-1 at
    T = 500 on the benchmark's graph (S = 4097), ours to 1e-4, and ours must be the closer of the two; logZ -- what
    "within 1e-4 of the reference build" can be demonstrated on directly -- agrees to 1e-4 all three ways.

    The TIGHT pin (round 4): the reference's `alpha` buffer [B][T+1][S] is deterministic (serial in-arc loop,
    den_calculate.cu:96-100) and is compared ENTRY BY ENTRY with the oracle's forward table: every finite entry of the fp64
    oracle within 1e-5 * max(1, |alpha|) of the reference's at T = 120 (3e-5 at T = 500, where the reference's own fp32 drift is
    1.4e-5 of |alpha| ~ 1500), the same -inf pattern (a wrong arc order, an off-by-one frame or a
    mis-read label moves entries by O(1)), and the fp32 build of the oracle -- the reference's arithmetic type, same
    operation order, only the libm differs from the device's -- within 2e-6 (a few ulp): the restatement follows the
    reference's arithmetic, not just its mathematics."""
    g, p = small_synth(tmp_path, 72, hist, fan, 4 if hist == 256 else 0)
    logits, _, lx, _ = make_batch(g, B, T, 72, seed=4, ragged=True)
    gref, cref, aref = _ref_den(p, logits, lx)
    gg = fst_io.read_fst(p)
    gor, cor, _, a64 = oracle.den_alpha(gg, logits, lx)
    g32, c32, _, a32 = oracle.den_alpha(gg, logits, lx, precision="f32")
    # --- the alpha tables, entry by entry (rows 0 .. lx[b]; the reference never writes the others) ---
    worst64 = worst32 = 0.0
    for b in range(B):
        n = int(lx[b]) + 1
        r, o64, o32 = aref[b, :n], a64[b, :n], a32[b, :n]
        assert np.array_equal(np.isfinite(r), np.isfinite(o64)) and np.array_equal(np.isfinite(r), np.isfinite(o32)), \
            "the -inf pattern of the forward table differs from the reference's"
        m = np.isfinite(r)
        scale = np.maximum(1.0, np.abs(r[m]))
        worst64 = max(worst64, float((np.abs(o64[m] - r[m]) / scale).max()))
        worst32 = max(worst32, float((np.abs(o32[m] - r[m]) / scale).max()))
    print(f"alpha table vs the reference's, max |d| / max(1, |alpha|): fp64 oracle {worst64:.2e}, fp32 oracle {worst32:.2e}")
    # measured on the GPU box (round 4): T = 120: fp64 1.8e-6, fp32 1.3e-7; T = 500: fp64 1.4e-5, fp32 2.1e-7 -- the fp32 build of the
    # oracle reproduces the reference's table to a few ulp (same operations in the same order; only libm's expf / log1pf differ
    # from the device's), which is what pins the restatement's arc order, frame indexing and label look-ups on the reference
    assert worst64 <= tol_alpha and worst32 <= 2e-6 and worst32 <= worst64
    # --- gradients: the reference accumulates them with CAS log-adds into 32 striped slots (den_calculate.cu:37-49, 221-223), in an
    # order that changes from run to run, so neither build of the oracle is "the same arithmetic" there: both sit at the
    # reference's own fp32 noise (measured 1.4e-2 / 1.5e-2 at T = 500) and are held to the bound the reference itself is held to
    e32_ref, e64_ref = rel_err(g32, gref), rel_err(gor, gref)
    print(f"gradient vs the reference's: fp32 oracle {e32_ref:.2e}, fp64 oracle {e64_ref:.2e}")
    assert e32_ref <= tol_ref and e64_ref <= tol_ref
    core = crf._C
    ctx = crf.CRFContext(p, 0)
    x = torch.tensor(logits, device="cuda:0")
    _, gd, ex = core.loss_fwd_bwd(x, None, torch.tensor(lx), None, 1.0, 0.0, core.graph_for(x.device), True)
    ours, cours = gd.cpu().numpy(), ex["costs_alpha"].cpu().numpy()
    e_ref, e_ours = rel_err(gref, gor), rel_err(ours, gor)
    print(f"S={g['S']} T={T}: reference kernels vs fp64 oracle: {e_ref:.2e}; this repo vs fp64 oracle: {e_ours:.2e}; "
          f"logZ: reference {np.abs(cref / cor - 1).max():.1e}, ours {np.abs(cours / cor - 1).max():.1e}, ours vs reference {np.abs(cours / cref - 1).max():.1e}")
    assert np.allclose(cref, cor, rtol=1e-4) and np.allclose(cours, cor, rtol=TOL) and np.allclose(cours, cref, rtol=1e-4)
    assert e_ref <= tol_ref and e_ours <= TOL and e_ours <= e_ref
    assert rel_err(ours, gref) <= tol_ref
    del ctx


def _ref_ctc(logits, labels, lx, ly, poison=True):
    """The reference's own numerator (gpu_ctc/ctc_entrypoint.cu + gpu_ctc.h + gpu_ctc_kernels.h, compiled in place for gfx950 into oracle/_ref by
    `make -C oracle ref`; only the two moderngpu includes and hostdevice.h are redirected to own stubs) through ITS C entry points
    (ctc.h:76-109): -> status, costs [B] (= +log p, gpu_ctc.h:364-369), grads [B,T,V], the alpha workspace per utterance (list of [lx_b, 2 ly_b + 1])."""
    so = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libctc_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libctc_ref.so not built (needs /root/reference at build time)")
    lib = ctypes.CDLL(so)

    class Opt(ctypes.Structure):
        _fields_ = [("stream", ctypes.c_void_p), ("blank_label", ctypes.c_int)]
    B, T, V = logits.shape
    ly_a, lx_a = np.ascontiguousarray(ly, dtype=np.int32), np.ascontiguousarray(lx, dtype=np.int32)
    lab_a = np.ascontiguousarray(labels, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    opt = Opt(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0)
    size = ctypes.c_size_t(0)
    lib.get_workspace_size.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, Opt, ctypes.POINTER(ctypes.c_size_t)]
    assert lib.get_workspace_size(ip(ly_a), ip(lx_a), V, B, opt, ctypes.byref(size)) == 0
    act = torch.tensor(logits, device="cuda:0").transpose(0, 1).contiguous()      # [T, B, V] (binding.cpp:86-117 hands the kernels this layout)
    grads = torch.zeros_like(act)
    ws = torch.full(((size.value + 3) // 4,), float("nan") if poison else 0.0, device="cuda:0")   # (what the reference returns for an invalid utterance is workspace)
    costs = np.full(B, np.nan, dtype=np.float32)
    lib.compute_ctc_loss.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                     ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, Opt]
    torch.cuda.synchronize()
    status = lib.compute_ctc_loss(act.data_ptr(), grads.data_ptr(), ip(lab_a), ip(ly_a), ip(lx_a), V, B,
                                  costs.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ws.data_ptr(), opt)
    torch.cuda.synchronize()
    # the alpha workspace (gpu_ctc.h:100-236): 6 B words of scalars, Lmax * B + Smax * B label words, then B blocks of S_ * T_ floats, S_ / T_ over the
    # VALID utterances; utterance b's row t starts at t * (2 ly_b + 1) of its block (gpu_ctc_kernels.h:123, 196)
    rep = [int(sum(labels[int(sum(ly[:b])) + i] == labels[int(sum(ly[:b])) + i - 1] for i in range(1, int(ly[b])))) for b in range(B)]
    valid = [int(ly[b]) + rep[b] <= int(lx[b]) for b in range(B)]
    S_ = 2 * max([int(ly[b]) for b in range(B) if valid[b]] + [0]) + 1
    T_ = max([int(lx[b]) for b in range(B) if valid[b]] + [0])
    Lmax, Smax = int(max(ly)), 2 * int(max(ly)) + 1
    off = 6 * B + Lmax * B + Smax * B
    w = ws.cpu().numpy()
    alphas = []
    for b in range(B):
        S = 2 * int(ly[b]) + 1
        blk = w[off + b * S_ * T_: off + (b + 1) * S_ * T_]
        alphas.append(blk[: int(lx[b]) * S].reshape(int(lx[b]), S).astype(np.float64) if valid[b] else None)
    return status, costs.astype(np.float64), grads.transpose(0, 1).contiguous().cpu().numpy(), alphas, valid


@pytest.mark.parametrize("case", ["fixture", "L83_T500", "L250_T1500", "repeats", "empty_label", "invalid"])
def test_numerator_vs_reference_kernels(crf, golden_dir, case):
    """Rows a10 - a14 of SURVEY 8a pinned on the REFERENCE'S OWN numerator kernels (round 6; until now the oracle's numerator half rested on a
    brute-force enumerator and torch's ctc_loss only).  `compute_ctc_loss` of the reference -- compute_alpha_kernel / compute_betas_and_grad_kernel
    (gpu_ctc_kernels.h:87-213, 218-458) compiled in place for gfx950 -- against
      * the fp32 build of the oracle (same arithmetic type, same operation order): the alpha workspace ENTRY BY ENTRY (same -inf pattern; a few
        ulp of |alpha|), costs, gradients;
      * the fp64 oracle: costs to 1e-5 relative;
      * this repo's HIP numerator (gpu_ctc mirror of binding.cpp:86-117): costs and gradients within 1e-4.
    Quirks of the reference, observed here and stated in DESIGN 1: the cost is +log p(l | x) (gpu_ctc.h:364-369 copies nll_forward_, which
    compute_alpha_kernel fills with the log-likelihood itself, gpu_ctc_kernels.h:198-212); an utterance with L + repeats > T makes both kernels
    return early (:108-109, :261-262) -- its cost is whatever the workspace held (NaN-poisoned here: NaN comes back), its gradient rows stay as
    the caller zeroed them; this repo returns cost 0, zero rows and invalid = 1 for it."""
    rng = np.random.default_rng(11)
    V = 72
    if case == "fixture":                                   # the reference's only test input (src/ctc_crf/test/main.py:15-28): L = 3, T = 5
        k = json.load(open(os.path.join(golden_dir, "kat_fixture.json")))
        logits = np.log(np.array(k["probs"], dtype=np.float32))[None]
        labels, lx, ly = np.array(k["labels"], dtype=np.int32), np.array([5], dtype=np.int32), np.array([3], dtype=np.int32)
        V = logits.shape[2]
    else:
        B, T, L = {"L83_T500": (3, 500, 83), "L250_T1500": (2, 1500, 250), "repeats": (3, 60, 20), "empty_label": (3, 40, 6), "invalid": (3, 24, 10)}[case]
        logits = log_softmax_np(rng.normal(0.0, 2.0, size=(B, T, V))).astype(np.float32)
        lx = np.array([T, T - T // 5, T - T // 3][:B], dtype=np.int32)
        ly = np.array([L, L - L // 4, L // 2][:B], dtype=np.int32)
        lab = [rng.integers(1, V, size=int(n)) for n in ly]
        if case == "repeats":                               # runs of equal labels: the blank between them is mandatory (L + repeats <= T)
            lab = [np.repeat(rng.integers(1, V, size=(int(n) + 2) // 3), 3)[: int(n)] for n in ly]
        if case == "empty_label":
            ly[1] = 0; lab[1] = lab[1][:0]
        if case == "invalid":                               # utterance 1: L + repeats > T
            lx[1] = 8; ly[1] = 10; lab[1] = rng.integers(1, V, size=10)
        labels = np.concatenate(lab).astype(np.int32)
    B, T = logits.shape[0], logits.shape[1]
    # what the reference's fp32 log-domain GRADIENT is held to against exact arithmetic: like its denominator it drifts with T (every alpha
    # rounded at ulp(|alpha|); measured 2.4e-3 at T = 500, 1.2e-2 at T = 1 500) -- the alpha table and the costs are the tight pins
    tol_ref = 3e-2 if T >= 1000 else 6e-3 if T >= 300 else 2e-3
    status, cref, gref, aref, valid = _ref_ctc(logits, labels, lx, ly)
    assert status == 0
    g32, c32, v32, a32 = oracle.ctc_alpha(logits, labels, lx, ly, precision="f32")
    g64, c64, v64, _ = oracle.ctc_alpha(logits, labels, lx, ly)
    assert list(v32) == [int(v) for v in valid] == list(v64)
    worst = 0.0
    for b in range(B):
        if not valid[b]:
            assert np.isnan(cref[b]) and np.all(gref[b] == 0.0), "an invalid utterance: the reference leaves cost and gradient rows untouched"
            continue
        n, S = int(lx[b]), 2 * int(ly[b]) + 1
        r, o = aref[b], a32[b, :n, :S]
        assert np.array_equal(np.isfinite(r), np.isfinite(o)), f"utterance {b}: the -inf pattern of the forward table differs from the reference's"
        m = np.isfinite(r)
        if m.any():
            worst = max(worst, float((np.abs(o[m] - r[m]) / np.maximum(1.0, np.abs(r[m]))).max()))
        assert abs(cref[b] - c32[b]) <= 2e-6 * max(1.0, abs(c32[b])) and abs(cref[b] - c64[b]) <= 1e-5 * max(1.0, abs(c64[b])), (b, cref[b], c32[b], c64[b])
        assert np.all(gref[b, n:] == 0.0)
        # (fp32 oracle vs the reference's gradient: the same arithmetic type, but the reference sums a label's states by a segmented reduction
        # in sorted-label order, gpu_ctc_kernels.h:395-402, the oracle in state order: 2.4e-4 at T = 500)
        assert rel_err(gref[b, :n], g32[b, :n]) <= 1e-3 and rel_err(gref[b, :n], g64[b, :n]) <= tol_ref, (b, rel_err(gref[b, :n], g32[b, :n]), rel_err(gref[b, :n], g64[b, :n]))
    print(f"{case}: numerator alpha table, fp32 oracle vs the reference's, max |d| / max(1, |alpha|): {worst:.2e}; costs reference {cref}, fp64 oracle {c64}")
    assert worst <= 2e-6
    # --- this repo's numerator through the pybind-shaped mirror (binding.cpp:86-117) ---
    core = crf._C
    act = torch.tensor(logits, device="cuda:0").transpose(0, 1).contiguous()
    gc = torch.zeros_like(act)
    cc = torch.zeros(B)
    core.gpu_ctc(act, gc, torch.tensor(labels, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), B, cc, 0)
    ours = gc.transpose(0, 1).cpu().numpy()
    for b in range(B):
        n = int(lx[b])
        if not valid[b]:
            assert cc[b].item() == 0.0 and np.all(ours[b] == 0.0)
            continue
        assert abs(cc[b].item() - c64[b]) <= TOL * max(1.0, abs(c64[b])) and abs(cc[b].item() - cref[b]) <= TOL * max(1.0, abs(cref[b]))
        assert rel_err(ours[b, :n], g64[b, :n]) <= TOL and rel_err(ours[b, :n], gref[b, :n]) <= tol_ref
        assert rel_err(ours[b, :n], g64[b, :n]) <= rel_err(gref[b, :n], g64[b, :n]) + 1e-6, "this repo is at least as close to exact arithmetic as the reference's fp32 kernels"


def test_functional_and_errors(crf, golden_dir, tmp_path):
    k = json.load(open(os.path.join(golden_dir, "kat_fixture.json")))
    lp = torch.tensor(np.log(np.array(k["probs"], dtype=np.float32))[None], device="cuda:0", requires_grad=True)
    loss = crf.ctc_crf_loss(lp, torch.tensor(k["labels"]), torch.tensor([5]), torch.tensor([3]),
                            os.path.join(golden_dir, "den_lm_fixture.fst"), lamb=k["lamb"])
    assert loss.shape == (1,) and abs(loss.item() - k["loss"]) <= TOL * abs(k["loss"])
    (2.0 * loss).backward()  # backward multiplies the saved grads by grad_output (__init__.py:92-94)
    assert rel_err(lp.grad[0].cpu().numpy(), 2.0 * np.array(k["grad"])) <= TOL
    with pytest.raises(RuntimeError):
        crf.CRFContext(os.path.join(str(tmp_path), "missing.fst"), 0)
    with pytest.raises(RuntimeError):
        crf.CRFContext(os.path.join(golden_dir, "den_lm_fixture.fst"), 99)
    bad = os.path.join(str(tmp_path), "bad.fst")
    open(bad, "wb").write(b"not an fst at all")
    with pytest.raises(RuntimeError):
        crf.CRFContext(bad, 0)
    with pytest.raises(AssertionError):  # dtype asserts of CTC_CRF_LOSS.forward (__init__.py:116-124)
        crf.CTC_CRF_LOSS()(lp.double(), torch.tensor([1], dtype=torch.int32), torch.tensor([5], dtype=torch.int32),
                           torch.tensor([1], dtype=torch.int32))


@pytest.mark.parametrize("H,fanout", [(256, 16), (96, 12)])
def test_small_graphs_spread_rows(crf, tmp_path, H, fanout):
    """Small den_lm (fewer slices of rows than the workgroup has waves) get their rows cut into pieces on adjacent lanes so that every wave has a
    short list (res_layout.cpp "SMALL graphs", a cost model calibrated on S = 513: round-4 advisor).  Both layouts -- the spread one the planner
    picks and the unspread one behind `res_no_spread` -- against the oracle, on the calibration graph and on a second one."""
    from cat_amd.den_lm import synth_den_lm
    V = 72
    p = os.path.join(str(tmp_path), "small.fst")
    g = synth_den_lm(V, H, fanout, 0, path=p)
    logits, labels, lx, ly = make_batch(g, 5, 150, V, seed=H, ragged=True)
    ref = oracle.ctc_crf(fst_io.read_fst(p), logits, labels, lx, ly, lamb=0.1)
    stats = {}
    for name, opts in (("spread", {}), ("unspread", {"res_no_spread": 1})):
        with crf._C.debug_opts(**opts):
            ctx = crf.CRFContext(p, 0)
            st = crf._C.graph_stats(crf._C.graph_for(torch.device("cuda", 0)))
            assert st["fac"] == 1 and st["fac_geom"] == 4, st          # the 1024-thread geometry takes both
            stats[name] = (st["fac_fwd_slots"], st["fac_bwd_slots"], st["fac_chunks"], st["fac_Gf"], st["fac_Gb"])
            x = torch.tensor(logits, device="cuda:0", requires_grad=True)
            loss = crf.CTC_CRF_LOSS(lamb=0.1)(x, torch.tensor(labels, dtype=torch.int32), torch.tensor(lx, dtype=torch.int32), torch.tensor(ly, dtype=torch.int32))
            loss.backward()
            grad = x.grad.cpu().numpy()
            del ctx
        assert abs(loss.item() - ref["loss"]) <= TOL * abs(ref["loss"]), name
        for b in range(5):
            assert rel_err(grad[b], ref["grad"][b]) <= TOL, (name, b)
    print(stats)
    if H == 256:
        assert stats["spread"] != stats["unspread"], stats            # (the calibration graph IS spread by default)
