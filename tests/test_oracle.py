"""CPU tests (no GPU): the oracle (oracle/crf_oracle.c) against every pin we have --
the committed golden vectors, a fresh brute-force enumeration, torch's CPU ctc_loss, and its own
invariants.  The oracle is test infrastructure; these tests are what makes it trustworthy."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import fst_io
from oracle.brute import brute_ctc, brute_den
from cat_amd.den_lm import synth_den_lm, write_fst
from tests.util import make_batch, small_synth


def test_fixture_bytes_match_reference_hash(golden_dir):
    """tests/golden/den_lm_fixture.fst was re-created from text (make_golden.py) and verified there to
    be byte-identical to reference src/ctc_crf/test/den_lm.fst; pin the hash."""
    k = json.load(open(os.path.join(golden_dir, "kat_fixture.json")))
    sha = hashlib.sha256(open(os.path.join(golden_dir, "den_lm_fixture.fst"), "rb").read()).hexdigest()
    assert sha == k["sha256_den_lm"]
    ref = "/root/reference/src/ctc_crf/test/den_lm.fst"
    if os.path.exists(ref):
        assert hashlib.sha256(open(ref, "rb").read()).hexdigest() == sha
    g = fst_io.read_fst(os.path.join(golden_dir, "den_lm_fixture.fst"))
    assert (g["S"], g["A"], g["start"]) == (9, 24, 0)
    assert sorted(np.flatnonzero(np.isfinite(g["end_w"]))) == [4, 6]


@pytest.mark.parametrize("prec,tol", [("f64", 1e-7), ("f32", 2e-6)])
def test_oracle_vs_fixture_kat(golden_dir, prec, tol):
    k = json.load(open(os.path.join(golden_dir, "kat_fixture.json")))
    g = fst_io.read_fst(os.path.join(golden_dir, "den_lm_fixture.fst"))
    logits = np.log(np.array(k["probs"], dtype=np.float32))[None]
    r = oracle.ctc_crf(g, logits, np.array(k["labels"]), np.array([5]), np.array([3]), lamb=k["lamb"], precision=prec)
    assert abs(r["loss"] - k["loss"]) <= tol * abs(k["loss"])
    assert abs(r["costs_den"][0] - k["logZ_den"]) <= tol * abs(k["logZ_den"])
    assert abs(r["costs_ctc"][0] - k["logp_ctc"]) <= tol * abs(k["logp_ctc"])
    assert np.abs(r["grad"][0] - np.array(k["grad"])).max() <= max(tol, 1e-7)
    # SURVEY section 4 table (derived independently there)
    assert abs(k["loss"] - (-2.478624763)) < 1e-6 and abs(k["logZ_den"] - (-6.258327797)) < 1e-6


def test_oracle_vs_random_golden(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "kat_random.json"))):
        g = fst_io.read_fst(os.path.join(golden_dir, c["fst"]))
        lg = np.array(c["logits"], dtype=np.float32)[None]
        T = lg.shape[1]
        gd, ca, cb = oracle.den(g, lg, np.array([T]))
        assert abs(ca[0] - c["logZ_den"]) < 1e-9 and abs(cb[0] - c["logZ_den"]) < 1e-9
        assert np.abs(gd[0] - np.array(c["gamma_den"])).max() < 1e-7
        gc, cc, valid = oracle.ctc(lg, np.array(c["labels"], dtype=np.int32), np.array([T]), np.array([len(c["labels"])]))
        if c["logp_ctc"] is None:
            assert valid[0] == 0 and cc[0] == 0.0 and np.all(gc == 0)
        else:
            assert valid[0] == 1 and abs(cc[0] - c["logp_ctc"]) < 1e-9
            assert np.abs(gc[0] - np.array(c["gamma_ctc"])).max() < 1e-7


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_oracle_vs_fresh_brute_force(tmp_path, seed):
    """Random tiny non-deterministic graphs with repeats / empty labels, enumerated exhaustively."""
    rng = np.random.default_rng(100 + seed)
    S, V, T = int(rng.integers(2, 5)), int(rng.integers(3, 5)), int(rng.integers(3, 6))
    n = S * 3
    src, dst = rng.integers(0, S, n), rng.integers(0, S, n)
    il = rng.integers(1, V + 1, n)
    cost = rng.uniform(0, 2, n)
    final = np.where(rng.random(S) < 0.7, rng.uniform(0, 1, S), np.inf)
    final[-1] = 0.3
    p = os.path.join(str(tmp_path), "g.fst")
    write_fst(p, S, 0, src, dst, il, il, cost, final)
    g = fst_io.read_fst(p)
    x = rng.normal(0, 2, (T, V))
    lg = (x - np.log(np.exp(x).sum(-1, keepdims=True))).astype(np.float32)
    labels = [[1, 1], [2], [], [1, 2, 1]][seed][: max(0, T - 1)]
    lz, gd = brute_den(g, lg.astype(np.float64))
    lp, gc = brute_ctc(lg.astype(np.float64), labels)
    r_gd, ca, cb = oracle.den(g, lg[None], np.array([T]))
    assert abs(ca[0] - lz) < 1e-9 and abs(cb[0] - lz) < 1e-9 and np.abs(r_gd[0] - gd).max() < 1e-7
    r_gc, cc, valid = oracle.ctc(lg[None], np.array(labels, dtype=np.int32), np.array([T]), np.array([len(labels)]))
    if np.isfinite(lp):
        assert valid[0] == 1 and abs(cc[0] - lp) < 1e-9 and np.abs(r_gc[0] - gc).max() < 1e-7
    else:
        assert valid[0] == 0


def test_oracle_ctc_vs_torch_cpu():
    """Third-party pin for the numerator: torch.nn.functional.ctc_loss on CPU (unrelated code)."""
    rng = np.random.default_rng(7)
    B, T, V = 5, 40, 13
    x = torch.tensor(rng.normal(size=(B, T, V)), dtype=torch.float32).log_softmax(-1)
    ly = np.array([6, 0, 11, 3, 8], dtype=np.int32)
    lx = np.array([40, 9, 31, 40, 22], dtype=np.int32)
    labels = rng.integers(1, V, size=int(ly.sum())).astype(np.int32)
    labels[0:2] = 4  # a repeat
    gc, cc, valid = oracle.ctc(x.numpy(), labels, lx, ly)
    assert valid.all()
    xr = x.clone().requires_grad_(True)
    nll = torch.nn.functional.ctc_loss(xr.transpose(0, 1), torch.tensor(labels).long(), torch.tensor(lx).long(),
                                       torch.tensor(ly).long(), blank=0, reduction="none")
    nll.sum().backward()
    assert np.allclose(-cc, nll.detach().numpy(), rtol=1e-5)
    mask = (np.arange(T)[None, :] < lx[:, None])[..., None]
    gamma_torch = (x.exp().numpy() * mask - xr.grad.numpy())  # torch returns softmax - gamma
    assert np.abs(gc - gamma_torch).max() < 2e-5


def test_oracle_invariants_and_ragged(tmp_path):
    g, p = small_synth(tmp_path, 10, 24, 5, 3)
    logits, labels, lx, ly = make_batch(g, 4, 30, 10, seed=2, ragged=True)
    gg = fst_io.read_fst(p)
    gd, ca, cb = oracle.den(gg, logits, lx)
    gc, cc, valid = oracle.ctc(logits, labels, lx, ly)
    assert np.allclose(ca, cb, rtol=1e-12) and valid.all()
    for b in range(4):
        n = int(lx[b])
        assert np.allclose(gd[b, :n].sum(-1), 1.0, atol=1e-6) and np.allclose(gc[b, :n].sum(-1), 1.0, atol=1e-6)
        assert np.all(gd[b, n:] == 0) and np.all(gc[b, n:] == 0)
    r = oracle.ctc_crf(gg, logits, labels, lx, ly, lamb=0.2, size_average=True)
    assert abs(r["loss"] - (ca - 1.2 * cc).sum() / 4) < 1e-9
    assert np.abs(r["grad"] - (gd - 1.2 * gc) / 4).max() < 1e-7
    # finite differences of the oracle loss w.r.t. a few log-prob entries (fp64 oracle, fp32 inputs)
    rng = np.random.default_rng(0)
    for _ in range(4):
        b, t, v = int(rng.integers(4)), int(rng.integers(int(lx.min()))), int(rng.integers(10))
        eps = 1e-2
        lp, lm = logits.copy(), logits.copy()
        lp[b, t, v] += eps
        lm[b, t, v] -= eps
        fd = (oracle.ctc_crf(gg, lp, labels, lx, ly, lamb=0.2)["loss"] - oracle.ctc_crf(gg, lm, labels, lx, ly, lamb=0.2)["loss"]) / (lp[b, t, v] - lm[b, t, v])
        assert abs(fd - r["grad"][b, t, v]) < 2e-4


def test_fst_writer_reader_roundtrip(tmp_path):
    p = os.path.join(str(tmp_path), "s.fst")
    g = synth_den_lm(12, 30, 5, seed=9, path=p)
    r = fst_io.read_fst(p)
    assert r["S"] == g["S"] == 61 and r["A"] == g["A"]
    for k in ("src", "dst", "lab"):
        assert np.array_equal(r[k], g[k])
    assert np.allclose(r["w"], g["w"]) and np.allclose(r["end_w"], g["end_w"]) and r["start_w"][0] == 0
    with pytest.raises(ValueError):
        write_fst(p, 2, 0, [0], [1], [0], [0], [0.0], [np.inf, 0.0])  # epsilon ilabel
