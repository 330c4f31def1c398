"""The identity behind the tilted numerator chains of the HIP kernels (cat_amd/csrc/crf_kernels.hip: ctc_rho, ctc_forward, ctc_backward;
reference recursion: src/ctc_crf/gpu_ctc/gpu_ctc_kernels.h:87-458), restated in numpy fp64 and checked against the oracle's plain CTC:

    A'_t[s] = A_t[s] rho^s,   Bx'_t[s] = Bx_t[s] rho^(Sx-1-s)
    A'_t[s]  = e_t[s] (A'_{t-1}[s] + rho A'_{t-1}[s-1] + rho^2 skip A'_{t-1}[s-2])
    Bx'_t[s] = Y'_{t+1}[s] + rho Y'_{t+1}[s+1] + rho^2 skip Y'_{t+1}[s+2],   Y' = e Bx'
    Z' = A'_T[Sx-1] + rho A'_T[Sx-2],   log Z = log Z' - (Sx-1) log rho,   gamma_t[s] = A'_t[s] Bx'_t[s] / Z'

exact for ANY rho > 0 -- what rho changes is only where the floating-point range is spent.  (The kernels themselves are tested on the
GPU: tests/test_gpu_parity.py::test_numerator_*.)"""
import numpy as np
import pytest

import oracle


def _tilted_ctc(logp, labels, rho):
    T, V = logp.shape
    L = len(labels)
    Sx = 2 * L + 1
    lab = np.zeros(Sx, dtype=np.int64)
    lab[1::2] = labels
    skip_f = np.array([s >= 2 and lab[s] != 0 and lab[s] != lab[s - 2] for s in range(Sx)])
    e = np.exp(logp[:, lab])                                  # [T, Sx]
    A = np.zeros((T, Sx))
    A[0, 0] = e[0, 0]
    if Sx > 1:
        A[0, 1] = e[0, 1] * rho
    for t in range(1, T):
        a = A[t - 1].copy()
        a[1:] += rho * A[t - 1, :-1]
        a[2:] += rho * rho * np.where(skip_f[2:], A[t - 1, :-2], 0.0)
        A[t] = e[t] * a
    Bx = np.zeros((T, Sx))
    Bx[T - 1, Sx - 1] = 1.0
    if Sx > 1:
        Bx[T - 1, Sx - 2] = rho
    for t in range(T - 2, -1, -1):
        Y = e[t + 1] * Bx[t + 1]
        b = Y.copy()
        b[:-1] += rho * Y[1:]
        b[:-2] += rho * rho * np.where(skip_f[2:], Y[2:], 0.0)   # the skip INTO s + 2 is allowed iff skip_f[s + 2]
        Bx[t] = b
    Zp = A[T - 1, Sx - 1] + (rho * A[T - 1, Sx - 2] if Sx > 1 else 0.0)
    logZ = np.log(Zp) - (Sx - 1) * np.log(rho)
    post = A * Bx / Zp                                        # [T, Sx] state posteriors
    gamma = np.zeros((T, V))
    for s in range(Sx):
        gamma[:, lab[s]] += post[:, s]
    return logZ, gamma


@pytest.mark.parametrize("rho", [1.0, 0.35, 0.05, 2.5])
def test_tilted_chains_are_exact(rho):
    rng = np.random.default_rng(7)
    T, V, L = 60, 9, 11
    x = rng.normal(size=(1, T, V)) * 1.5
    logp = (x - np.log(np.exp(x).sum(-1, keepdims=True))).astype(np.float32)
    labels = rng.integers(1, V, size=L).astype(np.int32)
    labels[3] = labels[2]                                     # a repeated label: no skip over the blank between them
    gref, cref, valid = oracle.ctc(logp, labels, np.array([T], dtype=np.int32), np.array([L], dtype=np.int32))
    assert valid.all()
    logZ, gamma = _tilted_ctc(logp[0].astype(np.float64), labels, rho)
    assert abs(logZ - cref[0]) <= 1e-9 * abs(cref[0])
    assert np.abs(gamma - gref[0]).max() <= 2e-7             # (the oracle returns its posteriors as float32)
    assert np.allclose(gamma.sum(-1), 1.0, atol=1e-12)
