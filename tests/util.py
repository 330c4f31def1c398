"""Shared helpers for the parity tests (seeded synthetic inputs, graph <-> file)."""
import os

import numpy as np

from cat_amd.den_lm import random_labels_from_graph, synth_den_lm, write_fst


def graph_to_file(g, path):
    write_fst(path, g["S"], int(g.get("start", 0)), g["src"], g["dst"], g["lab"] + 1, g["lab"] + 1,
              -g["w"], -g["end_w"])
    return path


def log_softmax_np(x):
    m = x.max(-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))).astype(np.float32)


def make_batch(g, B, T, V, seed=0, ragged=True, scale=2.0, label_frac=6, min_len=1):
    """SURVEY 8d synthetic inputs: log_softmax(N(0,1)*scale); lx ragged in [0.6T, T] sorted descending;
    ly = lx // label_frac; labels walk the graph."""
    rng = np.random.default_rng(seed)
    logits = log_softmax_np(rng.normal(0.0, 1.0, size=(B, T, V)) * scale)
    if ragged:
        lx = np.sort(rng.integers(max(min_len, int(0.6 * T)), T + 1, size=B))[::-1].astype(np.int32)
        lx[0] = T
    else:
        lx = np.full(B, T, dtype=np.int32)
    ly = np.maximum(lx // label_frac, 0).astype(np.int32)
    labels = np.concatenate([random_labels_from_graph(g, int(n), rng) for n in ly]) if ly.sum() else np.zeros(0, np.int32)
    return logits, labels.astype(np.int32), lx, ly


def rel_err(a, b):
    """Gradient parity metric (BASELINE 'within 1e-4 rel'): the larger of
      * max |a-b| / max |b|            (worst entry, relative to the largest entry), and
      * ||a-b||_2 / ||b||_2            (norm-wise relative error).
    An entry-by-entry relative error is NOT used for the combined gradient: it is a difference
    gamma_den - (1+lamb) gamma_ctc of two O(1) posteriors stored in fp32, so entries near zero carry
    cancellation noise of ~1e-7 absolute no matter how they are computed (the fp32 log-domain
    arithmetic of the reference itself is ~1e-2 away from exact arithmetic at T=500, measured with
    oracle f32 vs f64 -- see DESIGN.md 'Parity metric')."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-30)
    nrm = max(np.linalg.norm(b), 1e-30)
    return float(max(np.abs(a - b).max() / den, np.linalg.norm(a - b) / nrm))


def post_err(a, b, floor=1e-3):
    """Entry-by-entry relative error for a posterior matrix (non-negative, no cancellation), over the
    entries >= floor."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = b >= floor
    return float((np.abs(a - b)[m] / b[m]).max()) if m.any() else 0.0


def small_synth(tmpdir, vocab=8, histories=16, fanout=4, seed=1):
    p = os.path.join(str(tmpdir), f"synth_v{vocab}_h{histories}_d{fanout}_s{seed}.fst")
    g = synth_den_lm(vocab, histories, fanout, seed, path=p)
    return g, p
