"""cat_amd/synth.py -- seeded synthetic inputs of SURVEY.md 8(d): what bench.py, the tools and the parity
tests run on (no corpus, no network).  ``log_probs = log_softmax(N(0,1) * 2.0)``, frame lengths all = T or
ragged ``U[0.6T, T]`` sorted descending (CAT sorts by length, cat/shared/data.py:398-399), label lengths
``lx // 6``, labels that walk the synthetic LM so the numerator path has denominator mass."""
import numpy as np

from .den_lm import random_labels_from_graph


def log_softmax_np(x):
    m = x.max(-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))).astype(np.float32)


def make_batch(g, B, T, V, seed=0, ragged=True, scale=2.0, label_frac=6, min_len=1):
    """-> (log_probs [B,T,V] f32, labels [sum ly] i32, lx [B] i32, ly [B] i32)."""
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
