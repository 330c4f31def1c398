"""Shared helpers for the parity tests (seeded synthetic inputs, graph <-> file)."""
import os

import numpy as np

from cat_amd.den_lm import synth_den_lm, write_fst
from cat_amd.synth import log_softmax_np, make_batch  # noqa: F401  (re-exported: the tests import them from here)


def graph_to_file(g, path):
    write_fst(path, g["S"], int(g.get("start", 0)), g["src"], g["dst"], g["lab"] + 1, g["lab"] + 1,
              -g["w"], -g["end_w"])
    return path


def rel_err(a, b, floor=0.0):
    """Gradient parity metric (BASELINE 'within 1e-4 rel'): the larger of
      * max |a-b| / max |b|            (worst entry, relative to the largest entry), and
      * ||a-b||_2 / ||b||_2            (norm-wise relative error).
    An entry-by-entry relative error is NOT used for the combined gradient: it is a difference
    gamma_den - (1+lamb) gamma_ctc of two O(1) posteriors stored in fp32, so entries near zero carry
    cancellation noise of ~1e-7 absolute no matter how they are computed (the fp32 log-domain
    arithmetic of the reference itself is ~1e-2 away from exact arithmetic at T=500, measured with
    oracle f32 vs f64 -- see DESIGN.md 'Parity metric').
    `floor` (the fuzz: 0.05 x the scale of the two posterior terms, 1 or 1/B): with a peaked network output and a small lamb the WHOLE
    gradient of an utterance is the cancellation -- gamma_den = gamma_ctc to 1e-6, the gradient is -lamb * gamma_ctc, max |b| = lamb * scale --
    and 1e-4 of THAT would ask for posteriors exact to 1e-6 of their scale, below what fp32 emission factors give any implementation (each
    exp(logp - max) carries 6e-8; a path multiplies T of them: 1e-6 at T = 120, 4e-6 at T = 1 500).  The error is then measured against
    max(max |b|, floor); gamma_den and gamma_ctc themselves are held to 1e-4 ENTRY-wise in the same tests (post_err)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), floor, 1e-30)
    nrm = max(np.linalg.norm(b), floor * np.sqrt(b.shape[0] if b.ndim > 1 else 1.0), 1e-30)
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


def transform_graph(g, path, seed=0, renumber=True, reorder=True, push=False):
    """What OpenFst tools do to a den_lm between composition and the file the loss reads
    (cat/utils/tool/prep_den_lm.sh:48-49, ``fstcompose | fstdeterminizestar --use-log=true``): states are
    re-numbered (discovery order), the arcs of a state come in another order, and -- for `push` -- weights are
    moved along paths by a potential (w' = w + V(dst) - V(src), final' = final - V(s), V(start) = 0), which
    leaves every path weight, hence the loss, unchanged.  Returns the transformed graph (reference conventions)
    and writes it to `path`."""
    rng = np.random.default_rng(seed)
    S = int(g["S"])
    src, dst, lab = g["src"].astype(np.int64), g["dst"].astype(np.int64), g["lab"].astype(np.int32)
    w, end_w = g["w"].astype(np.float64), g["end_w"].astype(np.float64)
    start = int(g.get("start", 0))
    if push:
        pot = rng.normal(0.0, 1.0, size=S)
        pot[start] = 0.0
        w = w + pot[dst] - pot[src]
        end_w = np.where(np.isfinite(end_w), end_w - pot, end_w)
    perm = rng.permutation(S) if renumber else np.arange(S)          # old id -> new id
    src, dst = perm[src], perm[dst]
    end2 = np.empty(S, dtype=np.float64)
    end2[perm] = end_w
    order = rng.permutation(len(src)) if reorder else np.arange(len(src))
    src, dst, lab, w = src[order], dst[order], lab[order], w[order]
    write_fst(path, S, int(perm[start]), src, dst, lab + 1, lab + 1, -w.astype(np.float32), -end2.astype(np.float32))
    from oracle import fst_io
    return fst_io.read_fst(path)


class crf_env:
    """``with crf_env(CRF_NO_FACTORED=1, CRF_BAT_UL=16): ...`` -- the library's debug switches (include/ctc_crf_hip.h
    crf_debug_set; the library never reads the environment) under their historical upper-case names: CRF_X_Y is the switch
    x_y.  Values are restored when the block ends."""

    def __init__(self, **kw):
        self.kw = {k[4:].lower() if k.startswith("CRF_") else k: int(v) for k, v in kw.items()}

    def __enter__(self):
        from cat_amd.ctc_crf import _C
        self._cm = _C.debug_opts(**self.kw)
        self._cm.__enter__()
        return self

    def __exit__(self, *a):
        return self._cm.__exit__(*a)
