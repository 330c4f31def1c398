"""cat_amd/den_lm.py -- host-side denominator-graph tooling (numpy only, no OpenFst / Kaldi).

* ``write_fst``     -- OpenFst ``vector``/``standard`` binary writer (the format
                       ``StdVectorFst::Read`` consumes at reference src/ctc_crf/gpu_den/fst_read.cc:23).
* ``synth_den_lm``  -- synthetic ``T.fst o n-gram`` denominator graph: the CTC topology of
                       reference cat/utils/tool/build_ctc_topo.py:48-60 composed by hand with a random
                       token n-gram acceptor; what bench.py and the parity tests run on, because the
                       real pipeline (cat/utils/tool/prep_den_lm.sh:40-51) needs Kaldi + OpenFst.

The loader used by the product is the C++ one in cat_amd/csrc/fst_graph.cpp (``crf_graph_create``);
files written here go through it, so writer and loader check each other.
"""
import struct
from typing import Dict, Optional

import numpy as np

FST_MAGIC = 0x7EB2FDD6
PROP_EXPANDED_MUTABLE = 0x3


def write_fst(path: str, num_states: int, start: int, src, dst, ilabel, olabel, cost, final_cost,
              properties: int = PROP_EXPANDED_MUTABLE) -> None:
    """Write an OpenFst binary VectorFst<StdArc>.

    src/dst/ilabel/olabel/cost: per-arc arrays (any order; arcs are grouped by source state keeping
    their relative order).  cost / final_cost are TROPICAL costs (final_cost = +inf for non-final).
    ilabel 0 (epsilon) is rejected: the reference would index logits[-1] (fst_read.cc:55-56).
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int32)
    ilabel = np.asarray(ilabel, dtype=np.int32)
    olabel = np.asarray(olabel, dtype=np.int32)
    cost = np.asarray(cost, dtype=np.float32)
    final_cost = np.asarray(final_cost, dtype=np.float32)
    if len(final_cost) != num_states:
        raise ValueError("final_cost must have one entry per state")
    if len(src) and (ilabel.min() <= 0):
        raise ValueError("ilabel must be >= 1 (token id + 1); epsilon input labels are not allowed")
    order = np.argsort(src, kind="stable")
    counts = np.bincount(src, minlength=num_states).astype(np.int64)
    arc_dt = np.dtype([("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("nx", "<i4")])
    arcs = np.empty(len(src), dtype=arc_dt)
    arcs["il"], arcs["ol"], arcs["w"], arcs["nx"] = ilabel[order], olabel[order], cost[order], dst[order]
    offs = np.concatenate([[0], np.cumsum(counts)])
    with open(path, "wb") as f:
        f.write(struct.pack("<I", FST_MAGIC))
        for s in (b"vector", b"standard"):
            f.write(struct.pack("<i", len(s)) + s)
        # version 2, flags 0 (no symbol tables), properties, start, numstates, numarcs (0 = count per state)
        f.write(struct.pack("<iiQqqq", 2, 0, properties, start, num_states, 0))
        for s in range(num_states):
            f.write(struct.pack("<fq", float(final_cost[s]), int(counts[s])))
            f.write(arcs[offs[s]:offs[s + 1]].tobytes())


def synth_den_lm(vocab: int = 72, histories: int = 2048, fanout: int = 24, seed: int = 0,
                 path: Optional[str] = None) -> Dict:
    """Synthetic den_lm = CTC topology o random n-gram acceptor, built directly.

    G: ``histories`` LM states; history g is entered by exactly one token ``tok_in[g]``; it has
    ``fanout`` out-arcs to distinct tokens with log-softmax-normalised random weights (an extra
    softmax slot is the end-of-sentence mass -> final weight).  The CTC topology
    (build_ctc_topo.py:48-60: blank self-loop, token self-loop, token->blank, token->other token)
    is composed by hand: composed state = (history, last) with last in {blank, tok_in[history]}:
        state 0          : start  (history 0, nothing emitted yet)
        state 1 + 2g     : (g, blank)
        state 2 + 2g     : (g, tok_in[g])
    giving S = 2*histories + 1 and A ~= histories*(2*fanout + 3).  Every state is final.
    Returns the arc arrays in REFERENCE conventions (lab = ilabel-1, w = -cost, see fst_read.cc:43-59)
    plus S, A; writes an OpenFst binary to ``path`` when given.
    """
    V, H, d = int(vocab), int(histories), int(fanout)
    if not (2 <= V and 1 <= d <= V - 1 and H >= V - 1):
        raise ValueError("need vocab>=2, 1<=fanout<=vocab-1, histories>=vocab-1")
    rng = np.random.default_rng(seed)
    tok_in = np.concatenate([np.arange(1, V), rng.integers(1, V, size=H - (V - 1))]).astype(np.int32)
    rng.shuffle(tok_in)
    by_tok = [np.flatnonzero(tok_in == v) for v in range(V)]
    src, dst, lab, w = [], [], [], []
    final_w = np.zeros(2 * H + 1, dtype=np.float32)

    def sid(g, is_tok):
        return 1 + 2 * g + (1 if is_tok else 0)

    for g in range(H):
        toks = rng.choice(np.arange(1, V), size=d, replace=False)
        logit = rng.normal(0.0, 1.5, size=d + 1)
        logp = logit - np.log(np.exp(logit - logit.max()).sum()) - logit.max()
        nxt = np.array([by_tok[v][rng.integers(len(by_tok[v]))] for v in toks])
        last = int(tok_in[g])
        final_w[sid(g, False)] = final_w[sid(g, True)] = np.float32(logp[d])
        # (g, blank): blank self-loop; every LM arc
        src.append(sid(g, False)); dst.append(sid(g, False)); lab.append(0); w.append(0.0)
        for j in range(d):
            src.append(sid(g, False)); dst.append(sid(int(nxt[j]), True)); lab.append(int(toks[j])); w.append(logp[j])
        # (g, last): token self-loop, blank -> (g, blank), LM arcs except the repeated token
        src.append(sid(g, True)); dst.append(sid(g, True)); lab.append(last); w.append(0.0)
        src.append(sid(g, True)); dst.append(sid(g, False)); lab.append(0); w.append(0.0)
        for j in range(d):
            if int(toks[j]) != last:
                src.append(sid(g, True)); dst.append(sid(int(nxt[j]), True)); lab.append(int(toks[j])); w.append(logp[j])
        if g == 0:  # start state: same continuations as (0, blank); blank moves into (0, blank)
            final_w[0] = np.float32(logp[d])
            src.append(0); dst.append(sid(0, False)); lab.append(0); w.append(0.0)
            for j in range(d):
                src.append(0); dst.append(sid(int(nxt[j]), True)); lab.append(int(toks[j])); w.append(logp[j])
    src = np.asarray(src, dtype=np.int32)
    order = np.argsort(src, kind="stable")
    g = dict(
        S=2 * H + 1, A=len(src), start=0, vocab=V,
        src=src[order], dst=np.asarray(dst, dtype=np.int32)[order],
        lab=np.asarray(lab, dtype=np.int32)[order], w=np.asarray(w, dtype=np.float32)[order],
        start_w=np.full(2 * H + 1, -np.inf, dtype=np.float32), end_w=final_w,
    )
    g["start_w"][0] = 0.0
    if path is not None:
        write_fst(path, g["S"], 0, g["src"], g["dst"], g["lab"] + 1, g["lab"] + 1, -g["w"], -g["end_w"])
    return g


def random_labels_from_graph(g: Dict, length: int, rng: np.random.Generator) -> np.ndarray:
    """Walk the graph and return ``length`` emitted tokens (non-blank, CTC-collapsed), so that the
    numerator path has non-zero denominator mass (SURVEY.md 8d 'Synthetic inputs')."""
    src, dst, lab = g["src"], g["dst"], g["lab"]
    off = np.searchsorted(src, np.arange(g["S"] + 1))
    s, out, prev = int(g["start"]), [], -1
    while len(out) < length:
        k = rng.integers(off[s], off[s + 1])
        v = int(lab[k])
        if v != 0 and int(dst[k]) != s:  # a real token arc (not blank, not a self-loop)
            out.append(v)
            prev = v
        s = int(dst[k])
    del prev
    return np.asarray(out, dtype=np.int32)
