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
from typing import Dict, Iterable, Optional, Sequence, Tuple

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


# ------------------------------------------------------------------------------------------------
# den_lm tool-chain without Kaldi / OpenFst.  Replaces the pipeline of
# cat/utils/tool/prep_den_lm.sh:40-51:
#     corpus2index | chain-est-phone-lm --ngram-order=N --no-prune-ngram-order=M  ->  token_lm.fst
#     build_ctc_topo.py | fstcompile | fstarcsort                                 ->  T.fst
#     fstcompose T.fst token_lm.fst | fstdeterminizestar --use-log=true           ->  den_lm.fst
# PARITY UNPINNED: Kaldi is not in /root/reference, so its estimator cannot be run or read here.  What is
# reproduced is the CONTRACT the loss relies on: an epsilon-free, input-deterministic acceptor over
# ilabel = token + 1 (build_ctc_topo.py:6-11) in the tropical/log convention cost = -log p, whose language is
# (CTC topology) o (un-smoothed n-gram LM of the training transcripts with history-state pruning).
# ------------------------------------------------------------------------------------------------
def _select_by_likelihood(seqs, N: int, M: int, K: int):
    """Which histories of >= M tokens keep a state of their own: the greedy rule published for Kaldi's `chain-est-phone-lm`
    (LanguageModelEstimator: --no-prune-ngram-order, --num-extra-lm-states; cat/utils/tool/prep_den_lm.sh:40-44 calls it).  Every seen history
    starts as a state holding the next-token counts (0 = end of sentence) of the positions whose full history (up to N - 1 tokens) it is;
    a state's BACK-OFF state is its history without the oldest token.  While more than K states of >= M tokens are left, the state whose
    backing off LOSES THE LEAST training-data log-likelihood is merged into its back-off state -- only states that no remaining state backs
    off to (leaves), so the kept set stays closed under dropping the oldest token -- where, with c / n the state's counts and total and
    b / m its back-off state's (merged counts included),
        loss = sum_p c_p log(c_p / n) + sum_p b_p log(b_p / m) - sum_p (c_p + b_p) log((c_p + b_p) / (n + m))   >= 0.
    Ties: shorter history first, then token order.  Returns the kept set (all histories of < M tokens included)."""
    import heapq
    from collections import Counter
    own: Dict[tuple, Counter] = {(): Counter()}
    for s in seqs:
        for i in range(len(s) + 1):
            h = s[max(0, i - (N - 1)):i]
            own.setdefault(h, Counter())[s[i] if i < len(s) else 0] += 1
    for h in list(own):                                   # back-off states exist even where no position has exactly that history
        for k in range(1, len(h) + 1):
            own.setdefault(h[k:], Counter())
    active = set(own)
    nchild: Dict[tuple, int] = {h: 0 for h in active}
    kids: Dict[tuple, list] = {h: [] for h in active}
    for h in active:
        if h:
            nchild[h[1:]] += 1
            kids[h[1:]].append(h)

    def loglike(c):
        n = sum(c.values())
        return sum(v * np.log(v / n) for v in c.values()) if n else 0.0

    def loss(h):                                          # (rounded to 1e-9: equal losses computed along different paths must TIE, so that the documented order decides)
        c, b = own[h], own[h[1:]]
        return round(float(loglike(c) + loglike(b) - loglike(c + b)), 9)

    def prunable(h):
        return len(h) >= M and nchild[h] == 0

    heap = [(loss(h), len(h), h) for h in active if prunable(h)]
    heapq.heapify(heap)
    extra = sum(1 for h in active if len(h) >= M)
    while extra > K and heap:
        l0, _, h = heapq.heappop(heap)
        if h not in active or not prunable(h):
            continue
        l1 = loss(h)                                      # (its back-off state may have received counts since the entry was made)
        if l1 > l0 + 1e-12:
            heapq.heappush(heap, (l1, len(h), h))
            continue
        par = h[1:]
        own[par] = own[par] + own[h]
        active.discard(h)
        extra -= 1
        nchild[par] -= 1
        if prunable(par):
            heapq.heappush(heap, (loss(par), len(par), par))
        # the merged counts change what backing off into `par` costs its OTHER children -- and it can get CHEAPER (the back-off state's distribution has moved
        # towards theirs): fresh entries, so that the smallest CURRENT loss is always on top (an entry that has become too small is refreshed above when it is
        # popped; one that has become too large would otherwise sit behind states that cost more -- found by tests/test_den_lm_tools.py's naive restatement)
        for c in kids[par]:
            if c in active and prunable(c):
                heapq.heappush(heap, (loss(c), len(c), c))
    return active


def estimate_token_lm(seqs: Iterable[Sequence[int]], vocab_size: int, ngram_order: int = 4,
                      no_prune_ngram_order: int = 3, num_extra_states: int = 250, selection: str = "likelihood") -> Dict:
    """Un-smoothed n-gram LM over tokens 1..vocab_size-1 (0 = blank never occurs in transcripts), as a
    deterministic acceptor without epsilons -- the kind of LM `chain-est-phone-lm` builds for the denominator:
    a state is a history (tuple of up to ngram_order-1 previous tokens); all histories of fewer than
    ``no_prune_ngram_order`` tokens are kept, of the longer ones ``num_extra_states`` -- chosen by ``selection``:
    "likelihood" (default since round 6): Kaldi's published greedy rule, the states whose backing off would lose the most
    training-data log-likelihood (`_select_by_likelihood`), and EVERY history of fewer than ``no_prune_ngram_order`` tokens that
    occurs anywhere in a transcript (every seen bigram history at --no-prune-ngram-order=3: up to V^2 states -- what real den_lm
    files have); "count": the most frequent ones (rounds 1 - 5; that rule also kept a shorter history only where a transcript
    position has exactly that history, i.e. at sentence starts: fewer, higher-in-degree states -- kept for the records and tests
    made on such graphs) --; a position
    whose history is not kept is counted in its longest kept suffix.  Counts are collected by running the transcripts
    through that automaton, so every transition a transcript takes exists and every state's probabilities (tokens + end
    of sentence) sum to one.

    Returns dict(num_states, start, tok_in[g] (-1 for the start state), arcs[g] = [(token, next, logp)], final[g])."""
    seqs = [tuple(int(t) for t in s) for s in seqs]
    for s in seqs:
        for t in s:
            if not (1 <= t < vocab_size):
                raise ValueError(f"token {t} outside 1..{vocab_size - 1} (0 is the blank)")
    N = max(1, int(ngram_order))
    hist_count: Dict[tuple, int] = {}
    for s in seqs:
        for i in range(len(s) + 1):                      # history in front of token i (or of the end)
            h = s[max(0, i - (N - 1)):i]
            hist_count[h] = hist_count.get(h, 0) + 1
    keep = {()}
    for h in hist_count:
        if len(h) < max(1, no_prune_ngram_order):
            keep.add(h)
    if selection == "likelihood":
        keep |= _select_by_likelihood(seqs, N, max(1, no_prune_ngram_order), max(0, num_extra_states))
    elif selection == "count":
        extra = sorted((h for h in hist_count if len(h) >= max(1, no_prune_ngram_order)),
                       key=lambda h: (-hist_count[h], h))[:max(0, num_extra_states)]
        for h in extra:
            for k in range(len(h) + 1):                  # suffix-closed
                keep.add(h[k:])
    else:
        raise ValueError("selection must be 'likelihood' or 'count'")

    def lks(h):                                           # longest kept suffix
        h = h[max(0, len(h) - (N - 1)):]
        for k in range(len(h) + 1):
            if h[k:] in keep:
                return h[k:]
        return ()

    counts: Dict[tuple, Dict[int, int]] = {}
    for s in seqs:
        st = ()
        for t in s:
            counts.setdefault(st, {})
            counts[st][t] = counts[st].get(t, 0) + 1
            st = lks(st + (t,))
        counts.setdefault(st, {})
        counts[st][0] = counts[st].get(0, 0) + 1          # end of sentence
    states = sorted(counts, key=lambda h: (len(h), h))
    sid = {h: i for i, h in enumerate(states)}
    arcs, final, tok_in = [], [], []
    for h in states:
        tot = float(sum(counts[h].values()))
        a = []
        for t, c in sorted(counts[h].items()):
            if t == 0:
                continue
            a.append((t, sid[lks(h + (t,))], float(np.log(c / tot))))
        arcs.append(a)
        final.append(float(np.log(counts[h][0] / tot)) if counts[h].get(0, 0) else -np.inf)
        tok_in.append(h[-1] if h else -1)
    return dict(num_states=len(states), start=sid[()], tok_in=tok_in, arcs=arcs, final=final, histories=states)


def compose_ctc_topo(lm: Dict, vocab_size: int, path: Optional[str] = None) -> Dict:
    """CTC topology (build_ctc_topo.py:48-60: blank self-loop, token self-loop, token -> blank, token -> other
    token) composed with a deterministic, epsilon-free token LM (`estimate_token_lm`).  Composed states:
    (LM state g, last) with last in {nothing yet (start only), blank, the token just emitted}; an LM arc
    g --t--> g' becomes (g, *) --t--> (g', t), except from (g, t) itself (a repeated token needs a blank in
    between).  The result is input-deterministic and epsilon-free -- what fstdeterminizestar is called for in the
    reference -- and every state is entered with one label.
    Returns the arc arrays in reference conventions (see `synth_den_lm`); writes an OpenFst binary to ``path``."""
    g0 = lm["start"]
    ids: Dict[Tuple[int, int], int] = {}

    def sid(g, last):                                     # last: -1 = nothing emitted yet, 0 = blank, t >= 1 = token t
        k = (g, last)
        if k not in ids:
            ids[k] = len(ids)
        return ids[k]

    sid(g0, -1)
    src, dst, lab, w = [], [], [], []
    fin: Dict[int, float] = {}
    todo, seen = [(g0, -1)], {(g0, -1)}
    while todo:
        g, last = todo.pop()
        s = sid(g, last)
        fin[s] = lm["final"][g]

        def arc(d, label, weight):
            src.append(s); dst.append(sid(*d)); lab.append(label); w.append(weight)
            if d not in seen:
                seen.add(d); todo.append(d)
        if last >= 1:
            arc((g, last), last, 0.0)                     # token self-loop
        arc((g, 0), 0, 0.0)                               # blank (self-loop of (g, blank))
        for t, g2, lp in lm["arcs"][g]:
            if not (1 <= t < vocab_size):
                raise ValueError("LM token outside the vocabulary")
            if t == last:
                continue
            arc((g2, t), t, lp)
    S = len(ids)
    order = np.argsort(np.asarray(src), kind="stable")
    end_w = np.full(S, -np.inf, dtype=np.float32)
    for s, f in fin.items():
        end_w[s] = f
    g = dict(S=S, A=len(src), start=0, vocab=int(vocab_size),
             src=np.asarray(src, dtype=np.int32)[order], dst=np.asarray(dst, dtype=np.int32)[order],
             lab=np.asarray(lab, dtype=np.int32)[order], w=np.asarray(w, dtype=np.float32)[order],
             start_w=np.full(S, -np.inf, dtype=np.float32), end_w=end_w)
    g["start_w"][0] = 0.0
    if path is not None:
        cost_final = np.where(np.isfinite(g["end_w"]), -g["end_w"], np.inf).astype(np.float32)
        write_fst(path, S, 0, g["src"], g["dst"], g["lab"] + 1, g["lab"] + 1, -g["w"], cost_final)
    return g


def prep_den_lm(seqs: Iterable[Sequence[int]], vocab_size: int, path: str, ngram_order: int = 4,
                no_prune_ngram_order: int = 3, num_extra_states: int = 250, selection: str = "likelihood") -> Dict:
    """text (token ids) -> den_lm.fst: the whole of cat/utils/tool/prep_den_lm.sh without Kaldi/OpenFst."""
    lm = estimate_token_lm(seqs, vocab_size, ngram_order, no_prune_ngram_order, num_extra_states, selection)
    return compose_ctc_topo(lm, vocab_size, path)


def _main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="prepare the denominator LM for CRF training (prep_den_lm.sh without Kaldi): "
                                             "one transcript per line, space-separated token ids (1..vocab_size-1; an optional "
                                             "leading utterance id that is not an integer is skipped)")
    ap.add_argument("r_specifier", help="input text with token ids ('-' = stdin)")
    ap.add_argument("w_specifier", help="output den_lm.fst")
    ap.add_argument("--vocab-size", type=int, required=True, help="vocabulary size including the blank (id 0)")
    ap.add_argument("--ngram-order", type=int, default=4)
    ap.add_argument("--no-prune-ngram-order", type=int, default=3)
    ap.add_argument("--num-extra-lm-states", type=int, default=250)
    ap.add_argument("--state-selection", choices=["likelihood", "count"], default="likelihood",
                    help="which longer histories keep a state: by training-data log-likelihood (Kaldi's published rule) or by frequency")
    a = ap.parse_args(argv)
    import sys
    fh = sys.stdin if a.r_specifier == "-" else open(a.r_specifier)
    seqs = []
    for line in fh:
        f = line.split()
        if f and not f[0].lstrip("-").isdigit():
            f = f[1:]
        if f:
            seqs.append([int(x) for x in f])
    g = prep_den_lm(seqs, a.vocab_size, a.w_specifier, a.ngram_order, a.no_prune_ngram_order, a.num_extra_lm_states, a.state_selection)
    print(f"{a.w_specifier}: {g['S']} states, {g['A']} arcs from {len(seqs)} transcripts")


if __name__ == "__main__":
    _main()
