"""den_lm tool-chain without Kaldi/OpenFst (cat_amd/den_lm.py: estimate_token_lm, compose_ctc_topo, prep_den_lm;
SURVEY 8f-2, reference cat/utils/tool/prep_den_lm.sh:40-51).  The reference's own tools are not available, so these
tests pin the CONTRACT: a normalised, deterministic n-gram automaton that accepts every training transcript, and a
composed graph whose denominator (oracle, through the product's own FST reader) equals a brute-force sum over all
CTC paths of  p(path | x) * P_LM(collapse(path))."""
import itertools
import math
import os

import numpy as np
import pytest

import oracle
from cat_amd import den_lm
from oracle import fst_io


def _walk(lm, seq):
    g, lp = lm["start"], 0.0
    for t in seq:
        nxt = [(g2, w) for (tt, g2, w) in lm["arcs"][g] if tt == t]
        if not nxt:
            return -math.inf
        assert len(nxt) == 1                              # deterministic
        g, lp = nxt[0][0], lp + nxt[0][1]
    return lp + lm["final"][g]


def _corpus(seed, n, vocab, maxlen):
    rng = np.random.default_rng(seed)
    # a first-order source so that higher-order histories carry information
    trans = rng.dirichlet(np.ones(vocab - 1) * 0.5, size=vocab)
    out = []
    for _ in range(n):
        L, s, prev = int(rng.integers(1, maxlen + 1)), [], 0
        for _ in range(L):
            prev = 1 + int(rng.choice(vocab - 1, p=trans[prev]))
            s.append(prev)
        out.append(s)
    return out


@pytest.mark.parametrize("order,noprune,extra", [(1, 1, 0), (2, 2, 0), (3, 2, 4), (4, 3, 250)])
def test_token_lm_is_a_normalised_deterministic_automaton(order, noprune, extra):
    seqs = _corpus(order, 300, 7, 10)
    lm = den_lm.estimate_token_lm(seqs, 7, order, noprune, extra)
    for g in range(lm["num_states"]):
        toks = [t for t, _, _ in lm["arcs"][g]]
        assert len(toks) == len(set(toks))                # one arc per token
        tot = sum(math.exp(w) for _, _, w in lm["arcs"][g]) + (math.exp(lm["final"][g]) if math.isfinite(lm["final"][g]) else 0.0)
        assert abs(tot - 1.0) < 1e-9
        for t, g2, _ in lm["arcs"][g]:
            assert order == 1 or lm["tok_in"][g2] == t    # a (non-empty) history is entered by its last token
            assert len(lm["histories"][g2]) <= max(0, order - 1)
    for s in seqs:                                        # every training transcript is accepted
        assert math.isfinite(_walk(lm, s))
    # the probabilities of all strings up to some length sum to <= 1 and the mass of the training set is substantial
    if order == 1:
        assert lm["num_states"] == 1


def test_state_pruning():
    seqs = _corpus(5, 500, 6, 12)
    full = den_lm.estimate_token_lm(seqs, 6, 4, 3, 10 ** 6)
    small = den_lm.estimate_token_lm(seqs, 6, 4, 3, 3)
    assert small["num_states"] < full["num_states"]
    assert all(len(h) <= 3 for h in small["histories"]) and () in set(small["histories"])
    assert sum(len(h) == 3 for h in small["histories"]) <= 3      # at most num_extra_states full-order histories
    # more context can only raise the training likelihood of an ML-estimated n-gram
    ll_full = sum(_walk(full, s) for s in seqs)
    ll_small = sum(_walk(small, s) for s in seqs)
    assert ll_full >= ll_small - 1e-9


def test_composed_graph_equals_brute_force_sum_over_ctc_paths(tmp_path):
    V, T = 4, 5
    seqs = _corpus(2, 200, V, 4)
    lm = den_lm.estimate_token_lm(seqs, V, 3, 2, 50)
    p = str(tmp_path / "den_lm.fst")
    g = den_lm.prep_den_lm(seqs, V, p, 3, 2, 50)
    assert os.path.getsize(p) > 0
    gr = fst_io.read_fst(p)                               # what the loss would load
    assert gr["S"] == g["S"] and gr["A"] == g["A"]
    rng = np.random.default_rng(0)
    x = rng.normal(size=(1, T, V)).astype(np.float32)
    logp = x - np.log(np.exp(x).sum(-1, keepdims=True))
    ref = float(oracle.den(gr, logp, np.array([T], dtype=np.int32), precision="f64")[1][0])   # costs_alpha = logZ
    tot = 0.0
    for pi in itertools.product(range(V), repeat=T):
        seq, prev = [], -1
        for v in pi:                                      # CTC collapse: merge repeats, drop blanks
            if v != prev and v != 0:
                seq.append(v)
            prev = v
        lw = _walk(lm, seq)
        if math.isfinite(lw):
            tot += math.exp(lw + sum(float(logp[0, t, v]) for t, v in enumerate(pi)))
    assert abs(ref - math.log(tot)) <= 1e-6 * max(1.0, abs(ref))


def test_cli_and_bad_input(tmp_path):
    txt = tmp_path / "text.ids"
    txt.write_text("utt1 1 2 3 2\nutt2 2 2 1\n3 1\n")
    out = tmp_path / "den.fst"
    den_lm._main([str(txt), str(out), "--vocab-size", "4", "--ngram-order", "3", "--no-prune-ngram-order", "2"])
    g = fst_io.read_fst(str(out))
    assert g["S"] > 2 and int(g["lab"].max()) <= 3
    with pytest.raises(ValueError):
        den_lm.estimate_token_lm([[1, 4]], 4)             # token outside the vocabulary
    with pytest.raises(ValueError):
        den_lm.estimate_token_lm([[0, 1]], 4)             # the blank is not a transcript token


def test_factoring_survives_openfst_style_rewrites(tmp_path):
    """den_lm files come out of ``fstcompose | fstdeterminizestar --use-log=true`` (cat/utils/tool/prep_den_lm.sh:48-49):
    state numbers and arc order are whatever those tools leave.  The graph compiler must find the T o LM structure
    (factored layout: one CU per recursion) under ANY numbering and arc order, and in a weight-pushed graph too (potentials
    moved along the arcs -- determinizestar does not do this to an input-deterministic graph, but other OpenFst tools
    would): fst_graph.cpp regauge_pushed.  Host-only compile: no GPU needed."""
    from cat_amd.ctc_crf import _C
    from tests.util import crf_env, transform_graph

    def stats(path):
        h = _C.compile_graph_host_only(path)
        s = _C.graph_stats(h)
        _C._lib.crf_graph_destroy(_C._vp(h))
        return s

    p = str(tmp_path / "orig.fst")
    g = den_lm.synth_den_lm(72, 512, 12, seed=3, path=p)
    base = stats(p)
    assert base["fac"] == 1
    for seed in (1, 2):
        q = str(tmp_path / f"renum{seed}.fst")
        transform_graph(g, q, seed=seed, renumber=True, reorder=True)
        s = stats(q)
        assert s["fac"] == 1 and s["fac_matched_pairs"] == base["fac_matched_pairs"]
        assert s["fac_fwd_slots"] == base["fac_fwd_slots"] and s["fac_bwd_slots"] == base["fac_bwd_slots"]
    # a weight-pushed graph: the compiler re-gauges it (its own potentials undo the constant log-weight difference between
    # the two states of a history) and the factored layout is back, same size; CRF_NO_REGAUGE=1: generic layout
    q = str(tmp_path / "pushed.fst")
    transform_graph(g, q, seed=5, renumber=True, reorder=True, push=True)
    s = stats(q)
    assert s["S"] == base["S"] and s["A"] == base["A"]
    assert s["regauged"] == 1 and s["fac"] == 1 and s["fac_matched_pairs"] == base["fac_matched_pairs"]
    assert s["fac_fwd_slots"] == base["fac_fwd_slots"] and s["fac_bwd_slots"] == base["fac_bwd_slots"]
    assert base["regauged"] == 0
    with crf_env(CRF_REGAUGE_MINHASH=1):                    # the near-linear search graphs of millions of arcs take: same result
        s2 = stats(q)
    # (two states that share d rows collide in one of 12 min-hashes with probability 1 - (2 / (d + 2))^12: histories with one or
    # two successors may be missed -- they then simply do not factor; exact for the graphs the search is meant for)
    assert s2["regauged"] == 1 and s2["fac"] == 1 and 0.98 * base["fac_matched_pairs"] <= s2["fac_matched_pairs"] <= base["fac_matched_pairs"]
    with crf_env(CRF_NO_REGAUGE=1):
        s = stats(q)
    assert s["regauged"] == 0 and (s["fac"] == 0 or s["fac_matched_pairs"] < base["fac_matched_pairs"]) and (s["fac"] == 1 or s["res_K"] >= 1)


def test_reference_fixture_layout(golden_dir):
    """The only Kaldi-made den_lm available (the reference's 9-state test graph, byte-identical copy): which layout it
    takes is deterministic -- the factored one (4 (tail, main) pairs)."""
    from cat_amd.ctc_crf import _C
    h = _C.compile_graph_host_only(os.path.join(golden_dir, "den_lm_fixture.fst"))
    s = _C.graph_stats(h)
    _C._lib.crf_graph_destroy(_C._vp(h))
    assert s["S"] == 9 and s["fac"] == 1 and s["fac_matched_pairs"] == 4


def test_factored_rows_of_the_utterance_minor_kernels(tmp_path, golden_dir):
    """T o LM graphs get FACTORED rows for the utterance-minor kernels (fst_graph.cpp build_batch_factored: an entry U = a[tail] +
    a[main] per couple of states, the tail row folded into the main row, one backward row per couple).  Host check
    (crf_debug_facbatch_check): one step of both recursions through the factored rows equals the step through the plain tables
    on random vectors, every one-pair row is produced exactly once, descriptors carry the plain tables' pairs and labels.
    Records halve; graphs without the structure get none."""
    from cat_amd.ctc_crf import _C

    def check(path):
        h = _C.compile_graph_host_only(path)
        r = _C.debug_facbatch_check(h)
        st = _C.graph_stats(h)
        _C._lib.crf_graph_destroy(_C._vp(h))
        return st, r

    p = str(tmp_path / "tolm.fst")
    den_lm.synth_den_lm(72, 300, 10, seed=1, path=p)
    st, r = check(p)
    r0 = r
    assert r["NU"] == 299 and r["arcs"] == st["A"]               # every history but the start's is a couple
    assert r["fwd_records"] < 0.55 * st["A"] and r["bwd_records"] < 0.55 * st["A"]
    st, r = check(os.path.join(golden_dir, "den_lm_fixture.fst"))   # the reference's own 9-state graph: 4 couples
    assert r["NU"] == 4 and r["fwd_records"] < st["A"] and r["bwd_records"] < st["A"]
    for i in range(6):                                           # random general graphs: no couples, no factored rows
        st, r = check(os.path.join(golden_dir, f"rand{i}.fst"))
        assert r["NU"] == 0 and r["fwd_records"] == 0
    # the streams cut from the factored rows, for every group size: records = the rows' records, flags, three descriptor words
    # per row, at most 8 / 8 / 4 / 2 bundles per task
    h = _C.compile_graph_host_only(p)
    for UL in (8, 16, 32, 64):
        s4 = _C.debug_stream_check(h, -UL, 64)
        assert s4["arc_records"] == r0["fwd_records"] + r0["bwd_records"]   # every record of every factored row exactly once
        assert s4["rest_rows"] <= 2 and s4["tasks"] >= 2
    _C._lib.crf_graph_destroy(_C._vp(h))
    from tests.util import transform_graph                       # a renumbered, reordered, weight-pushed copy (re-gauged by the compiler)
    from oracle import fst_io
    g = fst_io.read_fst(p)
    q = str(tmp_path / "pushed.fst")
    transform_graph(g, q, seed=3, renumber=True, reorder=True, push=True)
    st, r = check(q)
    assert st["regauged"] == 1 and r["NU"] == 299


@pytest.mark.parametrize("UL", [8, 16, 32, 64])
def test_arc_streams_of_the_utterance_minor_kernels(tmp_path, golden_dir, UL):
    """The arc streams the utterance-minor kernels walk (fst_graph.cpp: build_stream_host) are built on the HOST for several
    graphs and checked record by record against the graph's row tables (crf_debug_stream_check): every row with one entering
    pair once, its records = its arcs in order, end-of-bundle flags, at most 8 bundles per task; the other rows in the rest
    list.  Graphs: T o LM synthetic, an estimated n-gram graph with rows of hundreds of arcs, random general graphs (states
    entered with several labels), the reference's 9-state fixture."""
    import json
    from cat_amd.ctc_crf import _C

    def check(path, want):
        h = _C.compile_graph_host_only(path)
        st = _C.graph_stats(h)
        r = _C.debug_stream_check(h, UL, want)
        _C._lib.crf_graph_destroy(_C._vp(h))
        return st, r

    p = str(tmp_path / "tolm.fst")
    den_lm.synth_den_lm(72, 300, 10, seed=1, path=p)
    st, r = check(p, 64)
    # both directions: every arc is a record exactly once, except the arcs of the rows that are not in the streams (the start
    # state: nobody enters it, so it is a per-row case in both directions -- its out-arcs are missing from the backward stream)
    assert r["rest_rows"] <= 2 and 2 * st["A"] - st["max_out_deg"] <= r["arc_records"] <= 2 * st["A"] and r["tasks"] >= 2
    assert r["steps"] * (256 // UL) * 1 >= r["arc_records"] // 2             # (padding only adds)
    st, r2 = check(p, 100000)                                    # more tasks wanted than bundles exist: still whole bundles
    assert r2["arc_records"] == r["arc_records"] and r2["tasks"] >= r["tasks"]   # (a task has at least 64 steps)
    V = 40                                                       # (the corpus of test_factored_layout_takes_an_estimated_ngram_graph)
    rng = np.random.default_rng(3)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(1200):
        sq, a, b = [], 0, 0
        for _ in range(int(rng.integers(8, 30))):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b]))
            sq.append(c)
            a, b = b, c
        seqs.append(sq)
    q = str(tmp_path / "est.fst")
    den_lm.prep_den_lm(seqs, V, q, 4, 3, 150, selection="count")
    st, r = check(q, 16)
    assert 2 * st["A"] - st["max_out_deg"] <= r["arc_records"] <= 2 * st["A"] and st["max_in_deg"] > 64
    for c in json.load(open(os.path.join(golden_dir, "kat_random.json"))):
        st, r = check(os.path.join(golden_dir, c["fst"]), 4)
        assert r["arc_records"] <= 2 * st["A"]
    st, r = check(os.path.join(golden_dir, "den_lm_fixture.fst"), 4)
    assert r["arc_records"] > 0


def test_estimator_on_a_hand_worked_count_table():
    """The estimator's rule, pinned on a table worked by hand (cat/utils/tool/prep_den_lm.sh:40-51 calls Kaldi's
    ``chain-est-phone-lm --ngram-order=N --no-prune-ngram-order=M --num-extra-lm-states=K``, which is not in this image:
    equivalence with it stays UNPINNED).  order N = 3, no-prune order M = 2, K = 2 extra states, tokens {1, 2, 3}:

        transcripts            1 2 3 | 1 2 3 | 1 2 1 | 2 3 | 3 1 2 3 | 2 1

    1. Histories (up to N - 1 = 2 tokens) in front of every position, the end included, with their frequencies
         ():6  (1):4  (2):2  (3):1  (1,2):4  (2,3):4  (2,1):2  (3,1):1
    2. Kept states (selection="count"; the default rule is pinned by the next test): every history shorter than M = 2 tokens -- (), (1), (2), (3) -- plus the K = 2 most frequent longer ones
       (ties in lexicographic order): (1,2) and (2,3); suffix-closed already.  (2,1) and (3,1) fall back to their longest kept
       suffix (1).
    3. Counts collected by running the transcripts through that automaton (0 = end of sentence):
         ()    : 1 x3, 2 x2, 3 x1                  total 6
         (1)   : 2 x4 (from 1 2 3, 1 2 3, 1 2 1 and 3 1 2 3: the last one arrives in (1) from (3,1)), end x2 (1 2 1 and 2 1)   total 6
         (2)   : 1 x1 (2 1), 3 x1 (2 3)            total 2
         (3)   : 1 x1                              total 1
         (1,2) : 3 x3, 1 x1                        total 4
         (2,3) : end x4                            total 4
    4. Maximum-likelihood probabilities, no smoothing, no back-off arcs; a token leads to the longest kept suffix of
       (history + token): 1 after (1,2) -> (2,1) is not kept -> state (1); 3 after (1,2) -> (2,3)."""
    seqs = [(1, 2, 3), (1, 2, 3), (1, 2, 1), (2, 3), (3, 1, 2, 3), (2, 1)]
    lm = den_lm.estimate_token_lm(seqs, 4, ngram_order=3, no_prune_ngram_order=2, num_extra_states=2, selection="count")   # (the frequency rule of rounds 1 - 5)
    assert lm["histories"] == [(), (1,), (2,), (3,), (1, 2), (2, 3)] and lm["start"] == 0 and lm["num_states"] == 6
    L = math.log
    want_arcs = [
        [(1, 1, L(3 / 6)), (2, 2, L(2 / 6)), (3, 3, L(1 / 6))],     # ()
        [(2, 4, L(4 / 6))],                                         # (1)
        [(1, 1, L(1 / 2)), (3, 5, L(1 / 2))],                       # (2)
        [(1, 1, 0.0)],                                              # (3)
        [(1, 1, L(1 / 4)), (3, 5, L(3 / 4))],                       # (1,2)
        [],                                                         # (2,3)
    ]
    want_final = [-math.inf, L(2 / 6), -math.inf, -math.inf, -math.inf, 0.0]
    for got, want in zip(lm["arcs"], want_arcs):
        assert [(t, n) for t, n, _ in got] == [(t, n) for t, n, _ in want]
        assert np.allclose([w for _, _, w in got], [w for _, _, w in want], rtol=0, atol=1e-12)
    assert np.allclose(lm["final"], want_final, rtol=0, atol=1e-12)
    assert lm["tok_in"] == [-1, 1, 2, 3, 2, 3]
    # every transcript is accepted with exactly the product of those probabilities
    for s in seqs:
        st, lp = 0, 0.0
        for t in s:
            (nxt, w), = [(n, w) for tt, n, w in lm["arcs"][st] if tt == t]
            st, lp = nxt, lp + w
        assert math.isfinite(lp + lm["final"][st])


def test_estimator_selects_states_by_likelihood_like_kaldi():
    """The DEFAULT state selection since round 6: Kaldi's published greedy rule for `chain-est-phone-lm
    --no-prune-ngram-order=M --num-extra-lm-states=K` (cat/utils/tool/prep_den_lm.sh:40-44) -- of the histories of >= M tokens, back off the one
    whose merging into its back-off state (the history without its oldest token) loses the LEAST training-data log-likelihood, until K are left --
    on a table worked by hand on which it DIFFERS from the frequency rule.  (No Kaldi binary or source is in this image: this pins the published
    rule as implemented here, not equality with a Kaldi run.)  order N = 3, M = 2, tokens {1, 2, 3, 4}:

        transcripts     1 2 4  (x 4)   |   3 2 1  (x 2)   |   2 4  (x 3)

    1. Next-token counts by FULL history (up to two tokens; 0 = end of sentence):
         ():    1 x4, 3 x2, 2 x3          (1): 2 x4        (2): 4 x3  (only "2 4" has a position whose whole history is (2))        (3): 2 x2
         (4):   none (4 never starts a transcript; the state exists as the back-off state of (2,4))
         (1,2): 4 x4        (3,2): 1 x2        (2,4): end x7        (2,1): end x2
    2. Frequency rule: the two-token histories by count -- (2,4): 7, (1,2): 4, (2,1): 2, (3,2): 2 -- K = 1 keeps (2,4), K = 2 keeps (2,4) and (1,2).
    3. Likelihood rule, loss of backing off c (total n) into b (total m) = sum c log(c/n) + sum b log(b/m) - sum (c+b) log((c+b)/(n+m)):
         (1,2) {4: 4} into (2) {4: 3}:       both distributions are "always 4": loss 0
         (2,4) {end: 7} into (4) {}:          an empty back-off state takes the counts as they are: loss 0
         (3,2) {1: 2} into (2) {4: 3}:        -(3 log(3/5) + 2 log(2/5)) = 3.365
         (2,1) {end: 2} into (1) {2: 4}:      -(4 log(4/6) + 2 log(2/6)) = 3.819
       Four states, K = 1: three are backed off, cheapest first: (1,2) and (2,4) at loss 0 -- now (2) = {4: 7}, (4) = {end: 7} -- then (3,2) costs
       -(7 log(7/9) + 2 log(2/9)) = 4.767 against (2,1)'s 3.819: (2,1) goes, (3,2) STAYS.  K = 2 stops one step earlier: (2,1) and (3,2) stay.
       The frequent histories are the ones whose predictions their suffix makes just as well; the rare (3,2) is the one that changes a prediction.
    4. K = 1: states (), (1), (2), (3), (4), (3,2); counts by running the transcripts through that automaton:
         ():    1 x4, 2 x3, 3 x2  (9)     (1): 2 x4, end x2  (6: "3 2 1" arrives in (1) from (3,2) and ends)     (2): 4 x7  ("1 2 4": (1,2) is not kept)
         (3):   2 x2 -> (3,2)             (4): end x7                                                         (3,2): 1 x2 -> (1)"""
    seqs = [(1, 2, 4)] * 4 + [(3, 2, 1)] * 2 + [(2, 4)] * 3
    kept = lambda K, sel: den_lm.estimate_token_lm(seqs, 5, ngram_order=3, no_prune_ngram_order=2, num_extra_states=K, selection=sel)["histories"]
    assert kept(1, "count") == [(), (1,), (2,), (3,), (2, 4)] and kept(2, "count") == [(), (1,), (2,), (3,), (1, 2), (2, 4)]
    assert kept(1, "likelihood") == [(), (1,), (2,), (3,), (4,), (3, 2)] and kept(2, "likelihood") == [(), (1,), (2,), (3,), (4,), (2, 1), (3, 2)]
    # the losses of step 3, through the function the estimator uses
    L = math.log
    assert abs(-(3 * L(3 / 5) + 2 * L(2 / 5)) - 3.365) < 1e-3 and abs(-(4 * L(4 / 6) + 2 * L(2 / 6)) - 3.819) < 1e-3 and abs(-(7 * L(7 / 9) + 2 * L(2 / 9)) - 4.767) < 1e-3
    lm = den_lm.estimate_token_lm(seqs, 5, ngram_order=3, no_prune_ngram_order=2, num_extra_states=1)      # the default IS the likelihood rule
    assert lm["histories"] == [(), (1,), (2,), (3,), (4,), (3, 2)]
    want_arcs = [
        [(1, 1, L(4 / 9)), (2, 2, L(3 / 9)), (3, 3, L(2 / 9))],     # ()
        [(2, 2, L(4 / 6))],                                         # (1)
        [(4, 4, 0.0)],                                              # (2)
        [(2, 5, 0.0)],                                              # (3)
        [],                                                         # (4)
        [(1, 1, 0.0)],                                              # (3,2)
    ]
    want_final = [-math.inf, L(2 / 6), -math.inf, -math.inf, 0.0, -math.inf]
    for got, want in zip(lm["arcs"], want_arcs):
        assert [(t, n) for t, n, _ in got] == [(t, n) for t, n, _ in want]
        assert np.allclose([w for _, _, w in got], [w for _, _, w in want], rtol=0, atol=1e-12)
    assert np.allclose(lm["final"], want_final, rtol=0, atol=1e-12)
    # both rules: normalised, deterministic, every transcript accepted (the contract the loss relies on)
    for sel in ("likelihood", "count"):
        m = den_lm.estimate_token_lm(seqs, 5, 3, 2, 2, selection=sel)
        for g in range(m["num_states"]):
            tot = sum(math.exp(w) for _, _, w in m["arcs"][g]) + (math.exp(m["final"][g]) if math.isfinite(m["final"][g]) else 0.0)
            assert abs(tot - 1.0) < 1e-12 and len({t for t, _, _ in m["arcs"][g]}) == len(m["arcs"][g])
        for sq in seqs:
            st = m["start"]
            for t in sq:
                (st,) = [n for tt, n, _ in m["arcs"][st] if tt == t]
            assert math.isfinite(m["final"][st])


@pytest.mark.parametrize("seed", range(6))
def test_likelihood_selection_against_a_naive_greedy(seed):
    """`_select_by_likelihood` keeps a heap with lazily refreshed losses; here the same published rule restated the slow way -- after every merge recompute the loss of
    EVERY remaining leaf from scratch and take the smallest (ties: shorter history, then token order) -- on random corpora, for every budget K from all states
    down to none.  The two must keep the same set at every K; the sets are nested (K - 1 is K minus one state); and the training-data log-likelihood of the
    estimated LM never rises when a state is taken away."""
    from collections import Counter
    rng = np.random.default_rng(100 + seed)
    V, N, M = int(rng.integers(3, 7)), int(rng.integers(3, 5)), int(rng.integers(1, 3))
    trans = rng.dirichlet(np.ones(V - 1) * 0.4, size=(V, V))
    seqs = []
    for _ in range(int(rng.integers(15, 60))):
        s, a, b = [], 0, 0
        for _ in range(int(rng.integers(1, 9))):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); s.append(c); a, b = b, c
        seqs.append(tuple(s))

    def ll(c):
        n = sum(c.values())
        return sum(v * math.log(v / n) for v in c.values()) if n else 0.0

    def naive(K):
        own = {(): Counter()}
        for s in seqs:
            for i in range(len(s) + 1):
                own.setdefault(s[max(0, i - (N - 1)):i], Counter())[s[i] if i < len(s) else 0] += 1
        for h in list(own):
            for k in range(1, len(h) + 1):
                own.setdefault(h[k:], Counter())
        active = set(own)
        while sum(len(h) >= M for h in active) > K:
            leaves = [h for h in active if len(h) >= M and not any(g[1:] == h for g in active if g)]
            h = min(leaves, key=lambda h: (round(ll(own[h]) + ll(own[h[1:]]) - ll(own[h] + own[h[1:]]), 9), len(h), h))
            own[h[1:]] = own[h[1:]] + own[h]
            active.discard(h)
        return active

    total = len(naive(10 ** 9))
    n_long = sum(len(h) >= M for h in naive(10 ** 9))
    prev, prev_ll = None, None
    for K in range(n_long, -1, -1):
        want = naive(K)
        got = den_lm._select_by_likelihood(seqs, N, M, K)
        assert got == want, (K, sorted(got ^ want))
        assert sum(len(h) >= M for h in got) == K
        if prev is not None:
            assert got < prev and len(prev) - len(got) == 1
        prev = got
        lm = den_lm.estimate_token_lm(seqs, V, N, M, K)
        cur = sum(_walk(lm, s) for s in seqs)
        assert math.isfinite(cur) and (prev_ll is None or cur <= prev_ll + 1e-9)
        prev_ll = cur
    assert total >= n_long
