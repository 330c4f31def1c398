"""oracle/brute.py -- fp64 brute-force path enumerator, independent of any DP code.

TEST INFRASTRUCTURE ONLY.  Pins oracle/crf_oracle.c (see its header).  Enumerates every label
sequence pi in V^T:
  denominator: score(pi) = sum over accepting paths of the den graph reading pi of
               exp(sum arc weights + end weight) * prod_t exp(logits[t, pi_t])
               (the semantics of den_calculate.cu:75-119: alpha recursion + end weights + LSE)
  numerator:   CTC collapse B(pi) == labels, prob = prod_t exp(logits[t, pi_t])
               (the path-sum that gpu_ctc_kernels.h:87-213 computes by DP)
Posteriors gamma[t, v] = sum_{pi: pi_t = v} score(pi) / Z.  Only usable for V^T up to ~1e6.
"""
import itertools
import math

import numpy as np


def _collapse(pi, blank=0):
    out, prev = [], None
    for p in pi:
        if p != prev and p != blank:
            out.append(p)
        prev = p
    return out


def brute_den(g, logits):
    """g: dict from oracle.fst_io.read_fst; logits: [T,V] float64 log-probs.
    returns (logZ, gamma[T,V])."""
    T, V = logits.shape
    S = g["S"]
    out = [[] for _ in range(S)]
    for k in range(g["A"]):
        out[int(g["src"][k])].append((int(g["lab"][k]), int(g["dst"][k]), float(g["w"][k])))
    start = {s: float(g["start_w"][s]) for s in range(S) if np.isfinite(g["start_w"][s])}
    Z = 0.0
    gamma = np.zeros((T, V))
    for pi in itertools.product(range(V), repeat=T):
        # weighted set of states after reading pi (graphs may be non-deterministic)
        cur = {s: math.exp(w) for s, w in start.items()}
        for t, v in enumerate(pi):
            nxt = {}
            for s, m in cur.items():
                for (lab, d, w) in out[s]:
                    if lab == v:
                        nxt[d] = nxt.get(d, 0.0) + m * math.exp(w)
            cur = nxt
            if not cur:
                break
        if not cur:
            continue
        gsc = sum(m * math.exp(float(g["end_w"][s])) for s, m in cur.items() if np.isfinite(g["end_w"][s]))
        if gsc == 0.0:
            continue
        sc = gsc * math.exp(sum(logits[t, v] for t, v in enumerate(pi)))
        Z += sc
        for t, v in enumerate(pi):
            gamma[t, v] += sc
    return math.log(Z), gamma / Z


def brute_ctc(logits, labels, blank=0):
    """returns (log p(labels|x), gamma[T,V]); (-inf, zeros) if infeasible."""
    T, V = logits.shape
    Z = 0.0
    gamma = np.zeros((T, V))
    labels = list(labels)
    for pi in itertools.product(range(V), repeat=T):
        if _collapse(pi, blank) != labels:
            continue
        sc = math.exp(sum(logits[t, v] for t, v in enumerate(pi)))
        Z += sc
        for t, v in enumerate(pi):
            gamma[t, v] += sc
    if Z == 0.0:
        return -math.inf, gamma
    return math.log(Z), gamma / Z
