#!/usr/bin/env python
"""tools/column_block_study.py -- CPU study for round 4's verdict, item 5 ("large graphs: attack the gathers"): would LDS-resident COLUMN BLOCKS of the
state vector pay on the utterance-minor kernels?

The utterance-minor frame kernel (crf_batch_frame_kernel) reads every arc once per frame for the whole batch and gathers the source entry of each arc --
one 128-byte segment per arc and group of 32 utterances -- from L2: ~150 MB per launch at S = 16 385.  The proposal: renumber the states so that an arc's
source lies in a block of 1 280 states (x 32 utterances x 4 B = 160 KB of LDS), stage a block ONCE per workgroup, gather from LDS, stream the arcs block-major.
What that costs instead: a destination whose in-arcs come from k different blocks gets k partial sums from k workgroups, each a 128-byte read-modify-write
through L2 -- so the traffic that matters is  #(source block, destination) groups  against  #arcs,  in both directions, under the best renumbering one
can find.  This tool counts exactly that, on the hashed synthetic graph of bench.py's `large` point (no locality: the floor) and on a den_lm estimated from a
synthetic corpus (suffix locality: the realistic case), for the identity order, a BFS order and reverse Cuthill-McKee.

usage: python tools/column_block_study.py [block_states=1280]"""
import os
import sys
import tempfile

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import breadth_first_order, reverse_cuthill_mckee

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cat_amd import den_lm
from oracle import fst_io

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1280


def arcs_of(path):
    g = fst_io.read_fst(path)
    return int(g["S"]), np.asarray(g["src"], dtype=np.int64), np.asarray(g["dst"], dtype=np.int64)


def groups(src, dst, perm, nb):
    """(#groups forward: distinct (block of src, dst); #groups backward: distinct (block of dst, src)) under the renumbering perm[old] = new."""
    s, d = perm[src], perm[dst]
    fwd = len(np.unique((s // nb).astype(np.int64) * (1 << 32) + d))
    bwd = len(np.unique((d // nb).astype(np.int64) * (1 << 32) + s))
    return fwd, bwd


def orders(S, src, dst):
    m = coo_matrix((np.ones(len(src), dtype=np.int8), (src, dst)), shape=(S, S)).tocsr()
    sym = ((m + m.T) > 0).astype(np.int8).tocsr()
    ident = np.arange(S)
    o = breadth_first_order(sym, 0, directed=False, return_predecessors=False)
    rest = np.setdiff1d(ident, o, assume_unique=False)
    bfs = np.empty(S, dtype=np.int64); bfs[np.concatenate([o, rest])] = ident
    r = reverse_cuthill_mckee(sym, symmetric_mode=True)
    rcm = np.empty(S, dtype=np.int64); rcm[r] = ident
    return {"identity": ident, "BFS": bfs, "RCM": rcm}


def report(name, path):
    S, src, dst = arcs_of(path)
    A = len(src)
    indeg = np.bincount(dst, minlength=S)
    print(f"{name}: S = {S}, A = {A}, in-arcs per state {A / S:.1f} (max {indeg.max()}), blocks of {NB} states: {-(-S // NB)}")
    for k, perm in orders(S, src, dst).items():
        f, b = groups(src, dst, perm, NB)
        s, d = perm[src], perm[dst]
        same = float(np.mean(s // NB == d // NB))
        print(f"  {k:9s} groups / arcs: forward {f / A:.3f}, backward {b / A:.3f};  arcs inside one block {same:.3f}")
    return S, A


def main():
    tmp = tempfile.mkdtemp()
    p1 = os.path.join(tmp, "large.fst")
    den_lm.synth_den_lm(72, 8192, 32, 0, path=p1)                       # bench.py --histories 8192 --fanout 32: the `large` point
    report("hashed synthetic T o LM (bench.py `large`)", p1)
    # an estimated 4-gram den_lm of about the same size (tools/bench_fst.py's corpus generator, more text and more extra states)
    V = 72
    rng = np.random.default_rng(0)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(120000):
        L, s, a, b = int(rng.integers(10, 40)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); s.append(c); a, b = b, c
        seqs.append(s)
    p2 = os.path.join(tmp, "est.fst")
    den_lm.prep_den_lm(seqs, V, p2, 4, 3, 6000, selection="count")
    report("den_lm estimated from 120 000 sentences", p2)


if __name__ == "__main__":
    main()
