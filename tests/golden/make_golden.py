"""tests/golden/make_golden.py -- regenerates the committed golden fixtures.

Run from the repo root:  python tests/golden/make_golden.py
  * den_lm_fixture.fst : the reference's test graph (src/ctc_crf/test/den_lm.fst, 9 states / 24 arcs)
                         re-created FROM ITS TEXT LISTING with our own writer;
                         when /root/reference is present the script checks the bytes are identical.
  * kat_fixture.json   : known-answer for exactly the inputs of src/ctc_crf/test/main.py:15-28
                         (lamb = 0.01, N = 1) from the fp64 brute-force enumerator oracle/brute.py
                         (the reference's test asserts nothing, main.py:35 prints the loss).
  * kat_random.json    : brute-force answers for small random graphs / inputs (seeded), used to pin
                         oracle/crf_oracle.c independently of any DP code.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cat_amd.den_lm import write_fst  # noqa: E402
from oracle import fst_io  # noqa: E402
from oracle.brute import brute_ctc, brute_den  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
INF = float("inf")
LN3, LN15, LN2 = 1.0986123085021973, 0.40546509623527527, 0.6931471824645996

# (src, ilabel, olabel, cost, dst) -- SURVEY.md section 4 listing of test/den_lm.fst
FIXTURE_ARCS = [
    (0, 1, 0, 0.0, 0), (0, 2, 1, 0.0, 1),
    (1, 2, 0, 0.0, 1), (1, 1, 0, 0.0, 2), (1, 3, 2, LN3, 3), (1, 5, 4, LN15, 4),
    (2, 1, 0, 0.0, 2), (2, 3, 2, LN3, 3), (2, 5, 4, LN15, 4),
    (3, 3, 0, 0.0, 3), (3, 1, 0, 0.0, 5), (3, 2, 1, 0.0, 1),
    (4, 5, 0, 0.0, 4), (4, 1, 0, 0.0, 6), (4, 4, 3, LN2, 7),
    (5, 1, 0, 0.0, 5), (5, 2, 1, 0.0, 1),
    (6, 1, 0, 0.0, 6), (6, 4, 3, LN2, 7),
    (7, 4, 0, 0.0, 7), (7, 1, 0, 0.0, 8), (7, 2, 1, 0.0, 1),
    (8, 1, 0, 0.0, 8), (8, 2, 1, 0.0, 1),
]
FIXTURE_FINAL = [INF, INF, INF, INF, LN2, INF, LN2, INF, INF]
FIXTURE_PROBS = [  # src/ctc_crf/test/main.py:16-24 (probabilities; .log() is applied)
    [0.1, 0.1, 0.5, 0.1, 0.2],
    [0.5, 0.1, 0.1, 0.2, 0.2],
    [0.1, 0.7, 0.1, 0.05, 0.05],
    [0.6, 0.1, 0.1, 0.1, 0.1],
    [0.1, 0.1, 0.1, 0.6, 0.1],
]
FIXTURE_LABELS = [2, 1, 4]
FIXTURE_LAMB = 0.01


def make_fixture_fst(path):
    a = np.array(FIXTURE_ARCS, dtype=object)
    write_fst(path, 9, 0, [x[0] for x in FIXTURE_ARCS], [x[4] for x in FIXTURE_ARCS],
              [x[1] for x in FIXTURE_ARCS], [x[2] for x in FIXTURE_ARCS], [x[3] for x in FIXTURE_ARCS],
              FIXTURE_FINAL, properties=0x50002000003)
    del a


def random_graph(rng, S, V, arcs_per_state):
    src, dst, il, cost = [], [], [], []
    for s in range(S):
        for _ in range(arcs_per_state):
            src.append(s); dst.append(int(rng.integers(S))); il.append(int(rng.integers(V)) + 1)
            cost.append(float(rng.uniform(0.0, 2.0)))
    final = [float(rng.uniform(0.0, 1.5)) if rng.random() < 0.6 else INF for _ in range(S)]
    if all(f == INF for f in final):
        final[-1] = 0.5
    return src, dst, il, cost, final


def main():
    fst_path = os.path.join(HERE, "den_lm_fixture.fst")
    make_fixture_fst(fst_path)
    sha = hashlib.sha256(open(fst_path, "rb").read()).hexdigest()
    ref = "/root/reference/src/ctc_crf/test/den_lm.fst"
    if os.path.exists(ref):
        ref_sha = hashlib.sha256(open(ref, "rb").read()).hexdigest()
        assert ref_sha == sha, "re-created fixture differs from the reference's den_lm.fst"
        print("fixture bytes identical to", ref)
    g = fst_io.read_fst(fst_path)
    logits = np.log(np.array(FIXTURE_PROBS, dtype=np.float32)).astype(np.float64)  # fp32 .log() as main.py
    lz, gd = brute_den(g, logits)
    lp, gc = brute_ctc(logits, FIXTURE_LABELS)
    loss = lz - (1 + FIXTURE_LAMB) * lp
    grad = gd - (1 + FIXTURE_LAMB) * gc
    kat = dict(sha256_den_lm=sha, probs=FIXTURE_PROBS, labels=FIXTURE_LABELS, lamb=FIXTURE_LAMB,
               logZ_den=lz, logp_ctc=lp, loss=loss, grad=grad.tolist(), gamma_den=gd.tolist(),
               gamma_ctc=gc.tolist())
    json.dump(kat, open(os.path.join(HERE, "kat_fixture.json"), "w"), indent=1)
    print(f"fixture KAT: logZ={lz:.9f} logp={lp:.9f} loss={loss:.9f}")

    rng = np.random.default_rng(20250704)
    cases = []
    for ci in range(6):
        S, V, T = int(rng.integers(2, 6)), int(rng.integers(3, 6)), int(rng.integers(2, 7))
        if V ** T > 20000:
            T = 5
        src, dst, il, cost, final = random_graph(rng, S, V, int(rng.integers(2, 5)))
        p = os.path.join(HERE, f"rand{ci}.fst")
        write_fst(p, S, 0, src, dst, il, il, cost, final)
        gg = fst_io.read_fst(p)
        x = rng.normal(0, 2.0, size=(T, V))
        logits = (x - np.log(np.exp(x).sum(-1, keepdims=True))).astype(np.float32).astype(np.float64)
        L = int(rng.integers(0, min(3, T) + 1))
        labels = [int(v) for v in rng.integers(1, V, size=L)]
        lz, gd = brute_den(gg, logits)
        lp, gc = brute_ctc(logits, labels)
        cases.append(dict(fst=f"rand{ci}.fst", logits=logits.tolist(), labels=labels, logZ_den=lz,
                          logp_ctc=(lp if np.isfinite(lp) else None), gamma_den=gd.tolist(),
                          gamma_ctc=gc.tolist()))
    json.dump(cases, open(os.path.join(HERE, "kat_random.json"), "w"))
    print("wrote", len(cases), "random KATs")


if __name__ == "__main__":
    main()
