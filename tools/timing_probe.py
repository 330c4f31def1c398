"""tools/timing_probe.py -- decode the in-kernel phase stamps of a TIMING build (see crf_kernels.hip CRF_TM).

    CRF_BUILD_DEFS=-DCRF_TIMING python -m cat_amd.build --force     # here (cross-compile)
    gpurun -- 'python tools/timing_probe.py'                        # on the GPU box
    python -m cat_amd.build --force                                 # back to the product build

Prints, in shader cycles, the median duration of every phase of a frame for the resident denominator
chains (utterance 3, frames 100..227, every CU of the recursion) and for one workgroup of the grad pass.
"""
import os
import statistics
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctc_crf  # noqa: E402
from cat_amd.ctc_crf import _C  # noqa: E402
from cat_amd.den_lm import synth_den_lm  # noqa: E402
from cat_amd.synth import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, T, V = 64, 1500, 72
    fst = os.path.join(tempfile.mkdtemp(prefix="crfprobe_"), "den_lm.fst")
    if len(sys.argv) > 2 and sys.argv[1] == "est":   # `est <sentences>`: a den_lm ESTIMATED with the default (Kaldi's) rule, as tools/bench_fst.py makes it
        import numpy as np
        from cat_amd import den_lm
        from oracle import fst_io
        rng = np.random.default_rng(0)
        trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
        seqs = []
        for _ in range(int(sys.argv[2])):
            s, a, b = [], 0, 0
            for _ in range(int(rng.integers(10, 40))):
                c = 1 + int(rng.choice(V - 1, p=trans[a, b])); s.append(c); a, b = b, c
            seqs.append(s)
        den_lm.prep_den_lm(seqs, V, fst, 4, 3, 250, selection="likelihood")
        g = fst_io.read_fst(fst)
        print(f"estimated den_lm from {sys.argv[2]} sentences: S={g['S']} A={g['A']}")
    else:
        H, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 24)   # histories, fan-out of the synthetic den_lm
        g = synth_den_lm(V, H, D, seed=0, path=fst)
        print(f"den_lm H={H} d={D}: S={g['S']} A={g['A']}")
    ctx = ctc_crf.CRFContext(fst, 0)  # noqa: F841
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=0, ragged=False)
    x = torch.tensor(logits, device=dev, requires_grad=True)
    crit = ctc_crf.CTC_CRF_LOSS(lamb=0.1)
    for _ in range(3):
        x.grad = None
        crit(x, torch.tensor(labels), torch.tensor(lx), torch.tensor(ly)).backward()
    torch.cuda.synchronize()
    tm = _C.timing_read()
    if not tm:
        print("not a timing build")
        return
    fac = bool(_C.graph_stats(_C.graph_for(dev))["fac"])
    names = ["frame top + tail rows", "main rows (gathers+epilogues)", "wave max + emission stage", "barrier"] if fac else \
        ["compute(gathers+epilogues)", "fetch peers", "wave max + emission stage", "barrier"]
    for d, dn in enumerate(("fwd", "bwd")):
        for k in range(2):
            base = (d * 4 + k) * 1024
            rows = [[tm[base + f * 8 + j] for j in range(5)] for f in range(128)]
            if not rows[0][0]:
                continue
            print(f"den {dn} CU {k}:")
            for j, nm in enumerate(names):
                v = [r[j + 1] - r[j] for r in rows]
                print(f"   {nm:32s} median {statistics.median(v):8.0f}  p90 {sorted(v)[int(.9 * len(v))]:8.0f} cyc")
            v = [rows[f + 1][0] - rows[f][0] for f in range(127)]
            print(f"   {'frame total':32s} median {statistics.median(v):8.0f} cyc")
    print("per-wave compute time (cycles, median of 8 frames) vs chunks / slices of the wave:")
    nw = {4: 16, 2: 8}.get(_C.graph_stats(_C.graph_for(dev)).get("fac_geom"), 12) if fac else 8   # 1024 / 512 / 768 threads
    for d, dn in enumerate(("fwd", "bwd")):
        for k in range(1 if fac else 2):
            line = []
            st = [tm[12288 + 1024 + (d * 4 + k) * 8 + f] for f in range(8)]
            for w in range(nw):
                o = 12288 + ((d * 4 + k) * 8 + w) * 16
                v = [tm[o + f] - st[f] for f in range(8)]
                line.append(f"w{w}: {statistics.median(v):5.0f} ({tm[o + 8]:2d}ch,{tm[o + 9]:2d}sl)")
            print(f"   den {dn} CU {k}: " + "  ".join(line))
            if fac:   # arrival at the frame barrier (after the wave maximum and, in the waves that hold emissions, their staging)
                line = []
                for w in range(nw):
                    v = [tm[15360 + (d * 16 + w) * 8 + f] - st[f] for f in range(8)]
                    line.append(f"w{w}: {statistics.median(v):5.0f}")
                print(f"   den {dn} arrival at the barrier: " + "  ".join(line))
    rows = [[tm[14336 + f * 8 + j] for j in range(6)] for f in range(128)]
    if rows[0][0]:
        print("ctc forward, utterance 3, wave 0:")
        for j, nm in enumerate(["emissions (wait batch + exp)", "issue next batch", "max + recurrence + stores", "wave max", "barrier"]):
            v = [r[j + 1] - r[j] for r in rows]
            print(f"   {nm:32s} median {statistics.median(v):8.0f}  p90 {sorted(v)[int(.9 * len(v))]:8.0f}  max {max(v):8.0f} cyc")
        v = [rows[f + 1][0] - rows[f][0] for f in range(127)]
        print(f"   {'frame total':32s} median {statistics.median(v):8.0f}  mean {statistics.mean(v):8.0f} cyc")
    base = 8192
    rows = [[tm[base + f * 8 + j] for j in range(7)] for f in range(15)]
    gn = ["issue prefetch", "gather/reduce", "barrier 1", "stage next rows (vm wait)", "epilogue+stores", "barrier 2"]
    print("grad den, one workgroup:")
    for j, nm in enumerate(gn):
        v = [r[j + 1] - r[j] for r in rows]
        print(f"   {nm:32s} median {statistics.median(v):8.0f}  max {max(v):8.0f} cyc")
    v = [rows[f + 1][0] - rows[f][0] for f in range(14)]
    print(f"   {'frame total':32s} median {statistics.median(v):8.0f} cyc")


if __name__ == "__main__":
    main()
