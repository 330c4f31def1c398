"""tools/probe_two_streams.py -- large-graph path (utterance-minor kernels, one launch per frame): do two callers with
HALF the batch each, on two streams, finish sooner than one caller with the whole batch?  (Each launch has ~14 us of fixed
cost -- dispatch, cold L2, tail -- around ~7 us of streaming: two interleaved launch chains could hide one's fixed part
behind the other's streaming.)   usage: python tools/probe_two_streams.py [H d B T]"""
import os, sys, time, tempfile, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_crf
from cat_amd.den_lm import synth_den_lm
from cat_amd.synth import make_batch

H, d, B, T = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (8192, 32, 64, 1500)
V = 72
core = ctc_crf._C
fst = os.path.join(tempfile.mkdtemp(), "den.fst")
g = synth_den_lm(V, H, d, seed=0, path=fst)
ctx = ctc_crf.CRFContext(fst, 0)
gh = core.graph_for(torch.device("cuda", 0))
print(f"S={g['S']} A={g['A']} kernels: {core.den_kernels(gh, B, T, V)}")


def mk(Bn, seed):
    lg, lab, lx, ly = make_batch(g, Bn, T, V, seed=seed, ragged=False)
    return torch.tensor(lg, device="cuda:0"), torch.tensor(lab), torch.tensor(lx), torch.tensor(ly)


whole, halves = mk(B, 0), [mk(B // 2, 1), mk(B // 2, 2)]


def call(x, lab, lx, ly):
    return core.loss_fwd_bwd(x, lab, lx, ly, 1.0 / B, 1.1 / B, gh, True)


for _ in range(2):
    call(*whole)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    call(*whole)
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / n
print(f"one caller, B={B}: {t1 * 1e3:.2f} ms/step, {B / t1:.0f} utt/s")
for _ in range(2):
    call(*halves[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    call(*halves[0])
torch.cuda.synchronize()
th = (time.perf_counter() - t0) / n
print(f"one caller, B={B // 2}: {th * 1e3:.2f} ms/step, {B / 2 / th:.0f} utt/s")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(i, reps):
    with torch.cuda.stream(streams[i]):
        for _ in range(reps):
            call(*halves[i])


for i in range(2):
    run(i, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(i, n)) for i in range(2)]
for t in ths:
    t.start()
for t in ths:
    t.join()
torch.cuda.synchronize()
t2 = (time.perf_counter() - t0) / n
print(f"two callers, B={B // 2} each on its own stream: {t2 * 1e3:.2f} ms per pair of steps, {B / t2:.0f} utt/s")
