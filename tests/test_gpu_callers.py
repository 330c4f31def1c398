"""The in-tree CALLERS of the package, call for call, with backward through the encoder:

* ``AMTrainer.forward``        cat/ctc/train.py:172-190   encoder -> log_softmax -> .cpu() metadata -> lazy CRFContext ->
                                                          autocast(enabled=False) -> CRFLoss(logits.float(), labels.int(), ...)
* CUSIDE ``forward``           cat/ctc/train_unified.py:241-270   the same criterion called TWICE per step: on the full-context
                                                          logits and on the chunked encoder output, concatenated back to
                                                          [N, T_chunk * num_chunks, V] and cut to [:, :lx[0], :] (a non-contiguous view)

CAT itself cannot be imported here (nine of its dependencies are absent) and /root/reference does not
exist on the GPU box, so a stand-in performs exactly those calls.  The reference result comes from the fp64 oracle wrapped
as an autograd Function, driven by the same encoder weights on the CPU."""
import os

import numpy as np
import pytest
import torch
from torch import nn
from torch.amp import autocast

import oracle
from oracle import fst_io
from tests.util import make_batch, small_synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctc_crf
    return ctc_crf


class _OracleCRF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, g, labels, lx, ly, lamb):
        r = oracle.ctc_crf(g, logp.detach().numpy(), labels.numpy(), lx.numpy(), ly.numpy(), lamb=lamb, size_average=True)
        ctx.grads = torch.tensor(r["grad"])
        return torch.tensor([r["loss"]], dtype=torch.float32)

    @staticmethod
    def backward(ctx, go):
        return ctx.grads * go, None, None, None, None, None


class _Encoder(nn.Module):
    """Stand-in for model_zoo.AbsEncoder: (feats [N,T,F], lx) -> (logits [N,T,V], lx)."""

    def __init__(self, F, H, V):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(F, H), nn.Tanh(), nn.Linear(H, V))

    def forward(self, feats, lx):
        return self.net(feats), lx


class _AMTrainer(nn.Module):
    """The calls of cat/ctc/train.py:100-190 (use_crf=True), nothing else."""

    def __init__(self, crf, encoder, den_lm, lamb=0.01):
        super().__init__()
        self.encoder = encoder
        self.den_lm = den_lm
        assert den_lm is not None and os.path.isfile(den_lm)
        self.criterion = crf.CTC_CRF_LOSS(lamb=lamb)          # `from ctc_crf import CTC_CRF_LOSS as CRFLoss`  (:118-120)
        self._crf = crf
        self._crf_ctx = None

    def register_crf_ctx(self, den_lm=None):                  # :132-141
        self._crf_ctx = self._crf.CRFContext(den_lm, next(iter(self.encoder.parameters())).device.index)

    def forward(self, feats, lx, labels, ly):                 # :172-190
        logits, lx = self.encoder(feats, lx)
        logits = torch.log_softmax(logits, dim=-1)
        labels = labels.cpu()
        lx = lx.cpu()
        ly = ly.cpu()
        if self._crf_ctx is None:
            self.register_crf_ctx(self.den_lm)                # lazy init
        with autocast("cuda", enabled=False):
            loss = self.criterion(logits.float(), labels.to(torch.int), lx.to(torch.int), ly.to(torch.int))
        return loss


class _CUSIDETrainer(_AMTrainer):
    """cat/ctc/train_unified.py:235-270: full-context loss + chunk loss on the re-assembled chunk output."""

    def __init__(self, *a, chunk=8, **k):
        super().__init__(*a, **k)
        self.chunk = chunk

    def chunk_forward(self, feats, lx):
        N, T, F = feats.shape
        nc = (T + self.chunk - 1) // self.chunk
        pad = nc * self.chunk - T
        x = torch.nn.functional.pad(feats, (0, 0, 0, pad))
        x = x.view(N * nc, self.chunk, F)                     # chunks become batch entries
        enc_out, _ = self.encoder(x, None)
        return enc_out.contiguous().view(N, self.chunk * nc, -1)   # (:227-229)

    def forward(self, feats, lx, labels, ly):
        logits, lx = self.encoder(feats, lx)
        logits = torch.log_softmax(logits, dim=-1)
        labels = labels.cpu(); lx = lx.cpu(); ly = ly.cpu()
        if self._crf_ctx is None:
            self.register_crf_ctx(self.den_lm)
        with autocast("cuda", enabled=False):
            loss = self.criterion(logits.float(), labels.to(torch.int), lx.to(torch.int), ly.to(torch.int))
        chunk_enc_out = self.chunk_forward(feats, lx)
        chunk_enc_out = chunk_enc_out[:, : lx[0], :]          # non-contiguous whenever the padded length exceeds lx[0]
        chunk_logits = torch.log_softmax(chunk_enc_out, dim=-1)
        with autocast("cuda", enabled=False):
            chunk_loss = self.criterion(chunk_logits.float(), labels.to(torch.int), lx.to(torch.int), ly.to(torch.int))
        return loss + chunk_loss


def _reference(enc_cpu, g, feats, labels, lx, ly, lamb, chunk=None):
    logits = torch.log_softmax(enc_cpu.net(feats), dim=-1)
    loss = _OracleCRF.apply(logits, g, labels, lx, ly, lamb)
    if chunk:
        N, T, F = feats.shape
        nc = (T + chunk - 1) // chunk
        x = torch.nn.functional.pad(feats, (0, 0, 0, nc * chunk - T)).view(N * nc, chunk, F)
        co = enc_cpu.net(x).view(N, chunk * nc, -1)[:, : int(lx[0]), :]
        loss = loss + _OracleCRF.apply(torch.log_softmax(co, dim=-1).contiguous(), g, labels, lx, ly, lamb)
    loss.backward()
    return float(loss.item()), [p.grad.clone() for p in enc_cpu.parameters()]


def _setup(tmp_path, N=4, T=45, F=10, H=16, V=12):
    g, fst = small_synth(tmp_path, V, 40, 6, 7)
    _, labels, lx, ly = make_batch(g, N, T, V, seed=2, ragged=True)
    lx = lx.copy(); lx[0] = T - 3                             # the longest utterance is shorter than the padded batch
    lx = np.sort(lx)[::-1].copy()                             # CAT sorts by length (lx[0] is the maximum)
    ly = np.maximum(lx // 6, 1).astype(np.int32)
    from cat_amd.den_lm import random_labels_from_graph
    rng = np.random.default_rng(3)
    labels = np.concatenate([random_labels_from_graph(g, int(n), rng) for n in ly]).astype(np.int32)
    torch.manual_seed(0)
    enc = _Encoder(F, H, V)
    feats = torch.randn(N, T, F)
    return g, fst, enc, feats, torch.tensor(labels).long(), torch.tensor(lx).long(), torch.tensor(ly).long()


@pytest.mark.parametrize("kind", ["amtrainer", "cuside"])
def test_trainer_call_sequence_vs_oracle(crf, tmp_path, kind):
    import copy
    g, fst, enc, feats, labels, lx, ly = _setup(tmp_path)
    lamb = 0.01
    ref_loss, ref_grads = _reference(copy.deepcopy(enc), fst_io.read_fst(fst), feats, labels.int(), lx.int(), ly.int(), lamb,
                                     chunk=8 if kind == "cuside" else None)
    enc_gpu = copy.deepcopy(enc).cuda()
    if kind == "amtrainer":
        model = _AMTrainer(crf, enc_gpu, fst, lamb=lamb)
    else:
        model = _CUSIDETrainer(crf, enc_gpu, fst, lamb=lamb, chunk=8)
    # imported and constructed AFTER the HIP runtime, the allocator and other streams are up -- as in CAT, where
    # `ctc_crf` is imported inside AMTrainer.__init__ (train.py:118) behind set_device + NCCL init (train.py:48-55)
    assert torch.cuda.is_initialized()
    loss = model(feats.cuda(), lx.cuda(), labels, ly)         # full-context logits are [N, T, V] with T > lx[0]
    loss.backward()
    assert loss.shape == (1,)
    assert abs(loss.item() - ref_loss) <= 2e-4 * abs(ref_loss)
    for pg, rg in zip(enc_gpu.parameters(), ref_grads):
        got = pg.grad.cpu()
        assert torch.isfinite(got).all()
        err = (got - rg).norm() / rg.norm().clamp_min(1e-30)
        assert err <= 1e-3, err                               # encoder matmuls: GPU fp32 vs CPU fp32
    # the same step under bf16 autocast (train.py runs the encoder under autocast; the loss leaves it, :184-186)
    model.zero_grad(set_to_none=True)
    with autocast("cuda", dtype=torch.bfloat16):
        loss16 = model(feats.cuda(), lx.cuda(), labels, ly)
    loss16.backward()
    assert abs(loss16.item() - ref_loss) <= 3e-2 * abs(ref_loss)
    assert all(torch.isfinite(pg.grad).all() for pg in enc_gpu.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [False, True])
def test_concurrent_caller_streams(crf, tmp_path, threads):
    """Two callers on one device that are NOT ordered on one stream -- two torch streams driven from one host thread, and two
    host threads with a stream each -- each issuing a run of loss calls with its own inputs.  Every call owns a context per
    (device, caller stream): counters, events and the side stream of one caller never meet the other's (round 1 had ONE
    context per device: a second caller's prep kernel could clear the stage counters the first caller's grad launches were
    waiting on).  Every result must equal the one the same inputs give alone (up to the summation order of the numerator's
    LDS float atomics: 1e-6 of the largest gradient entry)."""
    import threading
    g, fst = small_synth(tmp_path, 72, 256, 16, 9)
    core = crf._C
    ctx = crf.CRFContext(fst, 0)
    gh = core.graph_for(torch.device("cuda", 0))
    assert core.graph_stats(gh)["fac"] == 1                       # the staged schedule (stream-level waits on per-context counters)
    B, T, V, lamb = 8, 300, 72, 0.1
    data = [make_batch(g, B, T, V, seed=s, ragged=True) for s in (21, 22)]
    dev = [(torch.tensor(lg, device="cuda:0"), torch.tensor(lab), torch.tensor(lx), torch.tensor(ly)) for lg, lab, lx, ly in data]

    def call(i):
        x, lab, lx, ly = dev[i]
        loss, grad, _ = core.loss_fwd_bwd(x, lab, lx, ly, 1.0 / B, (1 + lamb) / B, gh, True)
        return loss, grad

    alone = [call(0), call(1)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]

    def run(i, reps):
        with torch.cuda.stream(streams[i]):
            for _ in range(reps):
                outs[i].append(call(i))

    if threads:
        th = [threading.Thread(target=run, args=(i, 12)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    else:
        for _ in range(12):                                         # interleaved enqueueing from one thread
            run(0, 1)
            run(1, 1)
    torch.cuda.synchronize()
    for i in range(2):
        l0, g0 = alone[i]
        assert len(outs[i]) == 12
        for loss, grad in outs[i]:
            assert torch.isfinite(loss).all()
            assert abs(loss.item() - l0.item()) <= 1e-6 * abs(l0.item()), (i, loss.item(), l0.item())
            assert (grad - g0).abs().max().item() <= 1e-6 * g0.abs().max().item()
    del ctx
