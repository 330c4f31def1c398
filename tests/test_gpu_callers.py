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
