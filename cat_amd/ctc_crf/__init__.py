"""cat_amd.ctc_crf -- the Python surface CAT imports from ``ctc_crf`` (reference
src/ctc_crf/ctc_crf/__init__.py), backed by the MI355X-native HIP library (cat_amd/csrc).

Same names, arguments and error behaviour as the reference:
  CTC_CRF_LOSS(lamb=0.1, size_average=True)(logits, labels, lx, ly) -> FloatTensor[1]   (:97-125)
  WARP_CTC_LOSS(size_average=True)(logits, labels, input_lengths, label_lengths)         (:128-144)
  CRFContext(den_lm, gpus)                                                               (:147-171)
  _CTC_CRF, _WARP_CTC_GPU  autograd Functions                                             (:25-94)
plus the functional form named by BASELINE.json:
  ctc_crf_loss(log_probs, labels, frame_lens, label_lens, den_lm, lamb=0.1, size_average=True)

Differences that are deliberate (DESIGN.md "boundary"): no host synchronisation in forward (the
reference has three), the loss stays on the GPU, and utterances the reference treats as
invalid (L + repeats > T) contribute 0 to the numerator instead of uninitialised memory.
"""
import os
from typing import Dict, List, Tuple, Union

import torch
from torch.autograd import Function
from torch.nn import Module

from . import _C as core

__version__ = "0.1.0"


def _assert_no_grad(tensor):
    assert not tensor.requires_grad, "shouldn't require grads"


class _WARP_CTC_GPU(Function):
    """Plain CTC NLL (reference __init__.py:25-55): costs = -sum_b logp_b, grad = -gamma_ctc."""

    @staticmethod
    def forward(ctx, logits, labels, input_lengths, label_lengths, size_average=True):
        logits = logits.contiguous()
        batch_size = logits.size(0)
        s = 1.0 / batch_size if size_average else 1.0
        costs, grads, _ = core.loss_fwd_bwd(logits, labels, input_lengths, label_lengths, 0.0, s, None)
        ctx.grads = grads
        return costs

    @staticmethod
    def backward(ctx, grad_output):
        return ctx.grads * grad_output.to(ctx.grads.device), None, None, None, None, None, None


class _CTC_CRF(Function):
    """reference __init__.py:58-94:
    costs = sum_b [logZ_den(b) - (1+lamb) logp_ctc(b)], grads = gamma_den - (1+lamb) gamma_ctc,
    both / N when size_average.  One fused native call instead of gpu_ctc + gpu_den + 5 torch ops."""

    @staticmethod
    def forward(ctx, logits, labels, input_lengths, label_lengths, lamb=0.1, size_average=True):
        logits = logits.contiguous()
        batch_size = logits.size(0)
        s = 1.0 / batch_size if size_average else 1.0
        costs, grads, _ = core.loss_fwd_bwd(logits, labels, input_lengths, label_lengths, s, s * (1.0 + lamb),
                                            core.graph_for(logits.device))
        ctx.grads = grads
        return costs

    @staticmethod
    def backward(ctx, grad_output):
        return ctx.grads * grad_output.to(ctx.grads.device), None, None, None, None, None, None


class _CTC_CRF_LOGITS(Function):
    """_CTC_CRF with the log_softmax in front of it fused in (the caller's
    ``logits = torch.log_softmax(netout, -1)`` + ``criterion(logits.float(), ...)``, cat/ctc/train.py:174-186):
    takes the RAW network output in fp32 / bf16 / fp16, the loss is that of log_softmax(netout), the gradient
    is d loss / d netout (log_softmax's backward included), computed in fp32 and returned in netout's dtype."""

    @staticmethod
    def forward(ctx, netout, labels, input_lengths, label_lengths, lamb=0.1, size_average=True):
        netout = netout.contiguous()
        batch_size = netout.size(0)
        s = 1.0 / batch_size if size_average else 1.0
        costs, grads, _ = core.loss_fwd_bwd(netout, labels, input_lengths, label_lengths, s, s * (1.0 + lamb),
                                            core.graph_for(netout.device), fused=True)
        ctx.grads = grads
        ctx.out_dtype = netout.dtype
        return costs

    @staticmethod
    def backward(ctx, grad_output):
        return (ctx.grads * grad_output.to(ctx.grads.device)).to(ctx.out_dtype), None, None, None, None, None, None


class CTC_CRF_LOSS(Module):
    def __init__(self, lamb: float = 0.1, size_average: bool = True, fuse_log_softmax: bool = False):
        """
        lamb (float): weight for auxiliary CTC loss, final loss = lamb * loss_ctc + loss_crf
        size_average (bool): whether to do average over batch size dimension.
        fuse_log_softmax (bool, not in the reference): ``forward`` takes the RAW network output (fp32, bf16 or
            fp16) instead of log-probs; log_softmax and its backward run inside the loss kernels.
        """
        super(CTC_CRF_LOSS, self).__init__()
        self.ctc_crf = _CTC_CRF_LOGITS.apply if fuse_log_softmax else _CTC_CRF.apply
        self.lamb = lamb
        self.size_average = size_average
        self.fuse_log_softmax = fuse_log_softmax

    def forward(self, logits, labels, lx, ly) -> torch.FloatTensor:
        """
        logits (torch.FloatTensor): size (N, T, V), on GPU device (log-probs; no softmax is applied).
        labels (torch.IntTensor)  : size (sum(ly), ) flattened without padding, on CPU
        lx (torch.IntTensor) : size (N, ), on CPU
        ly (torch.IntTensor) : size (N, ), on CPU
        """
        assert len(labels.size()) == 1
        if self.fuse_log_softmax:
            assert logits.dtype in (torch.float, torch.bfloat16, torch.float16), f"expect float/bfloat16/float16 network output, instead: {logits.dtype}"
        else:
            assert logits.dtype == torch.float, f"expect logits to be torch.float object, instead: {logits.dtype}"
        assert labels.dtype == torch.int, f"expect labels to be torch.int object, instead: {labels.dtype}"
        assert lx.dtype == torch.int, f"expect lx to be torch.int object, instead: {lx.dtype}"
        assert ly.dtype == torch.int, f"expect ly to be torch.int object, instead: {ly.dtype}"
        _assert_no_grad(labels)
        _assert_no_grad(lx)
        _assert_no_grad(ly)
        return self.ctc_crf(logits, labels, lx, ly, self.lamb, self.size_average)


class WARP_CTC_LOSS(Module):
    """Kept for parity with the reference (which itself recommends torch.nn.CTCLoss)."""

    def __init__(self, size_average=True):
        super(WARP_CTC_LOSS, self).__init__()
        self.ctc = _WARP_CTC_GPU.apply
        self.size_average = size_average

    def forward(self, logits, labels, input_lengths, label_lengths):
        assert len(labels.size()) == 1
        _assert_no_grad(labels)
        _assert_no_grad(input_lengths)
        _assert_no_grad(label_lengths)
        return self.ctc(logits, labels, input_lengths, label_lengths, self.size_average)


class CRFContext:
    def __init__(self, den_lm: str, gpus: Union[int, List[int]]) -> None:
        """
        den_lm (str): path to the denominator LM (OpenFst vector/standard binary, as produced by
            cat/utils/tool/prep_den_lm.sh).
        gpus   (int, List[int]): which GPU(s) to load it on.
        """
        if not os.path.isfile(den_lm):
            raise RuntimeError(f"Denominator LM model location is invalid: {den_lm}.")
        if isinstance(gpus, int):
            gpus = [gpus]
        nprocs = torch.cuda.device_count()
        if not all([i >= 0 and i < nprocs for i in gpus]):
            raise RuntimeError(f"Available GPU={nprocs}, invalid GPU ids: {gpus}.")
        gpus_t = torch.IntTensor(gpus)
        core.init_env(den_lm, gpus_t)
        self._gpus = gpus_t
        # the graphs THIS context created: __del__ releases these and nothing else (a later context on the same
        # device replaces the graph; the reference's Release() would free whatever is current, den_calculate.cu:394-425)
        self._handles = {int(i): core.graph_generation(int(i)) for i in gpus}   # {device: generation of the graph}
        self.den_lm = den_lm

    def owns_graph(self, idx: int) -> bool:
        """True while the graph this context loaded on device `idx` is still the device's current graph."""
        gen = getattr(self, '_handles', {}).get(idx)
        return gen is not None and core.graph_generation(idx) == gen

    def __del__(self):
        if hasattr(self, '_handles'):
            try:
                core.release_handles(self._handles)
            except Exception:  # interpreter shutdown
                pass
            del self._handles


_CTX_CACHE: Dict[Tuple[str, int], CRFContext] = {}


def ctc_crf_loss(log_probs: torch.Tensor, labels: torch.Tensor, frame_lens: torch.Tensor,
                 label_lens: torch.Tensor, den_lm: str, lamb: float = 0.1, size_average: bool = True,
                 fuse_log_softmax: bool = False) -> torch.Tensor:
    """Functional CTC-CRF loss (BASELINE.json north_star signature).  Lazily builds and caches the
    CRFContext for (den_lm, device) exactly as AMTrainer does (cat/ctc/train.py:180-182).
    fuse_log_softmax=True: ``log_probs`` is the raw network output (fp32 / bf16 / fp16), see _CTC_CRF_LOGITS."""
    dev = log_probs.device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (os.path.abspath(den_lm), idx)
    ctx = _CTX_CACHE.get(key)
    if ctx is None or not ctx.owns_graph(idx):   # not loaded yet, or somebody replaced / released the device's graph
        ctx = None                                # (drop every reference BEFORE the replacement is created)
        for k in [k for k in _CTX_CACHE if k[1] == idx]:  # one graph per device, like the reference
            del _CTX_CACHE[k]
        _CTX_CACHE[key] = CRFContext(den_lm, idx)
    if fuse_log_softmax:
        return _CTC_CRF_LOGITS.apply(log_probs, labels.int().cpu(), frame_lens.int().cpu(), label_lens.int().cpu(),
                                     lamb, size_average)
    return _CTC_CRF.apply(log_probs.float(), labels.int().cpu(), frame_lens.int().cpu(), label_lens.int().cpu(),
                          lamb, size_average)
