"""The PRODUCT in two processes at once (the reference runs one process per GPU: cat/shared/coreutils.py:493-504, DDP wrap
cat/ctc/train.py:352, process group cat/ctc/train.py:45-55).  A 1-GPU box has one device, so both ranks sit on cuda:0 and
talk gloo -- what is exercised is everything a rank owns privately: its HIP context, its graph replica, its side stream,
its fine-grained flag words and stream-level waits, its pinned staging ring -- while ANOTHER process drives the same
kernels on the same device at the same time.

  (a) two ranks, each a DDP-wrapped stand-in encoder + the real CTC_CRF_LOSS on its half of the batch; the all-reduced
      (averaged) parameter gradients equal the single-process full-batch gradients computed with the fp64 oracle behind the
      same encoder weights, and both ranks' losses average to the full-batch loss;
  (b) the same on a graph that takes the two-CUs-per-recursion layout forced (co-resident workgroups that spin on their
      peers: the second process's grid competes for the CUs);
  (c) bench.py --gpus 2 --share-device: the harness' world_size > 1 legs (self-spawn, all_gather_object, strong scaling,
      DDP head) executed once; the record says "not measured" instead of a value.
"""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

import oracle
from oracle import fst_io
from tests.conftest import ROOT
from tests.util import make_batch, small_synth

pytestmark = pytest.mark.gpu


class _OracleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, g, labels, lx, ly, lamb):
        r = oracle.ctc_crf(g, logp.detach().numpy(), labels, lx, ly, lamb=lamb, size_average=True, threads=2)
        ctx.grads = torch.tensor(r["grad"])
        return torch.tensor([r["loss"]], dtype=torch.float32)

    @staticmethod
    def backward(ctx, go):
        return ctx.grads * go, None, None, None, None, None


def _encoder(F, V):
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(F, 32), torch.nn.Tanh(), torch.nn.Linear(32, V))


def _worker(rank, world, port, fst, payload, out_dir, switches, steps):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)                                   # BOTH ranks on the one device of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import ctc_crf                                             # after the process group, as in CAT (train.py:118)
    for k, v in switches.items():
        ctc_crf._C.debug_set(k, v)
    ctx = ctc_crf.CRFContext(fst, 0)
    feats, labels, lx, ly = payload
    B = feats.shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    off = np.concatenate([[0], np.cumsum(ly)])
    model = torch.nn.parallel.DistributedDataParallel(_encoder(feats.shape[-1], 10).cuda(), device_ids=[0])
    crit = ctc_crf.CTC_CRF_LOSS(lamb=0.1)
    x = torch.tensor(feats[sl], device="cuda:0")
    lab = torch.tensor(labels[off[rank * B]:off[(rank + 1) * B]])
    dist.barrier()
    for _ in range(steps):                                     # several steps back to back: the two processes' calls interleave
        model.zero_grad(set_to_none=True)
        loss = crit(model(x).log_softmax(-1), lab, torch.tensor(lx[sl]), torch.tensor(ly[sl]))
        loss.backward()
    torch.cuda.synchronize()
    t = loss.detach().cpu().clone()
    dist.all_reduce(t)
    if rank == 0:
        torch.save({"grads": [p.grad.cpu().clone() for p in model.module.parameters()], "loss": t / world,
                    "kernel": ctc_crf._C.last_den_kernel()}, os.path.join(out_dir, "ddp.pt"))
    dist.barrier()
    del ctx
    dist.destroy_process_group()


@pytest.mark.parametrize("switches", [{}, {"fac_k2": 1}])
def test_two_processes_share_a_device(tmp_path, switches):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    g, fst = small_synth(tmp_path, 10, 24, 5, 3)
    B, T, F = 8, 40, 6
    _, labels, lx, ly = make_batch(g, B, T, 10, seed=5, ragged=True)
    feats = np.random.default_rng(1).normal(size=(B, T, F)).astype(np.float32)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_dir = tempfile.mkdtemp()
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, port, fst, (feats, labels, lx, ly), out_dir, switches, 5), nprocs=2, join=True)
    got = torch.load(os.path.join(out_dir, "ddp.pt"))
    if switches:
        assert got["kernel"].startswith("crf_fac2_pair_kernel"), got["kernel"]
    # single process, full batch, fp64 oracle behind the same encoder: each rank normalises by its LOCAL batch
    # (ctc_crf/__init__.py:85-87), DDP averages the ranks -- equal shards, so that is the full-batch mean
    enc = _encoder(F, 10)
    off = np.concatenate([[0], np.cumsum(ly)])
    gref = fst_io.read_fst(fst)
    total = 0.0
    for r in range(2):
        sl = slice(r * (B // 2), (r + 1) * (B // 2))
        lp = enc(torch.tensor(feats[sl])).log_softmax(-1)
        loss = _OracleLoss.apply(lp, gref, labels[off[sl.start]:off[sl.stop]], lx[sl], ly[sl], 0.1)
        (loss / 2).backward()
        total += float(loss.item()) / 2
    assert abs(float(got["loss"]) - total) <= 1e-4 * abs(total)
    for a, p in zip(got["grads"], enc.parameters()):
        assert torch.allclose(a, p.grad, rtol=2e-4, atol=2e-6), (a - p.grad).abs().max()


def test_bench_world_size_two_on_one_device():
    """bench.py --gpus 2 --share-device: both ranks on cuda:0 over gloo; every world_size > 1 leg runs, nothing is reported as a
    measurement (two ranks on one GPU are not the scaling the metric asks for)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "3", "--warmup", "1",
                          "--B", "8", "--T", "200", "--ddp-layers", "1", "--ddp-steps", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["value"] is None and "not_measured" in rec and rec["n_gpus"] == 2
    sh = rec["shared_device_run"]
    assert sh["world_size"] == 2 and sh["ms_per_step"] > 0 and sh["strong_scaling"]["per_gpu"] == 4
    assert sh["ddp_head"]["parameters"] > 0 and np.isfinite(sh["loss"])
