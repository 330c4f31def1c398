"""CPU test of the N>1 path (world_size 2, gloo): the batch dimension shards over ranks with no
data-path collective, each rank normalises by its LOCAL N (ctc_crf/__init__.py:85-87), and DDP's
gradient averaging of the acoustic model then equals the single-process full-batch gradient when the
shards are equal -- the contract bench.py and CAT's trainer (cat/ctc/train.py:352, manager.py:546)
rely on.  The loss itself has no CPU implementation (neither has the reference, setup.py:15-16), so
the oracle stands in for the kernel here; the GPU tests check the kernel against the same oracle."""
import os
import socket
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import fst_io
from tests.util import make_batch, small_synth


class _OracleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, g, labels, lx, ly, lamb):
        r = oracle.ctc_crf(g, logp.detach().numpy(), labels, lx, ly, lamb=lamb, size_average=True, threads=1)
        ctx.grads = torch.tensor(r["grad"])
        return torch.tensor([r["loss"]], dtype=torch.float32)

    @staticmethod
    def backward(ctx, go):
        return ctx.grads * go, None, None, None, None, None


def _worker(rank, world, port, fst, payload, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = fst_io.read_fst(fst)
    feats, labels, lx, ly = payload
    B = feats.shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    off = np.concatenate([[0], np.cumsum(ly)])
    torch.manual_seed(0)
    model = torch.nn.parallel.DistributedDataParallel(torch.nn.Linear(feats.shape[-1], 10))
    lp = model(torch.tensor(feats[sl])).log_softmax(-1)
    loss = _OracleLoss.apply(lp, g, labels[off[rank * B]:off[(rank + 1) * B]], lx[sl], ly[sl], 0.1)
    loss.backward()
    t = loss.detach().clone()
    dist.all_reduce(t)
    if rank == 0:
        torch.save({"w": model.module.weight.grad.clone(), "b": model.module.bias.grad.clone(), "loss": t / world},
                   os.path.join(out_dir, "ddp.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_sharding_matches_full_batch(tmp_path):
    g, fst = small_synth(tmp_path, 10, 24, 5, 3)
    _, labels, lx, ly = make_batch(g, 4, 12, 10, seed=5, ragged=False)
    rng = np.random.default_rng(1)
    feats = rng.normal(size=(4, 12, 6)).astype(np.float32)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_dir = tempfile.mkdtemp()
    mp.spawn(_worker, args=(2, port, fst, (feats, labels, lx, ly), out_dir), nprocs=2, join=True)
    got = torch.load(os.path.join(out_dir, "ddp.pt"))
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 10)
    lp = lin(torch.tensor(feats)).log_softmax(-1)
    loss = _OracleLoss.apply(lp, fst_io.read_fst(fst), labels, lx, ly, 0.1)
    loss.backward()
    assert torch.allclose(got["loss"], loss.detach(), rtol=1e-5)
    assert torch.allclose(got["w"], lin.weight.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(got["b"], lin.bias.grad, rtol=1e-4, atol=1e-6)
