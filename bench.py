#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its config: utterances/sec of the CTC-CRF loss
forward+backward on synthetic [B=64, T=1500, V=72] batches (B per GPU: weak scaling over --gpus).

  python bench.py --gpus N --steps K --warmup W        # N > 1 without WORLD_SIZE: spawns N ranks itself
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch already resident in HBM:
``CTC_CRF_LOSS(lamb)(log_probs, labels, lx, ly)`` + ``loss.backward()`` (= reference
_CTC_CRF.forward + .backward with grad_output = 1, ctc_crf/__init__.py:60-94), through the
reference-shaped Python surface and the C ABI.  The batch dimension shards over ranks with no
data-path collective: every rank runs B utterances with its own graph replica (one
process per GPU, like the reference's mp.spawn, cat/shared/coreutils.py:493-504).

Rank 0 prints ONE JSON line: `value` (exactly K steps between barrier + synchronize on both sides, max
over ranks), `roofline` (dominant kernel = the denominator recursions' launch, HIP-event timed on its own
stream), `cpu_baseline` (oracle port on the host cores, N = 1 only), `event_blocks` (hipEvent median over
5 blocks of K steps), and for N > 1 `strong_scaling` (global batch B split over the ranks) and `ddp_head`
(the same step behind a DDP-wrapped stand-in acoustic model whose gradient size is stated; RCCL all-reduce
over xGMI) -- extra information, never `value`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--B", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--V", type=int, default=72)
    ap.add_argument("--histories", type=int, default=2048, help="synthetic den_lm: LM states")
    ap.add_argument("--fanout", type=int, default=24, help="synthetic den_lm: tokens per LM state")
    ap.add_argument("--lamb", type=float, default=0.1)
    ap.add_argument("--ragged", action="store_true", help="T_b ~ U[0.6T, T] instead of all = T")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="utterances in the CPU sample (0 = one per core)")
    ap.add_argument("--ddp-head", action="store_true", help="time the DDP stand-in model also at N = 1")
    ap.add_argument("--ddp-layers", type=int, default=12, help="stand-in acoustic model: residual MLP blocks (8.4 M parameters each)")
    ap.add_argument("--ddp-steps", type=int, default=5)
    ap.add_argument("--allow-serial", action="store_true",
                    help="print a headline line even if the loss ran its serial schedule (no stream beside the caller's: ~1.6x slower); "
                         "without it such a run prints a `not_measured` record and exits 3")
    ap.add_argument("--share-device", action="store_true",
                    help="harness check on a 1-GPU box: N ranks on cuda:0 over gloo; every world_size > 1 leg runs, `value` is null (not a measurement)")
    return ap.parse_args()


def algorithmic_bytes(lx, ly, S, A, V):
    """SURVEY.md 8(d) official figures, and the share of the denominator-recursion launch."""
    lx = np.asarray(lx, dtype=np.int64)
    sp = 2 * np.asarray(ly, dtype=np.int64) + 1
    graph = 2 * (12 * A + 8 * S) + 8 * S
    bytes_den = int((lx * (12 * V + 8 * S)).sum() + graph)
    bytes_num = int((lx * (8 * sp + 8 * np.minimum(sp, V))).sum())
    # the launch that runs both denominator recursions: both emission reads, one state vector per direction and
    # frame (the survey's "alpha write + alpha read"), both arc tables; the grad row write (4V) belongs to the grad pass
    bytes_den_pair = int((lx * (8 * V + 8 * S)).sum() + graph)
    return bytes_den, bytes_num, bytes_den_pair


def workload_key(args, dims):
    return f"B{args.B}_T{args.T}_V{args.V}_S{dims['S']}_A{dims['A']}_{'ragged' if args.ragged else 'full'}"


def self_spawn(args):
    """--gpus N > 1 without a launcher: one process per GPU under torch.distributed.run (RCCL rendezvous on 127.0.0.1)."""
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not args.share_device:
        print(json.dumps({"metric": "utterances/sec CTC-CRF fwd+bwd", "value": None, "unit": "utterances/s", "n_gpus": args.gpus,
                          "not_measured": f"--gpus {args.gpus} but this node exposes {ndev} GPU(s)"}), flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class StandInAM(torch.nn.Module):
    """Stand-in acoustic model for the DDP leg: Linear(80 -> d) + L residual MLP blocks (d -> 4d -> d) + Linear(d -> V).
    d = 1024, L = 12: 101 M parameters = 403 MB of fp32 gradients per step (cf. 115 M in egs/libri/exp/crf-v1/readme.md:5)."""

    def __init__(self, V, layers, d=1024, F=80):
        super().__init__()
        self.inp = torch.nn.Linear(F, d)
        self.blocks = torch.nn.ModuleList(
            torch.nn.Sequential(torch.nn.LayerNorm(d), torch.nn.Linear(d, 4 * d), torch.nn.GELU(), torch.nn.Linear(4 * d, d))
            for _ in range(layers))
        self.out = torch.nn.Linear(d, V)

    def forward(self, x):
        h = self.inp(x)
        for b in self.blocks:
            h = h + b(h)
        return self.out(h)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        sys.exit(self_spawn(args))
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    share = args.share_device and world > 1
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = "WORLD_SIZE" in os.environ
    devices = [f"{socket.gethostname()}:cuda:{dev_index}"]
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")                 # RCCL refuses two ranks on one device; gloo carries the collectives
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        ids = [None] * world
        dist.all_gather_object(ids, (socket.gethostname(), dev_index, str(torch.cuda.get_device_properties(dev).uuid)
                                     if hasattr(torch.cuda.get_device_properties(dev), "uuid") else str(dev_index)))
        assert share or len(set(ids)) == world, f"ranks share a device: {ids}"
        devices = [f"{h}:cuda:{i}" for h, i, _ in ids]

    # `ctc_crf` is imported AFTER the HIP runtime, the device and RCCL are up -- as in CAT, where the import sits inside
    # AMTrainer.__init__ (cat/ctc/train.py:118) behind set_device + init_process_group (train.py:48-55).  Nothing in the
    # package depends on environment variables being set before HIP starts.
    import ctc_crf
    from cat_amd.den_lm import synth_den_lm
    from cat_amd.synth import make_batch

    # synthetic den_lm (seed 0, identical on every rank) written as an OpenFst binary and loaded
    # through the product's own reader -- the same path CRFContext takes in CAT
    tmp = tempfile.mkdtemp(prefix=f"crfbench{rank}_")
    fst = os.path.join(tmp, "den_lm.fst")
    g = synth_den_lm(args.V, args.histories, args.fanout, seed=0, path=fst)
    ctx = ctc_crf.CRFContext(fst, dev_index)
    dims = ctc_crf._C.graph_dims(ctc_crf._C.graph_for(dev))
    B, T, V = args.B, args.T, args.V
    logits, labels, lx, ly = make_batch(g, B, T, V, seed=rank, ragged=args.ragged)
    x = torch.tensor(logits, device=dev, requires_grad=True)
    labels_t, lx_t, ly_t = torch.tensor(labels), torch.tensor(lx), torch.tensor(ly)
    crit = ctc_crf.CTC_CRF_LOSS(lamb=args.lamb)

    def step():
        x.grad = None
        loss = crit(x, labels_t, lx_t, ly_t)
        loss.backward()
        return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    dt = timed(step, args.steps, args.warmup)
    loss_val = float(step().item())
    # a timing of a call whose gradient is NaN is not a measurement (round 3 found T = 3000 points of rounds 1 - 2 that were)
    assert np.isfinite(loss_val) and bool(torch.isfinite(x.grad).all().item()), "non-finite loss or gradient at this bench point"
    side_stream, call_streams = ctc_crf._C.last_side_stream(), ctc_crf._C.last_call_streams()
    fallback_den, fallback_num = ctc_crf._C.last_fallback_counts(torch.cuda.current_stream(dev).cuda_stream)
    # A run on the SERIAL schedule (the library found no stream that runs beside the caller's) is a 1.6 x slower configuration that a
    # correctly set-up process does not have: it must not become a headline number silently (round 3 measured one without noticing).
    serial_run = call_streams < 2 and not share
    if serial_run and not args.allow_serial:
        if rank == 0:
            print(json.dumps({"metric": "utterances/sec CTC-CRF fwd+bwd", "value": None, "unit": "utterances/s", "n_gpus": world,
                              "not_measured": f"the loss ran its serial schedule (call_streams={call_streams}, side stream: {side_stream}); "
                                              "pass --allow-serial to measure it anyway",
                              "serial_ms_per_step": round(dt / args.steps * 1e3, 4)}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        sys.exit(3)
    workspace_bytes = int(ctc_crf._C._lib.crf_workspace_bytes(ctc_crf._C.graph_for(dev), B, T, V, int(max(ly))))
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # --- hipEvent timing on the compute stream, median over 5 blocks of K steps ----------
    blocks = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        blocks.append(e0.elapsed_time(e1) / args.steps)
    event_blocks = {"blocks": 5, "steps_per_block": args.steps, "median_ms_per_step": round(float(np.median(blocks)), 4),
                    "min_ms_per_step": round(min(blocks), 4), "max_ms_per_step": round(max(blocks), 4),
                    "utt_per_s_at_median": round(B / (float(np.median(blocks)) * 1e-3), 1)}

    # --- per-kernel durations, measured live with HIP events on the launch streams -------------
    ctc_crf._C.profile_enable(True)
    acc = {}
    nprof = max(3, min(10, args.steps))
    for _ in range(nprof):
        step()
        for k, v in ctc_crf._C.profile_read().items():
            if v >= 0:
                acc.setdefault(k, []).append(v)
    ctc_crf._C.profile_enable(False)
    kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}
    bytes_den, bytes_num, bytes_den_pair = algorithmic_bytes(lx, ly, dims["S"], dims["A"], V)
    dom = "den_fwd_chain"   # forward and backward recursions are ONE launch (slots den_fwd_chain and den_bwd_chain both time it)
    dom_ms = kern_ms[dom]
    achieved = bytes_den_pair / (dom_ms * 1e-3) / 1e9
    gstats = ctc_crf._C.graph_stats(ctc_crf._C.graph_for(dev))
    den_path = ctc_crf._C.den_kernels(ctc_crf._C.graph_for(dev), B, T, V)   # what a call of this shape takes (asked of the library)
    batch_path = den_path == "batch"
    kname = {"factored": "crf_fac2_pair_kernel" if gstats.get("fac_geom") == 3 else "crf_fac_pair_kernel", "resident": "crf_res_pair_kernel",
             "batch": "crf_batch_persist_kernel", "streaming": "crf_den_pair_kernel"}[den_path]
    den_symbol = ctc_crf._C.last_den_kernel()    # e.g. crf_fac_pair_kernel<true,768,21,4,4,false,false,0> (this thread's last call)
    # utterance-minor kernels: one persistent launch (round 6), or one launch per frame (forward frame j + backward frame T-j) when the grid is not co-resident
    launches = (T + 1) if batch_path and "persist" not in (den_symbol or "") else 1
    # HBM traffic from the PMC counters: measured by tools/gpu_prof.sh (separate rocprofv3 --pmc passes) and committed
    # keyed by workload; any other workload prints null instead of a number that does not belong to it
    traffic, traffic_path = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # ... AND by the template instantiation the library reports for the call that was timed: a number captured on another
            # build of the kernel does not belong to this line either
            if tj.get("workload") == workload_key(args, dims) and tj.get("den_kernel") == den_symbol:
                traffic = tj.get("kernels", {}).get(kname)
                traffic_path = tj.get("whole_path")
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": "%s (denominator forward + backward recursions of all utterances, %s)" % (
            den_symbol or kname, "one launch" if launches == 1 else f"{launches} launches per call, one per frame; bytes and time below are per call"),
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "kernel_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": bytes_den_pair,
        "kernels_ms": {k: round(v, 4) for k, v in kern_ms.items()},
        # the BASELINE target is stated on the whole denominator forward-backward: official
        # bytes_den (+ bytes_num) over the span first launch .. end of the grad pass
        "den_fwd_bwd": {"bytes": bytes_den, "bytes_num": bytes_num, "ms": round(kern_ms.get("call", 0.0), 4),
                        "frac_of_peak": round((bytes_den + bytes_num) / (kern_ms.get("call", 1e9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        # every kernel of the call, PMC bytes over algorithmic bytes (Q and BP rows are both
                        # materialised and read back; the numerator rows are fp64)
                        "traffic_whole_path": traffic_path},
    }

    # secondary ceilings: arc evaluations and the 2T-step dependency chain bind before HBM does
    arc_evals = 2 * int(np.asarray(lx, dtype=np.int64).sum()) * dims["A"]
    roofline["ceilings"] = {
        "arc_evals_per_step": arc_evals,
        "arc_evals_per_s": round(arc_evals / (dom_ms * 1e-3)),
        "dependent_frames": 2 * int(max(lx)),
        "us_per_frame_den": round(dom_ms * 1e3 / max(1, int(max(lx))), 3),
        "arc_stream_bytes_reference_style": 2 * B * int(max(lx)) * dims["A"] * 12,
    }

    # attainable HBM rate on this device ("confirm on the box"): a 1 GiB device-to-device copy, read + write
    try:
        buf = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(buf)
        dst.copy_(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            dst.copy_(buf)
        e1.record()
        torch.cuda.synchronize()
        roofline["ceilings"]["hbm_copy_GBs_measured"] = round(4 * 2 * buf.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del buf, dst
    except Exception:  # informational only
        pass

    # --- strong scaling (N > 1): the SAME global batch of B utterances split over the ranks -------------
    strong = None
    if world > 1 and B % world == 0:
        Bs = B // world
        lg2, lab2, lx2, ly2 = make_batch(g, Bs, T, V, seed=100 + rank, ragged=args.ragged)
        x2 = torch.tensor(lg2, device=dev, requires_grad=True)
        l2, lx2t, ly2t = torch.tensor(lab2), torch.tensor(lx2), torch.tensor(ly2)

        def step2():
            x2.grad = None
            crit(x2, l2, lx2t, ly2t).backward()

        dts = timed(step2, args.steps, args.warmup)
        strong = {"global_batch": B, "per_gpu": Bs, "value": round(B * args.steps / dts, 2), "ms_per_step": round(dts / args.steps * 1e3, 4)}

    # --- the same step behind a DDP-wrapped stand-in acoustic model (N > 1, or --ddp-head) -------------
    ddp_info = None
    if world > 1 or args.ddp_head:
        torch.manual_seed(0)
        model = StandInAM(V, args.ddp_layers).to(dev)
        nparam = sum(p.numel() for p in model.parameters())
        feats = torch.randn(B, T, 80, device=dev)
        if use_dist:
            from torch.nn.parallel import DistributedDataParallel as DDP
            model = DDP(model, device_ids=[dev_index], bucket_cap_mb=100, gradient_as_bucket_view=True)

        def ddp_step():
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):   # bf16 encoder + fp32 loss (BASELINE config #4)
                out = model(feats)
            lp = torch.log_softmax(out.float(), dim=-1)
            crit(lp, labels_t, lx_t, ly_t).backward()

        dt2 = timed(ddp_step, args.ddp_steps, 2)
        assert all(bool(torch.isfinite(p.grad).all().item()) for p in model.parameters() if p.grad is not None), "non-finite gradient behind the DDP head"
        ddp_info = {"value": round(world * B * args.ddp_steps / dt2, 2), "ms_per_step": round(dt2 / args.ddp_steps * 1e3, 4),
                    "steps": args.ddp_steps, "parameters": nparam, "den_kernel": ctc_crf._C.last_den_kernel(), "call_streams": ctc_crf._C.last_call_streams(), "gradient_bytes_per_step": 4 * nparam,
                    "model": f"Linear(80->1024) + {args.ddp_layers} x residual MLP(1024->4096->1024) + Linear(1024->{V}), bf16 autocast, "
                             "log_softmax + CTC_CRF_LOSS in fp32" + (", torch DDP (bucket 100 MB), gradients all-reduced by RCCL" if use_dist else ", no DDP (single process)")}
        del model, feats

    # --- CPU baseline: the oracle port (fp32, OpenMP over utterances) on this box's cores ------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        from oracle import fst_io
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        n = args.cpu_sample or min(B, cores)
        gg = fst_io.read_fst(fst)
        off = np.concatenate([[0], np.cumsum(ly)])
        t0 = time.perf_counter()
        oracle.ctc_crf(gg, logits[:n], labels[:off[n]], lx[:n], ly[:n], lamb=args.lamb, precision="f32", threads=cores)
        cdt = time.perf_counter() - t0
        cpu = {"value": round(n / cdt, 4), "unit": "utterances/s", "cores": int(min(cores, n)), "kind": "port",
               "sample": f"{n} of the {B} utterances of the same batch (T={T}, V={V}, same den_lm), "
                         f"oracle/crf_oracle.c fp32, one OpenMP thread per utterance, {cdt:.1f} s"}

    if rank == 0 and share:
        rec = {"metric": "utterances/sec CTC-CRF fwd+bwd", "value": None, "unit": "utterances/s", "n_gpus": world,
               "not_measured": f"--share-device: {world} ranks on ONE GPU over gloo -- a check of the harness' world_size > 1 legs, not a measurement",
               "shared_device_run": {"world_size": world, "devices": devices, "ms_per_step": round(ms_per_step, 4), "loss": round(loss_val, 6),
                                     "den_kernel": den_symbol, "strong_scaling": strong, "ddp_head": ddp_info}}
        print(json.dumps(rec), flush=True)
    elif rank == 0:
        out = {
            "metric": "utterances/sec CTC-CRF fwd+bwd", "value": round(value, 2), "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"B={B} per GPU, T={T}, V={V}-phone synthetic den_lm (T.fst o n-gram, "
                                   f"H={args.histories}, d={args.fanout}, seed 0): S={dims['S']} states, "
                                   f"A={dims['A']} arcs, P={dims['P']} (dst,label) pairs; "
                                   f"{'ragged lx' if args.ragged else 'lx = T'}, ly = lx//6, lamb={args.lamb}",
                       "den_kernels": ("factored register-resident, 2 CUs per recursion and utterance (products exchanged through L2 every "
                                       "frame), forward + backward one launch" if den_path == "factored" and gstats.get("fac_geom") == 3 else
                                       "factored register-resident, 1 CU per recursion and utterance, forward + backward one launch; the grad "
                                       "pass follows them in stages released by stream-level waits" if den_path == "factored" else
                                       f"register-resident, K={gstats['res_K']} CUs per recursion" if den_path == "resident" else
                                       "utterance-minor (state vectors [state][utterance] in global memory, arcs read once per frame for the "
                                       "whole batch, one launch per frame)" if batch_path else "streaming"),
                       "global_batch": world * B, "parallelism": f"dp{world} (batch sharded, no data-path collective)",
                       "world_size": world, "devices": devices},
            "loss": round(loss_val, 6), "grad_finite": True,
            "schedule": {"call_streams": call_streams, "side_stream": side_stream, "rccl_initialised": bool(use_dist and not share),
                         "serial": bool(serial_run)},
            "fallback_utterances": {"denominator": fallback_den, "numerator": fallback_num},
            "workspace_bytes": workspace_bytes,
            "event_blocks": event_blocks,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if strong:
            out["strong_scaling"] = strong
        if ddp_info:
            out["ddp_head"] = ddp_info
        print(json.dumps(out), flush=True)
    del ctx
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
