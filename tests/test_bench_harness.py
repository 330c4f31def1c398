"""bench.py's launcher contract: `--gpus N` without a launcher spawns N ranks itself (one process per GPU, as the
reference does with mp.spawn, cat/shared/coreutils.py:493-504); a node with fewer GPUs gets a "not measured" line, never
an extrapolation; under torch.distributed.run the rank initialises RCCL ("nccl"), checks world size and device
distinctness, and the DDP stand-in leg runs (1-rank NCCL on the single GPU of the test box)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_not_measured_when_gpus_missing():
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stderr[-500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["value"] is None and "not_measured" in d and d["n_gpus"] == have + 2


@pytest.mark.gpu
def test_one_rank_nccl_with_ddp_stand_in():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--B", "8", "--T", "200", "--histories", "256", "--fanout", "8", "--no-cpu-baseline", "--ddp-head", "--ddp-layers", "1",
           "--ddp-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["world_size"] == 1 and len(d["config"]["devices"]) == 1
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["event_blocks"]["median_ms_per_step"] > 0
    assert d["ddp_head"]["value"] > 0 and d["ddp_head"]["gradient_bytes_per_step"] == 4 * d["ddp_head"]["parameters"]
    assert "DDP" in d["ddp_head"]["model"]


def test_committed_pmc_traffic_belongs_to_the_default_bench_line():
    """profiles/pmc_traffic.json (tools/pmc_summary.py) is what bench.py quotes as `roofline.traffic`: it must be keyed by the
    DEFAULT bench workload and by the kernel instantiation the timed (staged) schedule runs -- a file captured on another
    workload or on the serial schedule's instantiation would make every default bench line print `traffic: null` -- and the
    dominant kernel's bytes must be its algorithmic bytes, not some other kernel's."""
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert d["workload"] == "B64_T1500_V72_S4097_A103791_full"
    assert d["den_kernel"].startswith("crf_fac_pair_kernel<true,1024,")
    alg = 2 * (64 * 1500 * (4 * 72 + 4 * 4097) + 12 * 103791 + 12 * 4097)
    assert 0.95 * alg <= d["kernels"]["crf_fac_pair_kernel"] <= 1.25 * alg
    assert 1.0 < d["whole_path"]["ratio"] < 3.0
    src = d["source"].split(":")[0]
    assert os.path.exists(os.path.join(ROOT, src)), src
