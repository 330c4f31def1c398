#!/bin/bash
# tools/gpu_r4b.sh TAG -- round 4: the N > 1 code path of bench.py under torchrun (1 rank, RCCL, DDP head) + the new tests
TAG=${1:-r4b}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --ddp-head 2> $OUT/torchrun1_$TAG.err | grep '^{' > $OUT/torchrun1_$TAG.json; echo "torchrun rc=$?"
python - <<PY
import json
r = json.load(open("$OUT/torchrun1_$TAG.json"))
print({k: r.get(k) for k in ("value", "ms_per_step", "side_stream")}, r["roofline"]["kernel"][:60], r.get("ddp_head", {}).get("ms_per_step"))
PY
grep -c "ctc_crf_hip" $OUT/torchrun1_$TAG.err
timeout 900 python -m pytest tests/test_gpu_under_nccl.py "tests/test_gpu_parity.py::test_denominator_vs_reference_kernels" -q -x -s 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25
