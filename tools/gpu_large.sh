#!/bin/bash
# tools/gpu_large.sh TAG -- the large-graph points: S = 16 385 (H=8192, d=32) at B=64/T=1500, and BASELINE config #5
# (V=5000, H=32768, d=64 => S=65 537, A=4.3 M; T=3000; B=8 per GPU = the 8-GPU share of B=64), one JSON line each.
TAG=${1:-r2}
OUT=$PWD/gpurun_out; mkdir -p $OUT
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > $OUT/pt_${TAG}_$name.json 2> $OUT/pt_${TAG}_$name.err || tail -5 $OUT/pt_${TAG}_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/pt_${TAG}_$name.json"))
    k = d["roofline"]["kernels_ms"]
    print("$name: %.1f utt/s, %.3f ms/step, den %.2f ms, roofline frac %.4f, %s" % (d["value"], d["ms_per_step"], k.get("den_fwd_chain", -1), d["roofline"]["frac"], d["config"]["workload"].split(":")[1][:60]))
except Exception as e:
    print("$name: no result", e)
PY
}
run large --histories 8192 --fanout 32 --steps 3 --warmup 1
CRF_NO_BATCH=1 run large_streaming --histories 8192 --fanout 32 --steps 2 --warmup 1
run c5 --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1
