#!/bin/bash
# tools/gpu_sweep.sh -- graph-size sweep of the bench (fits frame time = c0 + c1 * arcs)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for cfg in "72 2" "512 24" "1024 24" "2048 8" "2048 24" "2048 48"; do
  set -- $cfg
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --histories $1 --fanout $2 > $OUT/sweep_$1_$2.json 2> $OUT/sweep_$1_$2.err || tail -2 $OUT/sweep_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/sweep_$1_$2.json")); k=d["roofline"]["kernels_ms"]
    print("H=$1 d=$2", d["config"]["workload"].split(":")[1].split(";")[0], "ms/step", d["ms_per_step"], "fwd", k["den_fwd_chain"], "bwd", k["den_bwd_chain"], "ctc", k["ctc_fwd_chain"], "grad", k["grad"])
except Exception as e: print("H=$1 d=$2 failed", e)
PY
done
