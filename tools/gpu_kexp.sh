#!/bin/bash
# K-split experiment: same graph with K forced to 1, 2, 4 (isolates exchange cost from compute)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for cfg in "1024 24 1" "1024 24 2" "1024 24 4" "512 24 1" "512 24 2" "512 24 4" "2048 24 2" "2048 24 4"; do
  set -- $cfg
  CRF_DEBUG=res_mink=$3 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --histories $1 --fanout $2 > $OUT/kexp_$1_$2_$3.json 2> $OUT/kexp.err || tail -2 $OUT/kexp.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/kexp_$1_$2_$3.json")); k=d["roofline"]["kernels_ms"]
    print("H=$1 d=$2 minK=$3 ms/step", d["ms_per_step"], "fwd", k["den_fwd_chain"], "bwd", k["den_bwd_chain"], "us/frame fwd %.2f"%(k["den_fwd_chain"]/1.5))
except Exception as e: print("H=$1 d=$2 K=$3 failed", e)
PY
done
