#!/bin/bash
# tools/gpu_ablib.sh name1 name2 ... -- the benchmark with alternative builds of the library (cat_amd/lib_ab/lib<name>.so, built with
# CRF_BUILD_OUT / CRF_BUILD_DEFS; "default" = the product build), two passes each, kernel times of the second
OUT=$PWD/gpurun_out; mkdir -p $OUT
for pass in 1 2; do
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L=$PWD/cat_amd/lib_ab/lib$v.so; fi
  CRF_LIB=$L timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $OUT/ab_$v.json 2> $OUT/ab_$v.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_$v.json")); k = d["roofline"]["kernels_ms"]
    print("pass $pass $v: %.4f ms/step (median %.4f), den pair %.4f, call %.4f, grad %.3f" % (d["ms_per_step"], d["event_blocks"]["median_ms_per_step"], k["den_fwd_chain"], k["call"], k["grad"]))
except Exception as e:
    print("$v: no result", e); print(open("$OUT/ab_$v.err").read()[-400:])
PY
done; done
