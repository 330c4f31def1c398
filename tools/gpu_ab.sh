#!/bin/bash
# tools/gpu_ab.sh -- quick A/B of bench variants selected by environment variables (one gpurun call).
# usage: bash tools/gpu_ab.sh tag "ENV1=.. ENV2=.." "ENVa=.." ...   (each quoted arg = one variant; "" = default)
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
i=0
for V in "$@"; do
  echo "== variant $i: [$V]"
  env $V timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/ab_${TAG}_$i.json 2> $OUT/ab_${TAG}_$i.err || tail -3 $OUT/ab_${TAG}_$i.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_${TAG}_$i.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["kernels_ms"])
except Exception as e:
    print("no result", e)
PY
  i=$((i+1))
done
