#!/bin/bash
# tools/gpu_large.sh TAG [sweep] -- the large-graph points: S = 16 385 (H=8192, d=32) at B=64/T=1500, and BASELINE config #5
# (V=5000, H=32768, d=64 => S=65 537, A=4.3 M; T=3000; B=8 per GPU = the 8-GPU share of B=64), one JSON line each.
# "sweep": also utterances per group (CRF_BAT_UL) and steps per task (CRF_BAT_TASK), and a kernel trace of one step.
TAG=${1:-r2}
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > $OUT/pt_${TAG}_$name.json 2> $OUT/pt_${TAG}_$name.err || tail -5 $OUT/pt_${TAG}_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/pt_${TAG}_$name.json"))
    k = d["roofline"]["kernels_ms"]
    print("$name: %.1f utt/s, %.3f ms/step, den %.2f ms, grad %.2f ms, roofline frac %.4f, %s" % (d["value"], d["ms_per_step"], k.get("den_fwd_chain", -1), k.get("grad", -1), d["roofline"]["frac"], d["config"]["workload"].split(":")[1][:60]))
except Exception as e:
    print("$name: no result", e)
PY
}
LARGE="--histories 8192 --fanout 32 --steps 3 --warmup 1"
run large $LARGE
if [ "$2" == "sweep" ]; then
  CRF_DEBUG=bat_ul=64 run large_ul64 $LARGE
  CRF_DEBUG=bat_ul=16 run large_ul16 $LARGE
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_large_$TAG -o trace -- python $REPO/bench.py --no-cpu-baseline --histories 8192 --fanout 32 --steps 1 --warmup 1 > $OUT/prof_large_$TAG.log 2>&1; echo "rocprof rc=$?" )
  F=$(find $OUT/prof_large_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -8 "$F" | cut -c1-220
  find $OUT/prof_large_$TAG -name "*kernel_trace.csv" -size +8M -delete
else
  CRF_DEBUG=no_batch=1 run large_streaming --histories 8192 --fanout 32 --steps 2 --warmup 1
fi
run c5 --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1
