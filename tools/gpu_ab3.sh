#!/bin/bash
# tools/gpu_ab3.sh spec1 spec2 ... -- A/B of the benchmark step on ONE box.  spec = name[@lib][+switch=value,...]:
#   default            the product build, no switches
#   pipe+fac_pipe=1    the product build with crf_debug_set("fac_pipe", 1)
#   p4@pipe4+fac_pipe=1   cat_amd/lib_ab/libpipe4.so with that switch
# Two passes over all specs; prints step / den pair / call of each.  EXTRA="--B 128" adds bench arguments.
OUT=$PWD/gpurun_out; mkdir -p $OUT
for pass in 1 2; do
for spec in "$@"; do
  name=${spec%%[@+]*}; rest=${spec#$name}
  lib=""; dbg=""
  case "$rest" in @*) lib=${rest#@}; lib=${lib%%+*};; esac
  case "$rest" in *+*) dbg=${rest#*+};; esac
  L=""; [ -n "$lib" ] && L=$PWD/cat_amd/lib_ab/lib$lib.so
  CRF_LIB=$L CRF_DEBUG=$dbg timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_$name.json")); k = d["roofline"]["kernels_ms"]
    print("pass $pass %-10s %.4f ms/step (median %.4f), den pair %.4f, call %.4f, grad %.3f, ctc %.3f, loss %.6f" % ("$name", d["ms_per_step"], d["event_blocks"]["median_ms_per_step"], k["den_fwd_chain"], k["call"], k["grad"], k["ctc_fwd_chain"], d["loss"]))
except Exception as e:
    print("$name: no result", e); print(open("$OUT/ab_$name.err").read()[-600:])
PY
done; done
