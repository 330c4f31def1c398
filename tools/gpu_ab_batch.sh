#!/bin/bash
# tools/gpu_ab_batch.sh lib ... -- A/B of library builds (cat_amd/lib_ab/lib<name>.so; `prod` = the product library) on ONE box: the large-graph
# point (S = 16 385, B = 64, T = 1 500) as one persistent launch and as one launch per frame; den = the recursions by HIP events.
OUT=$PWD/gpurun_out; mkdir -p $OUT
for rep in 1 2; do
for n in "$@"; do
  lib=$PWD/cat_amd/lib_ab/lib$n.so; [ "$n" = prod ] && lib=$PWD/cat_amd/lib/libctc_crf_hip.so
  for ps in 1 0; do
    CRF_LIB=$lib CRF_DEBUG=bat_persist=$ps timeout 300 python bench.py --no-cpu-baseline --histories ${H:-8192} --fanout ${D:-32} --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']; print('$n persist=$ps: step %.3f ms  den %.2f ms  %s  fallback %s' % (d['ms_per_step'], k['den_fwd_chain'], d['roofline']['kernel'].split(' ')[0], d['fallback_utterances']))
except Exception as e: print('$n persist=$ps: no result', e)"
  done
done
done
