#!/bin/bash
# tools/gpu_ab_smallB.sh -- does a small batch (B <= 32: at most 64 of 256 CUs busy) gain from TWO CUs per recursion?  Metric graph, one box.
OUT=$PWD/gpurun_out; mkdir -p $OUT
{
for rep in 1 2; do
for B in 8 16 32; do
  for sw in "" fac_k2 "fac_k2,fac_threads=1024"; do
    CRF_DEBUG=$sw timeout 300 python bench.py --no-cpu-baseline --B $B 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']; print('B=$B [$sw]: step %.3f ms  den %.2f ms  %s' % (d['ms_per_step'], k['den_fwd_chain'], d['roofline']['kernel'].split(' ')[0]))
except Exception as e: print('B=$B [$sw]: no result', e)"
  done
done
done
} | tee $OUT/ab_smallB.txt
