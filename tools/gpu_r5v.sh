#!/bin/bash
# tools/gpu_r5v.sh -- round 5: staged schedule with the numerator's grad half BEHIND the den half (gctc_last=1) vs in front (default)
OUT=$PWD/gpurun_out; mkdir -p $OUT
CRF_DEBUG=gctc_last=1 timeout 600 python -m pytest tests/test_gpu_metric_shape.py -m gpu -q -x -k "poisoned" 2>&1 | tail -2
for B in 64 80 88 96; do
  EXTRA="--B $B --steps 10" bash tools/gpu_ab3.sh default last+gctc_last=1 2>&1 | grep "pass 2" | sed "s/^/B=$B /"
done | tee $OUT/r5v_ab.txt
