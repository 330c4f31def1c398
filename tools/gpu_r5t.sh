#!/bin/bash
# tools/gpu_r5t.sh -- round 5: batches between the staged schedule (<= 75 % of the CUs) and a full device: numerator chains beside the recursions on the CUs they
# leave (default) or behind them, beside the den half of the grad pass (ctc_after=1, what B >= 128 does)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for B in 100 104 112 120; do
  EXTRA="--B $B --steps 10" bash tools/gpu_ab3.sh default after+ctc_after=1 2>&1 | grep "pass 2" | sed "s/^/B=$B /"
done | tee $OUT/r5t_ab_after.txt
