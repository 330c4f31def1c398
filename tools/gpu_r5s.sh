#!/bin/bash
# tools/gpu_r5s.sh -- round 5: grad den kernel of the metric-size graphs at 128 registers (four workgroups per CU, gathers in batches of 8) vs 153 (three)
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_ab3.sh default wg4+gd_wg4=1 2>&1 | tee $OUT/r5s_ab.txt
EXTRA="--V 143" bash tools/gpu_ab3.sh default wg4+gd_wg4=1 2>&1 | grep "pass 2" | tee $OUT/r5s_ab_v143.txt
EXTRA="--B 96" bash tools/gpu_ab3.sh default wg4+gd_wg4=1 2>&1 | grep "pass 2" | tee $OUT/r5s_ab_b96.txt
EXTRA="--B 128 --steps 10" bash tools/gpu_ab3.sh default wg4+gd_wg4=1 2>&1 | grep "pass 2" | tee $OUT/r5s_ab_b128.txt
CRF_DEBUG=gd_wg4=1 timeout 600 python -m pytest tests/test_gpu_metric_shape.py -m gpu -q -x 2>&1 | tail -2
