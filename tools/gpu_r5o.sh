#!/bin/bash
# tools/gpu_r5o.sh -- round 5: one-launch grad pass (candidate-major grid), piece sweep
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_ab3.sh default p48+piece=48 p64+piece=64 p80+piece=80 p96+piece=96 p112+piece=112 stages+gd_stage_launches=1 2>&1 | tee $OUT/r5o_ab.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh default p64+piece=64 p80+piece=80 p96+piece=96 2>&1 | grep "pass 2" | tee $OUT/r5o_ab_v217.txt
EXTRA="--B 96" bash tools/gpu_ab3.sh default p64+piece=64 p80+piece=80 p96+piece=96 2>&1 | grep "pass 2" | tee $OUT/r5o_ab_b96.txt
EXTRA="--T 700" bash tools/gpu_ab3.sh default p64+piece=64 p80+piece=80 p96+piece=96 stages+gd_stage_launches=1 2>&1 | grep "pass 2" | tee $OUT/r5o_ab_t700.txt
