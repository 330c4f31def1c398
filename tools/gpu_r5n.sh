#!/bin/bash
# tools/gpu_r5n.sh -- round 5: the one-launch grad pass, sweep of the last stages (default now: taper 32, whole blocks)
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_ab3.sh stages+gd_stage_launches=1 default s8+gd_sub=8 s4+gd_sub=4 t48+taper=48 t64+taper=64 p96+piece=96 p160+piece=160 p192+piece=192 2>&1 | tee $OUT/r5n_ab.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh stages+gd_stage_launches=1 default s8+gd_sub=8 2>&1 | grep "pass 2" | tee $OUT/r5n_ab_v217.txt
EXTRA="--B 96" bash tools/gpu_ab3.sh stages+gd_stage_launches=1 default s8+gd_sub=8 2>&1 | grep "pass 2" | tee $OUT/r5n_ab_b96.txt
EXTRA="--T 3000 --steps 10" bash tools/gpu_ab3.sh stages+gd_stage_launches=1 default 2>&1 | grep "pass 2" | tee $OUT/r5n_ab_t3000.txt
EXTRA="--B 32" bash tools/gpu_ab3.sh stages+gd_stage_launches=1 default 2>&1 | grep "pass 2" | tee $OUT/r5n_ab_b32.txt
