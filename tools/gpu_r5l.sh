#!/bin/bash
# tools/gpu_r5l.sh -- round 5: ONE grad launch for all stages (workgroups wait themselves) vs per-stage launches behind stream waits; tapered last pieces,
# short last stages split among 16 / gd_sub workgroups per block
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_ab3.sh stages+gd_stage_launches=1 one one_t32+taper=32 t32_s16+taper=32,gd_sub=16 t32_s8+taper=32,gd_sub=8 t16+taper=16 t16_s2+taper=16,gd_sub=2 t64+taper=64 2>&1 | tee $OUT/r5l_ab.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh stages+gd_stage_launches=1 one_t32+taper=32 t16+taper=16 2>&1 | grep "pass 2" | tee $OUT/r5l_ab_v217.txt
EXTRA="--B 96" bash tools/gpu_ab3.sh stages+gd_stage_launches=1 one_t32+taper=32 t16+taper=16 2>&1 | grep "pass 2" | tee $OUT/r5l_ab_b96.txt
EXTRA="--B 128 --steps 10" bash tools/gpu_ab3.sh default 2>&1 | grep "pass 2" | tee $OUT/r5l_ab_b128.txt
CRF_DEBUG=taper=32 timeout 900 python -m pytest tests/test_gpu_metric_shape.py tests/test_gpu_under_nccl.py -m gpu -q -x 2>&1 | tail -3
CRF_DEBUG=taper=16 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "numerator or ctc or label" 2>&1 | tail -3
