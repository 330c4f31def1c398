#!/bin/bash
# tools/gpu_r5q.sh -- round 5: grad pass of graphs with 257 .. 512 label chunks on 512 threads x one chunk (128 VGPRs, 16 waves per CU) vs 256 x two (220 VGPRs, 8 waves)
OUT=$PWD/gpurun_out; mkdir -p $OUT
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh default old+gd_w512=0 2>&1 | tee $OUT/r5q_ab_v217.txt
EXTRA="--V 300" bash tools/gpu_ab3.sh default old+gd_w512=0 2>&1 | grep "pass 2" | tee $OUT/r5q_ab_v300.txt
EXTRA="--V 400" bash tools/gpu_ab3.sh default old+gd_w512=0 2>&1 | grep "pass 2" | tee $OUT/r5q_ab_v400.txt
timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -2
