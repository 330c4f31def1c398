#!/bin/bash
# tools/gpu_r5c.sh -- round 5: emission-wave penalty of the layout's cost model (runtime switch res_emis, read at graph creation), one box
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_ab3.sh default e0+res_emis=0 e3+res_emis=3 e8+res_emis=8 e12+res_emis=12 2>&1 | tee $OUT/r5c_ab_metric.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh default e0+res_emis=0 e8+res_emis=8 2>&1 | tee $OUT/r5c_ab_v217.txt
EXTRA="--histories 256 --fanout 16" bash tools/gpu_ab3.sh default e0+res_emis=0 2>&1 | tee $OUT/r5c_ab_small.txt
