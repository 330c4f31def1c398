#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python tools/dbg_t3000.py 3000 2200 1500 2>&1 | grep -v amdgpu.ids | tee $OUT/dbg_t3000.txt
bash tools/gpu_ab3.sh default gfirst@gfirst
