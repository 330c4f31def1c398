#!/bin/bash
# tools/gpu_r5i.sh -- round 5: what the grad pass's per-frame check costs (build switch CRF_X_GCHK), fuzz on the simplified check
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shrink or peaked or underflow or robust" 2>&1 | tail -2
bash tools/gpu_ab3.sh default g0@gchk0 2>&1 | sed "s/^/metric /" | tee $OUT/r5i_ab.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh default g0@gchk0 2>&1 | sed "s/^/V217 /" | tee -a $OUT/r5i_ab.txt
EXTRA="--V 500" bash tools/gpu_ab3.sh default g0@gchk0 2>&1 | sed "s/^/V500 /" | tee -a $OUT/r5i_ab.txt
