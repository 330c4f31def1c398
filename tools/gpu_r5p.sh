#!/bin/bash
# tools/gpu_r5p.sh -- round 5: finalize folded into the last fallback launch, one-pass prep
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_ab3.sh default nofold+no_fin_fold=1 prev@prev 2>&1 | tee $OUT/r5p_ab.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_metric_shape.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "peaked or underflow or robust or fallback or edge or functional" 2>&1 | tail -3
