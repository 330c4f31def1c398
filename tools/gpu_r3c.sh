#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "numerator or robust or warp_ctc or ctc_label or fused or edge_cases" > $OUT/pytest_num.log 2>&1; echo "numerator pytest rc=$?"; grep -v amdgpu.ids $OUT/pytest_num.log | tail -25
timeout 600 python tools/dbg_t3000.py 3000 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee $OUT/dbg_t3000b.txt
