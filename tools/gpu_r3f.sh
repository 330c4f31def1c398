#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 --deselect tests/test_gpu_config5.py > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_all.log
for spec in "v217 --V 217" "v500 --V 500" "v143 --V 143"; do set -- $spec; tag=$1; shift; echo "=== $tag $@"; bash tools/gpu_tail.sh $tag "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | grep -v "at::native\|rocclr\|stage_i32" | head -30; done
bash tools/gpu_ab3.sh default
