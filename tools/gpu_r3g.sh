#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "two_utterances or pair2" > $OUT/pytest_pair2.log 2>&1; echo "pair2 pytest rc=$?"; grep -v amdgpu.ids $OUT/pytest_pair2.log | tail -30
EXTRA="--B 128 --steps 10" bash tools/gpu_ab3.sh b128_one+fac_pair2=0 b128_two
EXTRA="--B 96 --steps 10" bash tools/gpu_ab3.sh b96_one+fac_pair2=0 b96_two
EXTRA="--B 256 --steps 5" bash tools/gpu_ab3.sh b256_one+fac_pair2=0 b256_two
EXTRA="--B 64" bash tools/gpu_ab3.sh b64_two+fac_pair2=1
for spec in "v500 --V 500" "b128 --B 128"; do set -- $spec; tag=$1; shift; echo "=== $tag $@"; bash tools/gpu_tail.sh $tag "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | grep -v "at::native\|rocclr\|stage_i32" | head -30; done
