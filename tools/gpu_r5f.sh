#!/bin/bash
# tools/gpu_r5f.sh -- round 5: numerator half and den half of the grad pass on two streams (grad_par3), parity then A/B
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metric_shape.py tests/test_gpu_callers.py tests/test_gpu_under_nccl.py -m gpu -x -q -k "not batch and not streaming and not resident and not k2" > $OUT/r5f_pytest.log 2>&1; tail -4 $OUT/r5f_pytest.log
bash tools/gpu_ab3.sh default two+grad_par3=0 2>&1 | sed "s/^/metric /" | tee $OUT/r5f_ab.txt
EXTRA="--histories 256 --fanout 16" bash tools/gpu_ab3.sh default two+grad_par3=0 2>&1 | sed "s/^/S513 /" | tee -a $OUT/r5f_ab.txt
EXTRA="--B 32 --T 500" bash tools/gpu_ab3.sh default two+grad_par3=0 2>&1 | sed "s/^/C2 /" | tee -a $OUT/r5f_ab.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh default two+grad_par3=0 2>&1 | sed "s/^/V217 /" | tee -a $OUT/r5f_ab.txt
EXTRA="--ragged" bash tools/gpu_ab3.sh default two+grad_par3=0 2>&1 | sed "s/^/ragged /" | tee -a $OUT/r5f_ab.txt
for a in "4000 250" "12000 800" "40000 2000"; do
  for sw in "" "grad_par3=0"; do echo "estimated $a [$sw]"; CRF_DEBUG=$sw timeout 600 python tools/bench_fst.py $a 2>/dev/null | tail -3; done
done | tee $OUT/r5f_estimated.txt
