#!/bin/bash
# tools/gpu_r5a.sh -- round 5, first GPU session: parity of the lagged scale (product build), A/B of the build switches on one box
# (lagged scale, late row constants, stage length), re-stamped phase tables of the dominant kernel (timing builds, both scale rules).
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import ctc_crf; print('switches', ctc_crf._C.build_switches())"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lagged or peaked or underflow or robust_fallback or edge_cases or synth_vs_oracle_ragged" > $OUT/r5a_pytest1.log 2>&1; tail -3 $OUT/r5a_pytest1.log
timeout 900 python -m pytest tests/test_gpu_metric_shape.py tests/test_gpu_under_nccl.py -m gpu -x -q > $OUT/r5a_pytest2.log 2>&1; tail -3 $OUT/r5a_pytest2.log
bash tools/gpu_ab3.sh default lag0@lag0 kcl@kcl p96+piece=96 2>&1 | tee $OUT/r5a_ab_metric.txt
EXTRA="--histories 256 --fanout 16" bash tools/gpu_ab3.sh default lag0@lag0 kcl@kcl 2>&1 | tee $OUT/r5a_ab_small.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh default lag0@lag0 2>&1 | tee $OUT/r5a_ab_v217.txt
for v in tm tm0; do
  CRF_LIB=$PWD/cat_amd/lib_ab/lib$v.so timeout 300 python tools/timing_probe.py > $OUT/r5a_timing_$v.txt 2>&1
  CRF_LIB=$PWD/cat_amd/lib_ab/lib$v.so timeout 300 python tools/timing_probe.py 256 16 > $OUT/r5a_timing_${v}_small.txt 2>&1
done
grep -A14 "den fwd CU 0" $OUT/r5a_timing_tm.txt | head -16
