#!/bin/bash
# tools/gpu_r5b.sh -- round 5, second session: gathers-first frame (GFIRST) x lagged scale (LAG), one box
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import ctc_crf; print('switches', ctc_crf._C.build_switches())"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shrink_window or peaked or underflow or robust_fallback or edge_cases or synth_vs_oracle_ragged" > $OUT/r5b_pytest1.log 2>&1; tail -3 $OUT/r5b_pytest1.log
CRF_LIB=$PWD/cat_amd/lib_ab/libl1g1.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shrink_window or peaked or underflow" > $OUT/r5b_pytest_l1g1.log 2>&1; tail -3 $OUT/r5b_pytest_l1g1.log
bash tools/gpu_ab3.sh default l1g1@l1g1 l1g0@l1g0 l0g0@l0g0 2>&1 | tee $OUT/r5b_ab_metric.txt
EXTRA="--histories 256 --fanout 16" bash tools/gpu_ab3.sh default l1g1@l1g1 l0g0@l0g0 2>&1 | tee $OUT/r5b_ab_small.txt
EXTRA="--V 217 --lamb 0.01" bash tools/gpu_ab3.sh default l1g1@l1g1 l0g0@l0g0 2>&1 | tee $OUT/r5b_ab_v217.txt
CRF_LIB=$PWD/cat_amd/lib_ab/libtm.so timeout 300 python tools/timing_probe.py > $OUT/r5b_timing_tm.txt 2>&1
CRF_LIB=$PWD/cat_amd/lib_ab/libtm.so timeout 300 python tools/timing_probe.py 256 16 > $OUT/r5b_timing_tm_small.txt 2>&1
grep -A30 "den fwd CU 0" $OUT/r5b_timing_tm.txt | cut -c1-300 | head -20
