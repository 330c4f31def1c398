#!/bin/bash
# first GPU call of round 3: new parity tests, the whole GPU suite, A/B of the pipelined gathers, in-kernel stamps
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_config5.py -m gpu -q -x -s --timeout 1200 > $OUT/pytest_c5.log 2>&1; echo "c5 pytest rc=$?"; grep -v amdgpu.ids $OUT/pytest_c5.log | tail -25
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 --deselect tests/test_gpu_config5.py > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_all.log
CRF_DEBUG=fac_pipe=1 timeout 900 python -m pytest tests/test_gpu_metric_shape.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "metric_shape or synth_vs_oracle or edge_cases or estimated or factored_sched or robust or fused" > $OUT/pytest_pipe.log 2>&1; echo "pipe pytest rc=$?"; tail -4 $OUT/pytest_pipe.log
bash tools/gpu_ab3.sh default pipe+fac_pipe=1 p4@pipe4+fac_pipe=1
for l in tim timpipe; do echo "== timing $l"; CRF_LIB=$PWD/cat_amd/lib_ab/lib$l.so timeout 300 python tools/timing_probe.py 2>&1 | grep -v amdgpu.ids | head -16; done
