#!/bin/bash
# tools/gpu_round3.sh TAG -- round 3, evidence run: the whole GPU suite, the bench line, rocprofv3 kernel stats + PMC passes, the other points of SURVEY 8(d)
TAG=${1:-r3}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1500 > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_$TAG.log; tail -6 $OUT/pytest_$TAG.log
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cut -c1-700 $OUT/bench_$TAG.json
bash tools/gpu_prof.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -25
bash tools/gpu_points.sh $TAG 2>&1 | grep -v amdgpu.ids
(timeout 300 python tools/soak.py 600; timeout 300 python tools/soak.py 300 3072) > $OUT/soak_$TAG.txt 2>&1; tail -3 $OUT/soak_$TAG.txt
