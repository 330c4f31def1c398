#!/bin/bash
# tools/gpu_round.sh TAG [quick] -- the evidence run of a round on ONE box for the build in the tree (one parametrised script; the one-off
# lease scripts of round 5, tools/gpu_r5[a-v].sh, are gone: an A/B is `tools/build_ab.sh name:"-Dswitch"` + `tools/gpu_ab3.sh default name@name`): the GPU suite, the bench line (with the CPU
# baseline), rocprofv3 kernel stats + PMC passes (separate runs), the 1-rank torchrun line (RCCL initialised, DDP head), the other points of
# SURVEY 8(d), soak.  `quick`: without the GPU suite, the points and the soak.  tools/pmc_summary.py TAG condenses it into profiles/.
TAG=${1:-r6}; QUICK=$2
OUT=$PWD/gpurun_out; mkdir -p $OUT
if [ -z "$QUICK" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1500 > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_$TAG.log; tail -4 $OUT/pytest_$TAG.log
fi
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_$TAG.json
bash tools/gpu_prof.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -22
# (gpu_prof.sh leaves RCCL's banner in front of the torchrun record: keep the JSON line only)
grep '^{' $OUT/torchrun1_$TAG.json > $OUT/torchrun1_$TAG.tmp && mv $OUT/torchrun1_$TAG.tmp $OUT/torchrun1_$TAG.json
grep -c "ctc_crf_hip" $OUT/torchrun1_$TAG.err | sed 's/^/library warnings under torchrun: /'
if [ -z "$QUICK" ]; then
  bash tools/gpu_points.sh $TAG 2>&1 | grep -v amdgpu.ids
  (timeout 300 python tools/soak.py 600; timeout 300 python tools/soak.py 300 3072) > $OUT/soak_$TAG.txt 2>&1; tail -3 $OUT/soak_$TAG.txt
fi
