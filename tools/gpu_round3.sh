#!/bin/bash
# tools/gpu_round3.sh TAG [what...] -- one gpurun call of round 2.  what: tests bench hwq prof pmc points (default: tests bench hwq prof)
TAG=${1:-r2a}; shift
WHAT=${@:-tests bench hwq prof}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for w in $WHAT; do case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_$TAG.log
  tail -12 $OUT/pytest_$TAG.log ;;
bench)
  timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err ;;
hwq)
  for q in "" 4 8; do for n in 0 2 6; do
    if [ -z "$q" ]; then env -u GPU_MAX_HW_QUEUES timeout 200 python tools/hwq_probe.py $n 2>&1 | tail -1; else GPU_MAX_HW_QUEUES=$q timeout 200 python tools/hwq_probe.py $n 2>&1 | tail -1; fi
  done; done | tee $OUT/hwq_$TAG.txt ;;
prof)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?" )
  F=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -14 "$F"
  find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +4M -delete ;;
pmc)
  BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
  ( cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq_$TAG -o pmc -- $BENCH > $OUT/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?" )
  find $OUT/pmc_*_$TAG -name "*kernel_trace.csv" -size +2M -delete ;;
large)
  bash tools/gpu_large.sh $TAG ;;
points)
  bash tools/gpu_points.sh $TAG ;;
esac; done
