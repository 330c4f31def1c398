#!/bin/bash
# tools/gpu_round.sh -- what one gpurun call does: GPU tests, the bench line, rocprofv3 kernel stats.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag] [bench args...]
TAG=${1:-r1}; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_$TAG.log
tail -15 $OUT/pytest_$TAG.log
timeout 900 python bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cat $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?" )
F=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -12 "$F"
# keep only the small summaries (the raw trace can be tens of MB)
find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +4M -delete
