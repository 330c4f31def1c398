#!/bin/bash
# tools/gpu_tail.sh tag -- rocprofv3 kernel trace of a short bench run + the timeline of one call (tools/tail_timeline.py)
TAG=${1:-t}; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$TAG -o trace -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/tl_$TAG.log 2>&1; echo "rocprof rc=$?" )
F=$(find $OUT/tl_$TAG -name "*kernel_trace.csv" | head -1)
python tools/tail_timeline.py "$F" | tee $OUT/tl_$TAG.txt
rm -rf $OUT/tl_$TAG
