#!/bin/bash
# tools/gpu_r5m.sh -- timelines (rocprofv3 --kernel-trace) of the grad pass schedules: VARIANTS="name:debug ..." 
OUT=$PWD/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-"stages:gd_stage_launches=1" "one:"}; do
  n=${v%%:*}; d=${v#*:}
  rm -rf /tmp/tl_$n
  CRF_DEBUG=$d timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$n -o trace --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 $EXTRA > /tmp/tl_$n.log 2>&1
  f=$(find /tmp/tl_$n -name "*kernel_trace.csv" | head -1)
  echo "== $n ($d)"; python $R/tools/tail_timeline.py $f 2>&1 | grep -v "elementwise\|stage_i32\|gate\|probe" | tail -22
done 2>&1 | tee $OUT/r5m_timelines_${TAG:-x}.txt
