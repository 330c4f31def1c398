#!/bin/bash
# tools/gpu_kstat.sh -- per-kernel average durations (rocprofv3 --kernel-trace --stats) for bench variants
# selected by environment variables.  usage: bash tools/gpu_kstat.sh tag "ENV1=.." "" ...
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for V in "$@"; do
  echo "== variant $i: [$V]"
  ( cd /tmp && env $V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_${TAG}_$i -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/ks_${TAG}_$i.log 2>&1 )
  F=$(find $OUT/ks_${TAG}_$i -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "crf" in r["Name"]:
        print("  %-60s %9.1f us" % (r["Name"][:60], float(r["AverageNs"]) / 1e3))
PY
  find $OUT/ks_${TAG}_$i -name "*kernel_trace.csv" -delete
  i=$((i+1))
done
