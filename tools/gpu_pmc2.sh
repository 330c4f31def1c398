#!/bin/bash
# tools/gpu_pmc2.sh TAG -- the short form of tools/gpu_prof.sh: bench line (no CPU baseline) + the two HBM-traffic PMC passes
TAG=${1:-t}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 60 python bench.py --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; cut -c1-200 $OUT/bench_$TAG.json
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
cd /tmp
CRF_DEBUG=trust_side=1 timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "fetch rc=$?"
CRF_DEBUG=trust_side=1 timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "write rc=$?"
find $OUT/pmc_*_$TAG -name "*kernel_trace.csv" -size +2M -delete
