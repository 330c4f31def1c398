#!/bin/bash
# tools/gpu_gradarr.sh TAG -- A/B of the bank-arranged pair order of the grad pass (CRF_DEBUG=no_grad_arrange=1 = as listed)
TAG=${1:-ga}
OUT=$PWD/gpurun_out
mkdir -p $OUT
ROOT=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or estimated or benchmarked or fixture" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/ga_${TAG}_tests.txt
cd /tmp && export TMPDIR=/tmp
for e in "X=0" "CRF_DEBUG=no_grad_arrange=1"; do
  rm -rf /tmp/prof_ga
  env $e timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ga -o ga --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/[$e] /"
  f=$(find /tmp/prof_ga -name "*kernel_stats.csv" | head -1)
  grep -E "grad_den|fac_pair" $f | cut -c1-160 | sed "s/^/[$e] /"
  env $e timeout 300 python $ROOT/tools/bench_fst.py 40000 2000 2>/dev/null | tail -2 | sed "s/^/[$e] /"
done | tee $OUT/ga_${TAG}_ab.txt
