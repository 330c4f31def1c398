#!/bin/bash
# tools/gpu_rcl.sh TAG -- row constants in an LDS table (768-thread factored kernels, fac_geom 1): parity tests of the new
# variant, then A/B against the other geometries on the estimated graphs and on the benchmark graph.  Run under gpurun.
TAG=${1:-rcl}
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rcl or many_rows or estimated" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/rcl_${TAG}_tests.txt
for a in "4000 250" "12000 800" "40000 2000"; do
  for e in "X=0" "CRF_DEBUG=fac_rcl=1" "CRF_DEBUG=fac_no_rcl=1"; do
    env $e timeout 300 python tools/bench_fst.py $a 2>/dev/null | tail -3 | sed "s/^/[$e] /"
  done
done | tee $OUT/rcl_${TAG}_estimated.txt
for e in "X=0" "CRF_DEBUG=fac_rcl=1"; do
  env $e timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-420 | sed "s/^/[$e] /"
done | tee $OUT/rcl_${TAG}_bench.txt
