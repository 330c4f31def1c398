#!/bin/bash
# tools/gpu_ab_k2_geometry.sh -- two CUs per recursion on 768 x 20 (default) against 1024 x 15 (fac_threads=1024) for den_lms with many rows and few arcs
# (the default estimator rule: a state per seen bigram history); one box.
OUT=$PWD/gpurun_out; mkdir -p $OUT
{
for rep in 1 2; do
for a in "4000 250" "40000 250"; do
  for sw in "" "fac_k2,fac_threads=1024"; do
    echo "== corpus $a  CRF_DEBUG=$sw"
    CRF_DEBUG=$sw DEN_LM_SELECTION=likelihood timeout 600 python tools/bench_fst.py $a 2>/dev/null | tail -3
  done
done
done
} | tee $OUT/ab_k2_geometry.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
