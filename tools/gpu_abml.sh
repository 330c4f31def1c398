#!/bin/bash
# tools/gpu_abml.sh -- A/B of the multi-lane row butterfly (DPP vs __shfl_xor: cat_amd/lib_ab/libmlshfl.so built with -DCRF_AB_ML_SHFL)
OUT=$PWD/gpurun_out
for lib in default mlshfl default mlshfl; do
  if [ "$lib" == "default" ]; then L=""; else L=$PWD/cat_amd/lib_ab/lib$lib.so; fi
  CRF_LIB=$L timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --V 150 > $OUT/abml_$lib.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/abml_$lib.json')); k=d['roofline']['kernels_ms']; print('V=150 $lib: step %.3f den %.3f' % (d['ms_per_step'], k['den_fwd_chain']))"
  for a in "12000 800" "40000 2000"; do CRF_LIB=$L timeout 300 python tools/bench_fst.py $a 2>/dev/null | tail -2 | head -1 | sed "s/^/$lib est $a: /"; done
done
