#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" 2>&1 | grep -v amdgpu.ids | tail -2
for f in 100 70 100 70 85 60; do
  CRF_BAT_FILL=$f timeout 400 python bench.py --no-cpu-baseline --histories 8192 --fanout 32 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']
print('fill $f large: %.1f utt/s, %.3f ms/step, den %.2f ms' % (d['value'], d['ms_per_step'], k.get('den_fwd_chain',-1)))"
done | tee $OUT/fill2_large.txt
