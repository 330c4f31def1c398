#!/bin/bash
# tools/gpu_fill.sh -- utterance-minor kernels: share of the device's workgroup slots one launch takes (CRF_BAT_FILL, percent)
OUT=gpurun_out; mkdir -p $OUT
for f in 100 85 70 55 45 35; do
  CRF_DEBUG=bat_fill=$f timeout 400 python bench.py --no-cpu-baseline --histories 8192 --fanout 32 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']
print('fill $f large: %.1f utt/s, %.3f ms/step, den %.2f ms' % (d['value'], d['ms_per_step'], k.get('den_fwd_chain',-1)))"
done | tee $OUT/fill_large.txt
for f in 100 70 50; do
  CRF_DEBUG=bat_fill=$f timeout 600 python bench.py --no-cpu-baseline --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']
print('fill $f c5: %.1f utt/s, %.3f ms/step, den %.2f ms' % (d['value'], d['ms_per_step'], k.get('den_fwd_chain',-1)))"
done | tee $OUT/fill_c5.txt
