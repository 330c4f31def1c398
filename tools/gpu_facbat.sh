#!/bin/bash
# tools/gpu_facbat.sh TAG -- factored streams of the utterance-minor kernels: parity tests, then the large-graph point with and
# without them (CRF_DEBUG=bat_no_fac=1).  Run under gpurun.
TAG=${1:-fb}
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/facbat_${TAG}_tests.txt
for e in "X=0" "CRF_DEBUG=bat_no_fac=1"; do
  env $e timeout 400 python bench.py --no-cpu-baseline --histories 8192 --fanout 32 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']
print('[$e] large: %.1f utt/s, %.3f ms/step, den %.2f ms, grad %.2f ms' % (d['value'], d['ms_per_step'], k.get('den_fwd_chain',-1), k.get('grad',-1)))"
done | tee $OUT/facbat_${TAG}_large.txt
for e in "X=0" "CRF_DEBUG=bat_no_fac=1"; do
  env $e timeout 600 python bench.py --no-cpu-baseline --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']
print('[$e] c5: %.1f utt/s, %.3f ms/step, den %.2f ms, grad %.2f ms' % (d['value'], d['ms_per_step'], k.get('den_fwd_chain',-1), k.get('grad',-1)))"
done | tee $OUT/facbat_${TAG}_c5.txt
