#!/bin/bash
# tools/gpu_k2.sh TAG -- factored layout over two CUs per recursion: parity tests, then the mid-size graphs with and without it
TAG=${1:-k2}
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_callers.py -x -q -m gpu -k "k2 or two_cus" --timeout 300 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/k2_${TAG}_tests.txt
for cfg in "2304 0" "3072 0" "3072 1" "4096 0" "4096 1"; do set -- $cfg
  CRF_DEBUG=fac_no_k2=$2 timeout 200 python bench.py --no-cpu-baseline --histories $1 --fanout 24 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']
    print('H=$1 no_k2=$2: %.0f utt/s, %.3f ms/step, den %.2f ms, grad %.2f | %s' % (d['value'], d['ms_per_step'], k.get('den_fwd_chain',-1), k.get('grad',-1), d['config']['den_kernels'][:60]))
except Exception as e:
    print('H=$1 no_k2=$2: no result', e)"
done | tee $OUT/k2_${TAG}_mid.txt
