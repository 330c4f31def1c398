#!/bin/bash
# tools/gpu_r5k.sh -- round 5: the grad den kernel back at <= 152 registers: the points that had regressed, the K2 emission-wave A/B, then fuzz + parity
OUT=$PWD/gpurun_out; mkdir -p $OUT
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels_ms']
print('$name: %.0f utt/s, %.3f ms/step (median %.3f), den %.3f ctc %.3f grad %.3f, fallback %s' % (d['value'], d['ms_per_step'], d['event_blocks']['median_ms_per_step'], k.get('den_fwd_chain', -1), k.get('ctc_fwd_chain', -1), k.get('grad', -1), d.get('fallback_utterances')))"; }
( run metric; run B128 --B 128 --steps 10; run B256 --B 256 --steps 5; run B192 --B 192 --steps 5; run B96 --B 96 --steps 10; run small --histories 256 --fanout 16; run V217 --V 217 --lamb 0.01
  run mid3072 --histories 3072 --steps 5; CRF_DEBUG=res_emis=0 run mid3072_emis0 --histories 3072 --steps 5
  run mid4096 --histories 4096 --steps 5; CRF_DEBUG=res_emis=0 run mid4096_emis0 --histories 4096 --steps 5
  run mid3072 --histories 3072 --steps 5; CRF_DEBUG=res_emis=0 run mid3072_emis0 --histories 3072 --steps 5
  run large --histories 8192 --fanout 32 --steps 3 --warmup 1; run T3000 --T 3000 --steps 10; run metric ) | tee $OUT/r5k_points.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
