#!/bin/bash
# tools/gpu_ab_c5_fill.sh -- BASELINE config #5 (S = 65 537, 4.3 M arcs, V = 5 000, T = 3 000, B = 8): the utterance-minor grid cut for 70 % of the device's
# workgroup slots (default: 800 workgroups > 512 slots, one launch per frame) against cuts that make the grid co-resident (one persistent launch).
OUT=$PWD/gpurun_out; mkdir -p $OUT
for sw in "" "bat_fill=35" "bat_fill=30" "bat_fill=35,bat_persist=0"; do
  CRF_DEBUG=verbose,$sw timeout 900 python bench.py --no-cpu-baseline --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1 2> $OUT/c5_fill.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms']; print('[$sw]: step %.2f ms  den %.2f ms  %s  fallback %s' % (d['ms_per_step'], k['den_fwd_chain'], d['roofline']['kernel'].split(' ')[0], d['fallback_utterances']))
except Exception as e: print('[$sw]: no result', e)"
  grep "utterance-minor" $OUT/c5_fill.err | tail -1
done | tee $OUT/ab_c5_fill.txt
