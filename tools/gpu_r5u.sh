#!/bin/bash
# tools/gpu_r5u.sh -- round 5: numerator chains behind the recursions for every batch beyond the staged schedule's limit (default now): parity at B = 112, points
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_metric_shape.py -m gpu -q -x -k "112 or 96" 2>&1 | tail -2
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels_ms']
print('$name: %.0f utt/s, %.3f ms/step (median %.3f), den %.3f ctc %.3f grad %.3f' % (d['value'], d['ms_per_step'], d['event_blocks']['median_ms_per_step'], k.get('den_fwd_chain', -1), k.get('ctc_fwd_chain', -1), k.get('grad', -1)))"; }
( run B96 --B 96; run B100 --B 100; run B104 --B 104; run B112 --B 112; run B120 --B 120; run B128 --B 128 ) | tee $OUT/r5u_points.txt
