#!/bin/bash
# tools/gpu_r5g.sh -- round 5: log-domain fp64 fallback + forward/backward consistency check: fuzz, fallback tests, false positives at the bench shapes
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $OUT/r5g_fuzz.log 2>&1; tail -12 $OUT/r5g_fuzz.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "robust or underflow or shrink or peaked or edge_cases or large_graph or fixture or fused" > $OUT/r5g_pytest.log 2>&1; tail -4 $OUT/r5g_pytest.log
for a in "" "--ragged" "--T 3000 --steps 5" "--V 217 --lamb 0.01" "--histories 256 --fanout 16" "--histories 3072 --steps 5" "--histories 8192 --fanout 32 --steps 2 --warmup 1"; do
  timeout 600 python bench.py --no-cpu-baseline --steps 10 $a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', '->', d['ms_per_step'], 'ms/step, fallback', d.get('fallback_utterances'), d['roofline']['kernel'][:50])"
done 2>&1 | tee $OUT/r5g_false_positives.txt
