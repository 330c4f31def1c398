#!/bin/bash
# tools/gpu_r5e.sh -- round 5: the two-utterance kernel on its own layout (512 threads x 30 chunks): parity, then batches above one staged device-full
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pair2_512 or two_utterances or default_kernel_for_a_batch" > $OUT/r5e_pytest.log 2>&1; tail -4 $OUT/r5e_pytest.log
for B in 128 256 192 96; do
  EXTRA="--B $B --steps 10" bash tools/gpu_ab3.sh p2+fac_pair2=1 one+fac_pair2=0 2>&1 | sed "s/^/B=$B /" | tee -a $OUT/r5e_ab_batches.txt
done
EXTRA="--B 64" bash tools/gpu_ab3.sh p2+fac_pair2=1 one+fac_pair2=0 2>&1 | sed "s/^/B=64 /" | tee -a $OUT/r5e_ab_batches.txt
