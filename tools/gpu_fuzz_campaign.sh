#!/bin/bash
# tools/gpu_fuzz_campaign.sh FIRST LAST -- EXTENDED fuzz campaigns beyond the suite's own seeds (tests/test_gpu_fuzz.py, CRF_FUZZ_CAMPAIGN): every
# campaign re-draws the 168 + 40 + 40 + 33 seeded cases (graphs, batches, shapes, scales) with the same coverage structure; one summary line each.
OUT=$PWD/gpurun_out; mkdir -p $OUT
for c in $(seq ${1:-1} ${2:-3}); do
  CRF_FUZZ_CAMPAIGN=$c timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -k "fuzz_vs_oracle or fused_log_softmax or long_utterances or fuzz_estimated" -p no:cacheprovider 2>&1 | tail -15 > $OUT/fuzz_campaign_$c.log
  echo "campaign $c: $(tail -1 $OUT/fuzz_campaign_$c.log)"
done | tee $OUT/fuzz_campaigns_${1:-1}_${2:-3}.txt
