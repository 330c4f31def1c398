cd /root/repo
for c in 1 5 6 7 8; do CRF_FUZZ_CAMPAIGN=$c timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider -k "fuzz_vs_oracle or fused or long_utt" 2>&1 | grep -E "^E  |FAILED|passed|failed" | cut -c1-700 > gpurun_out/fuzz_campaign_$c.log; echo "campaign $c: $(tail -1 gpurun_out/fuzz_campaign_$c.log)"; done | tee gpurun_out/fuzz_campaigns_b.txt
