cd /root/repo
for c in 0 9 10; do CRF_FUZZ_CAMPAIGN=$c timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider 2>&1 | grep -E "^E  |FAILED|passed|failed" | cut -c1-700 > gpurun_out/fuzz_campaign_$c.log; echo "campaign $c: $(tail -1 gpurun_out/fuzz_campaign_$c.log)"; done | tee gpurun_out/fuzz_campaigns_c.txt
for a in "" "--T 3000 --steps 5" "--ragged" "--V 217 --lamb 0.01" "--B 128 --steps 10"; do timeout 600 python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['fallback_utterances'])"; done | tee gpurun_out/fallback_counts_after_check.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "ctc or numerator or warp" 2>&1 | tail -2
