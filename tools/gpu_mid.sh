#!/bin/bash
# tools/gpu_mid.sh -- graphs between the benchmark graph and the large one (same generator, d = 24): which kernels take them, how
# fast; "nores": the same with CRF_DEBUG=no_resident=1 (utterance-minor kernels instead of the generic K-CU layout)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for cfg in "3072 24 0" "3072 24 1" "4096 24 0" "4096 24 1" "6144 24 0"; do set -- $cfg
  CRF_DEBUG=no_resident=$3 timeout 300 python bench.py --no-cpu-baseline --histories $1 --fanout $2 --steps 5 --warmup 2 > $OUT/pt_mid_$1_$3.json 2> $OUT/pt_mid_$1_$3.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/pt_mid_$1_$3.json")); k = d["roofline"]["kernels_ms"]
    print("H=$1 d=$2 nores=$3: %.0f utt/s, %.3f ms/step, den %.2f ms | %s | %s" % (d["value"], d["ms_per_step"], k.get("den_fwd_chain", -1), d["config"]["workload"].split(": ")[1][:40], d["config"]["den_kernels"][:50]))
except Exception as e:
    print("H=$1: no result", e); print(open("$OUT/pt_mid_$1_$3.err").read()[-500:])
PY
done
