OUT=$PWD/gpurun_out; mkdir -p $OUT
for cfg in "3072 24" "4096 24" "6144 24"; do set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --histories $1 --fanout $2 --steps 5 --warmup 2 > $OUT/pt_mid_$1.json 2> $OUT/pt_mid_$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/pt_mid_$1.json")); k = d["roofline"]["kernels_ms"]
    print("H=$1 d=$2: %.0f utt/s, %.3f ms/step, den %.2f ms | %s | %s" % (d["value"], d["ms_per_step"], k.get("den_fwd_chain", -1), d["config"]["workload"].split(": ")[1][:50], d["config"]["den_kernels"][:70]))
except Exception as e:
    print("H=$1: no result", e); print(open("$OUT/pt_mid_$1.err").read()[-500:])
PY
done
