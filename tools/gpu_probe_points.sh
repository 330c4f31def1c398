#!/bin/bash
# tools/gpu_probe_points.sh -- probe points around the limits: a graph just above the factored capacity, larger vocabularies, short
# utterances, a batch between 64 and 128 per GPU (one JSON each under gpurun_out/pr_*.json)
OUT=$PWD/gpurun_out; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/pr_$name.json 2> $OUT/pr_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/pr_$name.json")); k = d["roofline"]["kernels_ms"]
    print("$name: %.0f utt/s, %.3f ms/step | %s | %s | %s" % (d["value"], d["ms_per_step"], {a: round(b, 3) for a, b in k.items()}, d["config"]["workload"].split(": ")[1][:40], d["config"]["den_kernels"][:40]))
except Exception as e:
    print("$name: no result", e); print(open("$OUT/pr_$name.err").read()[-400:])
PY
}
run H2304 --histories 2304 --fanout 24
run H2560 --histories 2560 --fanout 24
run V500 --V 500
run V1000 --V 1000
run T200 --T 200
run B96 --B 96
