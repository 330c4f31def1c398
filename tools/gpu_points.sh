#!/bin/bash
# tools/gpu_points.sh TAG -- the other measurement points of SURVEY 8(d): ragged lengths, other batch shapes, small and
# large synthetic den_lm (one JSON line each, no CPU baseline)
TAG=${1:-r1}
OUT=$PWD/gpurun_out; mkdir -p $OUT
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $OUT/pt_${TAG}_$name.json 2> $OUT/pt_${TAG}_$name.err || tail -3 $OUT/pt_${TAG}_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/pt_${TAG}_$name.json"))
    k = d["roofline"]["kernels_ms"]
    print("$name: %.0f utt/s, %.3f ms/step, den fwd/bwd %.2f/%.2f ms, S=%s" % (d["value"], d["ms_per_step"], k.get("den_fwd_chain", -1), k.get("den_bwd_chain", -1), d["config"]["workload"].split("S=")[1].split(",")[0]))
except Exception as e:
    print("$name: no result", e)
PY
}
run ragged --ragged
run lamb001 --lamb 0.01
run B128 --B 128 --steps 10
run B32 --B 32
run B16 --B 16
run C2 --B 32 --T 500
run T3000 --T 3000 --steps 10
run small --histories 256 --fanout 16
run B256 --B 256 --steps 5
run B96 --B 96 --steps 10
run B80 --B 80 --steps 10
run V143 --V 143
run V217 --V 217 --lamb 0.01
run V500 --V 500
run mid3072 --histories 3072 --steps 5
run mid4096 --histories 4096 --steps 5
run large --histories 8192 --fanout 32 --steps 3 --warmup 1
# (the config #5 point is the long one -- a 4.3 M-arc graph to generate, compile and run at 148 ms per step: SKIP_C5=1 leaves it out of a short run)
[ -z "$SKIP_C5" ] && run c5 --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1
# estimated n-gram den_lm graphs (cat_amd.den_lm.prep_den_lm on a synthetic corpus): in-degree profile of a real LM
for a in "4000 250" "12000 800" "40000 2000"; do timeout 600 python tools/bench_fst.py $a 2>/dev/null | tail -3; done | tee $OUT/pt_${TAG}_estimated.txt
# ... and with the estimator's DEFAULT rule since round 6 (Kaldi's: a state per seen bigram history, extra states by log-likelihood): the shape real den_lm files have
for a in "4000 250" "12000 250" "40000 250"; do DEN_LM_SELECTION=likelihood timeout 600 python tools/bench_fst.py $a 2>/dev/null | tail -3; done | tee $OUT/pt_${TAG}_estimated_kaldi_rule.txt
