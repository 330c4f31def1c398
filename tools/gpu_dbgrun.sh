OUT=$PWD/gpurun_out; mkdir -p $OUT
for v in dbg1 dbg2; do
  CRF_LIB=$PWD/cat_amd/lib_ab/lib$v.so timeout 300 python bench.py --no-cpu-baseline --histories 8192 --fanout 32 --steps 3 --warmup 1 > $OUT/dbg_$v.json 2> $OUT/dbg_$v.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/dbg_$v.json")); k = d["roofline"]["kernels_ms"]
    print("$v: %.3f ms/step, den %.2f ms" % (d["ms_per_step"], k.get("den_fwd_chain", -1)))
except Exception as e:
    print("$v: no result", e); print(open("$OUT/dbg_$v.err").read()[-600:])
PY
done
