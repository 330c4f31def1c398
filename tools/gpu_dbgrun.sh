# tools/gpu_dbgrun.sh lib1 lib2 ... -- the large-graph point with alternative builds of the library (cat_amd/lib_ab/lib<name>.so; "default" = the product build)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for v in "$@"; do
  if [ "$v" == "default" ]; then L=""; else L=$PWD/cat_amd/lib_ab/lib$v.so; fi
  CRF_LIB=$L timeout 300 python bench.py --no-cpu-baseline --histories 8192 --fanout 32 --steps 3 --warmup 1 > $OUT/dbg_$v.json 2> $OUT/dbg_$v.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/dbg_$v.json")); k = d["roofline"]["kernels_ms"]
    print("$v: %.3f ms/step, den %.2f ms, loss %s" % (d["ms_per_step"], k.get("den_fwd_chain", -1), d.get("loss")))
except Exception as e:
    print("$v: no result", e); print(open("$OUT/dbg_$v.err").read()[-600:])
PY
done
