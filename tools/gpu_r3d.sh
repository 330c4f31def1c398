#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_two_processes.py -m gpu -q -x --timeout 800 > $OUT/pytest_2p.log 2>&1; echo "two-process pytest rc=$?"; grep -v amdgpu.ids $OUT/pytest_2p.log | tail -15
bash tools/gpu_ab3.sh default rcl+fac_rcl=1
( cd /tmp; rocprofv3 -L > $OUT/rocprof_counters.txt 2>&1; grep -c . $OUT/rocprof_counters.txt
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_inst -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_inst.log 2>&1; echo "pmc inst rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_inst2 -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_inst2.log 2>&1; echo "pmc inst2 rc=$?" )
python - <<'PY'
import csv, glob, collections
for d in ("pmc_inst", "pmc_inst2"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "fac_pair" in r["Kernel_Name"]: acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: int(sum(v) / len(v)) for c, v in cs.items()})
PY
find $OUT/pmc_inst $OUT/pmc_inst2 -name "*kernel_trace.csv" -size +1M -delete
