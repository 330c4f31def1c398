#!/bin/bash
# tools/gpu_prof.sh TAG -- rocprofv3 kernel stats + PMC passes (separate runs) for the default bench workload
TAG=${1:-r1}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- $BENCH > $OUT/prof_$TAG.log 2>&1; echo "stats rc=$?"
head -8 $(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
# PMC passes: HBM traffic (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -> separate passes), then LDS / occupancy
CRF_DEBUG=trust_side=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
CRF_DEBUG=trust_side=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
CRF_DEBUG=trust_side=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq_$TAG -o pmc -- $BENCH > $OUT/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?"
find $OUT/prof_$TAG $OUT/pmc_*_$TAG -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT/pmc_fetch_$TAG $OUT/pmc_sq_$TAG 2>/dev/null | head
cd $REPO
# single-rank torchrun smoke of the N>1 code path (NCCL init, DDP head) -- the driver runs N=2,4,8 on an 8-GPU node
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --ddp-head > $OUT/torchrun1_$TAG.json 2> $OUT/torchrun1_$TAG.err; echo "torchrun rc=$?"; tail -c 600 $OUT/torchrun1_$TAG.json
