#!/bin/bash
# tools/gpu_r4a.sh TAG -- round 4, item 1: which kind of side stream runs beside the caller's in a process with RCCL + DDP + busy streams
TAG=${1:-r4a}
OUT=$PWD/gpurun_out; mkdir -p $OUT
R=$OUT/nccl_probe_$TAG.txt; : > $R
run() { echo "## $*" >> $R; ( "$@" ) >> $R 2> $OUT/nccl_probe_$TAG.err.tmp; grep -c "ctc_crf_hip" $OUT/nccl_probe_$TAG.err.tmp | sed 's/^/library warnings: /' >> $R; grep "ctc_crf_hip" $OUT/nccl_probe_$TAG.err.tmp >> $R; }
run timeout 300 python tools/nccl_probe.py --streams 8
run env CRF_DEBUG=side_kind=1 timeout 300 python tools/nccl_probe.py --streams 8
run env CRF_DEBUG=side_kind=3 timeout 300 python tools/nccl_probe.py --streams 8
run env CRF_DEBUG=side_kind=2 timeout 300 python tools/nccl_probe.py --streams 8
run env CRF_DEBUG=side_kind=4 timeout 300 python tools/nccl_probe.py --streams 8
run timeout 300 python tools/nccl_probe.py --streams 0 --no-nccl
run timeout 300 python tools/nccl_probe.py --streams 24
run env CRF_DEBUG=no_side_stream=1 timeout 300 python tools/nccl_probe.py --streams 8
cat $R
