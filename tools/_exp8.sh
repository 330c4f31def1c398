#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for pc in 0 96 64 160 192 0; do echo "== S6836 piece=$pc"; CRF_DEBUG=piece=$pc timeout 300 python tools/bench_fst.py 40000 2000 2>/dev/null | tail -2 | cut -c1-200; done
echo "== V500"; EXTRA="--V 500" bash tools/gpu_ab3.sh default p96+piece=96 p160+piece=160 | grep -v "^pass 1"
