#!/bin/bash
# tools/gpu_batch.sh TAG [tm] -- the utterance-minor kernels on one box: parity (both forms: persistent launch / one launch per frame), the
# large-graph points persistent vs per-frame, and with `tm` the in-kernel phase stamps of a timing build (cat_amd/lib_ab/libtm.so).
TAG=${1:-r6}; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch" > $OUT/${TAG}_pytest_batch.log 2>&1; tail -2 $OUT/${TAG}_pytest_batch.log
pt() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || tail -3 $OUT/${TAG}_$name.err
  python -c "
import json
try:
    d = json.load(open('$OUT/${TAG}_$name.json')); k = d['roofline']['kernels_ms']
    print('$name: %.1f utt/s  %.3f ms/step  den %.2f ms  grad %.2f  %s' % (d['value'], d['ms_per_step'], k.get('den_fwd_chain', -1), k.get('grad', -1), d['roofline']['kernel'].split(' ')[0]))
except Exception as e: print('$name: no result', e)"; }
LARGE="--histories 8192 --fanout 32 --steps 3 --warmup 1"
pt large_persist $LARGE
CRF_DEBUG=bat_persist=0 pt large_frames $LARGE
if [ "$2" != "quick" ]; then
  pt h6144_persist --histories 6144 --fanout 24 --steps 3 --warmup 1
  pt c5_persist --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1
  CRF_DEBUG=bat_persist=0 pt c5_frames --B 8 --T 3000 --V 5000 --histories 32768 --fanout 64 --steps 2 --warmup 1
fi
if [ -f cat_amd/lib_ab/libtm.so ]; then
  CRF_LIB=$PWD/cat_amd/lib_ab/libtm.so timeout 300 python tools/timing_probe_batch.py > $OUT/${TAG}_timing_persist.txt 2>&1; grep "^wg" $OUT/${TAG}_timing_persist.txt | head -6 | cut -c1-400
fi
