#!/bin/bash
# tools/build_ab.sh name:"-Ddefs" ... -- A/B builds of the library into cat_amd/lib_ab/lib<name>.so, in parallel (hipcc cross-compiles
# without a GPU; ~45 s each, the kernel families compile in parallel).  `prod` as a name rebuilds the product library cat_amd/lib/libctc_crf_hip.so.
#   tools/build_ab.sh prod lag0:"-DCRF_X_LAG=0" tm:"-DCRF_TIMING"
mkdir -p cat_amd/lib_ab
for v in "$@"; do
  n=${v%%:*}; d=""; case "$v" in *:*) d=${v#*:};; esac
  if [ "$n" = prod ]; then
    (python -m cat_amd.build --force > /tmp/build_$n.log 2>&1; echo "$n rc=$?") &
  else
    (CRF_BUILD_OUT=$PWD/cat_amd/lib_ab/lib$n.so CRF_BUILD_DEFS="$d" python -m cat_amd.build --force > /tmp/build_$n.log 2>&1; echo "$n rc=$?") &
  fi
done
wait
