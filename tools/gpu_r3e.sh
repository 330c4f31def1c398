#!/bin/bash
for spec in "v217 --V 217" "v500 --V 500" "b96 --B 96" "b128 --B 128"; do set -- $spec; tag=$1; shift; echo "=== $tag $@"; bash tools/gpu_tail.sh $tag "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | head -40; done
