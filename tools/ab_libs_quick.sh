#!/bin/bash
# A/B of two builds of libpffft_hip.so on ONE box with tools/r6_quick.py (box-to-box spread is as large as most effects):
#     gpurun -- 'bash tools/ab_libs_quick.sh gpurun_exp/libA.so gpurun_exp/libB.so 2 16384:r:f32 8192:c:f32 ...'
set -u
A=$1; B=$2; REPS=$3; shift 3
cp pffft_amd/libpffft_hip.so /tmp/libpffft_hip.keep
for rep in $(seq 1 $REPS); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    cp $lib pffft_amd/libpffft_hip.so
    echo "== $v $rep"
    python tools/r6_quick.py "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/libpffft_hip.keep pffft_amd/libpffft_hip.so
