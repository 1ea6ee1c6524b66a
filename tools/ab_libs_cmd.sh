#!/bin/bash
# A/B of two builds of libpffft_hip.so on ONE box with an arbitrary command:   bash tools/ab_libs_cmd.sh libA.so libB.so <reps> '<command>'
set -u
A=$1; B=$2; REPS=$3; CMD=$4
cp pffft_amd/libpffft_hip.so /tmp/libpffft_hip.keep
for rep in $(seq 1 $REPS); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    cp $lib pffft_amd/libpffft_hip.so
    echo "== $v $rep"
    bash -c "$CMD" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/libpffft_hip.keep pffft_amd/libpffft_hip.so
