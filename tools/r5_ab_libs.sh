#!/bin/bash
# round 5: A/B of two builds of libpffft_hip.so on ONE box (the box-to-box spread of the pool, +-1.5 % on a family mean, is as large as most effects
# measured this round).  Build the two libraries here (e.g. `git stash; python -m pffft_amd.build; cp pffft_amd/libpffft_hip.so gpurun_exp/libA.so;
# git stash pop; python -m pffft_amd.build; cp ... gpurun_exp/libB.so`), then on the GPU box:
#     gpurun -- 'bash tools/r5_ab_libs.sh gpurun_exp/libA.so gpurun_exp/libB.so 12288 16000 f32 [reps=2] [grep pattern=^real]'
# -> gpurun_out/ab_libs.txt: tools/size_scan.py ... steady of the range under each library, alternating, `reps` times.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
A=$1; B=$2; LO=$3; HI=$4; P=$5; REPS=${6:-2}; PAT=${7:-^real}
OUT=gpurun_out/ab_libs.txt
mkdir -p gpurun_out; : > $OUT
cp pffft_amd/libpffft_hip.so /tmp/libpffft_hip.keep
for rep in $(seq 1 $REPS); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    cp $lib pffft_amd/libpffft_hip.so
    echo "== $v $rep ($lib)" >> $OUT
    timeout 600 python tools/size_scan.py $LO $HI $P 1 steady 2>&1 | grep -E "$PAT" >> $OUT
  done
done
cp /tmp/libpffft_hip.keep pffft_amd/libpffft_hip.so
cat $OUT | cut -c1-80
