#!/bin/bash
# Round 4, run-time tile plans (fft_tileg.h) against what the sizes ran before: tools/size_scan.py's short runs (256 MiB per launch, values
# checked) on a named set of sizes, "new" = the build's planner, "old" = the A/B switch that takes the change away.
#   tools/r4_gen_scan.sh <set> [f32|f64|both]
#   deep     five streaming sweeps before                              old: PFFFT_HIP_TILE_GENCOST=0 (no run-time lengths)
#   changed  sizes that HAD a plan and changed it                      old: PFFFT_HIP_TILE_GENCOST=0
#   left     sizes that left three streaming sweeps (cost < 286)       old: PFFFT_HIP_TILE_GENCOST=0
#   wide     float 2^4 / 2^5 sizes under the wide cost bar             old: PFFFT_HIP_TILE_WIDECOST=0
SET=${1:-deep}; PREC=${2:-both}
OLD="PFFFT_HIP_TILE_GENCOST=0"
case $SET in
  deep) S=104976,209952,291600,314928,384000,450000,500000,524880,600000;;
  changed) S=144000,155520,186624,256000,259200,288000,307200,311040,331776,345600,373248,409600,414720,442368,460800,497664,512000,518400,552960,614400,622080,663552,691200,746496,1119744,1280000;;
  left) S=8000,120000,139968,216000,233280,240000,279936,320000,384000;;
  wide) S=10800,11664,12000,18000,19440,20000,23328,30000,32400,34992,50000,54000,58320,60000,64800,69984,90000,97200,100000,104976,108000,116640,21600,36000,38880,200000; OLD="PFFFT_HIP_TILE_WIDECOST=0";;
  *) echo "unknown set $SET"; exit 1;;
esac
for p in f32 f64; do
  if [ $PREC != both ] && [ $PREC != $p ]; then continue; fi
  echo "=== $p new"; timeout 900 python tools/size_scan.py sizes $S $p 2>&1 | grep "cplx\|real"
  echo "=== $p old"; env $OLD timeout 900 python tools/size_scan.py sizes $S $p 2>&1 | grep "cplx\|real"
done
