#!/bin/bash
# steady scans of every legal size (tools/size_scan.py ... steady), beyond LDS and LDS-resident, both precisions -> gpurun_out/scans/
#     bash tools/scans.sh [all|beyond|resident]     then copy the four files to profiles/r<round>_scan_*.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/scans
mkdir -p $OUT
cd $ROOT
what=${1:-all}
if [ "$what" = all ] || [ "$what" = beyond ]; then
  for p in f32 f64; do
    lo=$([ $p = f32 ] && echo 10240 || echo 6000)
    timeout 1500 python tools/size_scan.py $lo 600000 $p 1 steady > $OUT/scan_beyond_lds_$p.txt 2>&1
    tail -2 $OUT/scan_beyond_lds_$p.txt
  done
fi
if [ "$what" = all ] || [ "$what" = resident ]; then
  for p in f32 f64; do
    timeout 1500 python tools/size_scan.py 16 $([ $p = f32 ] && echo 18432 || echo 9216) $p 1 steady > $OUT/scan_lds_resident_$p.txt 2>&1
    tail -2 $OUT/scan_lds_resident_$p.txt
  done
fi
