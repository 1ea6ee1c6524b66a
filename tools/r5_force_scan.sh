#!/bin/bash
# size_scan of explicit sizes under forced tile lengths: tools/r5_force_scan.sh f64 82944 "144,576" [more "N plan" pairs ...]
p=$1; shift
while [ $# -ge 2 ]; do
  N=$1; plan=$2; shift 2
  echo "== N=$N forced $plan"; PFFFT_HIP_TILE_FORCE="$plan" python tools/size_scan.py sizes $N $p steady 2>&1 | grep "cplx\|real\|pffft_hip"
  echo "== N=$N default"; python tools/size_scan.py sizes $N $p steady 2>&1 | grep "cplx\|real"
done
