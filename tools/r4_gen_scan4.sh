# sizes with 2^4 / 2^5 and a large odd part: three streaming sweeps (default) against two run-time tile passes (PFFFT_HIP_TILE_MAXCOST raised)
S=10800,11664,12000,18000,20000,23328,30000,50000,54000,100000,104976
for p in f32 f64; do
  echo "=== $p streaming"; timeout 600 python tools/size_scan.py sizes $S $p 2>&1 | grep "cplx"
  echo "=== $p tiles"; PFFFT_HIP_TILE_MAXCOST=340 timeout 600 python tools/size_scan.py sizes $S $p 2>&1 | grep "cplx"
done
