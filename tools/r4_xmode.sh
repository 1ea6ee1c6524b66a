# XCD-aware tile order on the register-tiled tile passes: PFFFT_HIP_TILE_XMODE 0 / 1 / 2 / 3 on power-of-two and odd-stage sizes (short scan runs)
S=${1:-32768,65536,262144,1048576,10368,14400,17280,36864,61440,115200}
for x in 0 1 2 3; do
  echo "=== XMODE $x f32"; PFFFT_HIP_TILE_XMODE=$x timeout 600 python tools/size_scan.py sizes $S f32 2>&1 | grep "cplx"
done
for x in 0 3; do
  echo "=== XMODE $x f64"; PFFFT_HIP_TILE_XMODE=$x timeout 600 python tools/size_scan.py sizes 32768,262144,1048576,9216,36864 f64 2>&1 | grep "cplx"
done
