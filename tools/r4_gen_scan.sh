# usage: tools/r4_gen_scan.sh [new|old|both]: the sizes without a register-tiled two-pass plan on the run-time tile plans (fft_tileg.h) / on the
# streaming passes (PFFFT_HIP_TILE_GENCOST=0), tools/size_scan.py's short runs
S=${2:-10800,12000,23328,50000,104976,250000,314928,600000}
W=${1:-both}
for p in f32 f64; do
  if [ $W != old ]; then echo "=== $p new"; timeout 300 python tools/size_scan.py sizes $S $p 2>&1 | grep -v "^#\|amdgpu.ids"; fi
  if [ $W != new ]; then echo "=== $p old"; PFFFT_HIP_TILE_GENCOST=0 timeout 300 python tools/size_scan.py sizes $S $p 2>&1 | grep -v "^#\|amdgpu.ids"; fi
done
