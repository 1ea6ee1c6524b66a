# the sizes that left the three streaming passes for a two-pass plan with a run-time tile length: new against old (PFFFT_HIP_TILE_GENCOST=0)
F=120000,139968,216000,233280,240000,279936,320000,384000
D=8000,120000,139968,216000,233280,240000,279936,320000,384000
echo "=== f32 new"; timeout 600 python tools/size_scan.py sizes $F f32 2>&1 | grep "cplx\|real"
echo "=== f32 old"; PFFFT_HIP_TILE_GENCOST=0 timeout 600 python tools/size_scan.py sizes $F f32 2>&1 | grep "cplx\|real"
echo "=== f64 new"; timeout 600 python tools/size_scan.py sizes $D f64 2>&1 | grep "cplx\|real"
echo "=== f64 old"; PFFFT_HIP_TILE_GENCOST=0 timeout 600 python tools/size_scan.py sizes $D f64 2>&1 | grep "cplx\|real"
