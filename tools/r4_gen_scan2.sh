# the sizes whose plan changed with the run-time tile lengths although they had a plan: new against old (PFFFT_HIP_TILE_GENCOST=0)
F=144000,155520,186624,256000,259200,288000,307200,311040,331776,345600,373248,409600,414720,442368,460800,497664,512000,518400,552960,614400,622080,663552,691200,746496,1119744,1280000
D=259200,288000,311040,345600,373248,409600,414720,460800,497664,512000,518400,552960,614400,622080,663552,691200,746496
echo "=== f32 new"; timeout 600 python tools/size_scan.py sizes $F f32 2>&1 | grep "cplx"

echo "=== f64 new"; timeout 600 python tools/size_scan.py sizes $D f64 2>&1 | grep "cplx"

