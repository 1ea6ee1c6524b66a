# float sizes that leave three streaming sweeps for two run-time tile passes under the wide threshold: new against PFFFT_HIP_TILE_WIDECOST=0
S=19440,32400,34992,58320,60000,64800,69984,90000,97200,108000,116640,21600,36000,38880,100000,200000
echo "=== new"; timeout 600 python tools/size_scan.py sizes $S f32 2>&1 | grep "cplx\|real"
echo "=== old"; PFFFT_HIP_TILE_WIDECOST=0 timeout 600 python tools/size_scan.py sizes $S f32 2>&1 | grep "cplx\|real"
