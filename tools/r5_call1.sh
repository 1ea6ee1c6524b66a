#!/bin/bash
# round 5, GPU call 1: tests + launch-shape / FIR / stall experiments
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call1
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for once in "4,0" "4,1" "4,4" "4,16" "2,4" "8,4" "8,16"; do
  PFFFT_HIP_C1024_ONCE=$once timeout 300 python tools/r5_c1024_batch.py >> $OUT/c1024.log 2>&1
done
PFFFT_HIP_C1024_ONCE="4,0" PFFFT_HIP_C1024_WGS=2 timeout 300 python tools/r5_c1024_batch.py >> $OUT/c1024.log 2>&1
cat $OUT/c1024.log
for x in 1 0; do PFFASTCONV_HIP_XCD=$x timeout 300 python tools/r5_fir.py >> $OUT/fir.log 2>&1; done
cat $OUT/fir.log
timeout 600 python tools/r5_stall_probe.py > $OUT/stall.log 2>&1
cat $OUT/stall.log
cd /tmp
for x in 1 0; do
  PFFASTCONV_HIP_XCD=$x timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fir_fetch_$x -o f -- python $ROOT/tools/r5_fir.py > $OUT/fir_fetch_$x.log 2>&1
  echo "XCD=$x"; python $ROOT/tools/r5_pmc_avg.py $OUT/fir_fetch_$x fastconv_split
done
rm -rf $OUT/fir_fetch_*/*/*.db 2>/dev/null
du -sh $OUT
