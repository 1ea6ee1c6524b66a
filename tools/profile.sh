#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of the headline bench.
# Outputs land in gpurun_out/prof_*; tools/pmc_summary.py turns them into profiles/*.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- $BENCH > $OUT/prof_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- $BENCH > $OUT/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/prof_lds -o lds -- $BENCH > $OUT/prof_lds.log 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/prof_trace.log
