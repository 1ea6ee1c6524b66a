#!/bin/bash
# Row f-4 evidence (run through gpurun): kernel trace + HBM byte counters (separate passes) of tools/mix_bench.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/mix_bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mix -o mix -- $CMD > $OUT/prof_mix.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_mix_fetch -o p -- $CMD > $OUT/prof_mix_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_mix_write -o p -- $CMD > $OUT/prof_mix_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/prof_mix_sq -o p -- $CMD > $OUT/prof_mix_sq.log 2>&1
grep -v "^/opt" $OUT/prof_mix.log | tail -8
