#!/bin/bash
# Run on the GPU box (through gpurun):   bash tools/profile_round.sh <round>      e.g. 06
# kernel trace of the default bench line + PMC passes over the dominant kernel of every BASELINE config (tools/prof_configs.py).
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, no trace domains besides --kernel-trace next to --pmc (MI355X_MICROARCH.md §HBM, gpurun
# rules).  Then, here:  python tools/pmc_round.py <round>  condenses gpurun_out/prof_r<round> into profiles/r<round>_* and
# profiles/pmc_traffic.json (source hash + capture time: bench.py replays `traffic` only from a capture of the SAME sources, < 24 h old).
set -u
R=${1:?round, e.g. 06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o trace -- python $ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 1 > $OUT/bench.log 2> $OUT/bench.err
CFG="python $ROOT/tools/prof_configs.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg -o trace -- $CFG > $OUT/cfg.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CFG > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CFG > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -o sq -- $CFG > $OUT/sq.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*.csv" | head -30
grep '^{' $OUT/bench.log | cut -c1-400
