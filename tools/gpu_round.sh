#!/bin/bash
# One gpurun call: new-row tests first, then measurements, then the whole GPU suite.  Logs under gpurun_out/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_pfdsp.py -m gpu -x -q > $OUT/t_pfdsp.log 2>&1; echo "pfdsp tests rc=$?"; tail -15 $OUT/t_pfdsp.log
timeout 200 python tools/mix_bench.py > $OUT/mix_bench.log 2>&1; echo "mix_bench rc=$?"; cat $OUT/mix_bench.log | tail -12
PFDSP_HIP_STATIC=1 timeout 200 python tools/mix_bench.py mixonly > $OUT/mix_bench_static.log 2>&1; echo "static:"; head -5 $OUT/mix_bench_static.log
timeout 300 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -3 $OUT/bench.log
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mix -o mix -- python $ROOT/tools/mix_bench.py > $OUT/prof_mix.log 2>&1; echo "prof rc=$?")
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -5 $OUT/t_all.log
