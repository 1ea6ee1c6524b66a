#!/bin/bash
# round 5, GPU call 2: the refactored planner - tests + the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 900 python bench.py --steps 100 --warmup 10 --cpu-seconds 2 > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"
cut -c1-6000 $OUT/bench.out
