#!/bin/bash
# round 5, evidence on the final build: GPU tests, default bench line, profiles (rocprofv3 trace + PMC passes), steady scans
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_final
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"
cut -c1-1500 $OUT/bench.out
bash tools/profile_r05.sh > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
bash tools/r5_scans.sh all > $OUT/scans.log 2>&1; tail -8 $OUT/scans.log
