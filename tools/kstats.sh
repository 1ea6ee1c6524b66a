#!/bin/bash
# usage (on the GPU box through gpurun): tools/kstats.sh <outdir-under-gpurun_out> <python script + args...>
# rocprofv3 kernel trace + stats of the command; prints name / calls / average ns per kernel.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $ROOT/tools && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- "$@" > $OUT/run.log 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
