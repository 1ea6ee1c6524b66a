#!/bin/bash
# usage: profile_cmd.sh <tag> <python script relative to repo> : two PMC passes, results in gpurun_out/prof_<tag>{,2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/$2"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/prof_$1 -o p -- $CMD > $OUT/prof_$1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/prof_$1_2 -o p -- $CMD > $OUT/prof_$1_2.log 2>&1
