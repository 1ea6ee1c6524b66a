#!/bin/bash
# PMC + trace of tools/bench_configs.py configs (development)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_configs.py configs"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/prof_cfg -o cfg -- $CMD > $OUT/prof_cfg.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/prof_cfg2 -o cfg -- $CMD > $OUT/prof_cfg2.log 2>&1
tail -3 $OUT/prof_cfg.log
