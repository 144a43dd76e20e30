#!/bin/bash
# SQ / HBM counters of one operator at full size (quicker than a whole bench pass):  bash tools/sq_op.sh TAG "opbench args"
TAG=${1:-op}; shift
ARGS=${@:-"--ops cmpt_sgs --reps 2"}
export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_stats -- python3 tools/opbench.py $ARGS > $O/${TAG}_stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_fetch -- python3 tools/opbench.py $ARGS > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_write -- python3 tools/opbench.py $ARGS > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/${TAG}_sq1 -- python3 tools/opbench.py $ARGS > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/${TAG}_sq2 -- python3 tools/opbench.py $ARGS > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $O/${TAG}_sq3 -- python3 tools/opbench.py $ARGS > /dev/null 2>&1
find $O/${TAG}_* -type f ! -name '*.csv' ! -name '*.log' -delete 2>/dev/null
grep -E "ms/call" $O/${TAG}_stats.log
