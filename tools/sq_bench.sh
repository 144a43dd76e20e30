#!/bin/bash
# kernel stats + two SQ counter passes of a short bench run:  bash tools/sq_bench.sh TAG [bench args]   (results in gpurun_out/TAG_*)
TAG=$1; shift
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu --configs none"}
export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_stats -- python3 bench.py $ARGS > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_stats.log
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/${TAG}_sq1 -- python3 bench.py $ARGS > /dev/null 2> $O/${TAG}_sq1.log
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/${TAG}_sq2 -- python3 bench.py $ARGS > /dev/null 2> $O/${TAG}_sq2.log
find $O/${TAG}_* -type f ! -name '*.csv' ! -name '*.json' ! -name '*.log' -delete 2>/dev/null
