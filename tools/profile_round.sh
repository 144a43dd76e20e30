#!/bin/bash
# One profiling round on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh TAG [bench args]
#   1. FETCH_SIZE / WRITE_SIZE calibration on streams of known size (tools/micro/calib.hip), one --pmc pass each
#   2. rocprofv3 --output-format csv --kernel-trace --stats of the bench command
#   3. --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of the same command (separate runs, no tracing besides kernel-trace)
#   4. two SQ passes (8 slots each): issue/wait/LDS cycles and instruction counts
# Everything lands in gpurun_out/TAG_*; profiles/summarize.py condenses it into profiles/TAG_*.
TAG=${1:-r02}; shift
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu --configs none"}      # (the side configurations have tools/profile_configs.sh; with them in the trace the output exceeds what a gpurun call merges back)
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
rm -rf $O/${TAG}_stats $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_sq1 $O/${TAG}_sq2 $O/${TAG}_calib_fetch $O/${TAG}_calib_write      # (a tag used before: its old files would be summed with the new ones)
if [ ! -x tools/micro/calib ]; then (cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o calib calib.hip); fi
./tools/micro/calib > $O/${TAG}_calib.jsonl 2>&1
if [ ! -x tools/micro/segcopy ]; then (cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o segcopy segcopy.hip); fi
./tools/micro/segcopy > $O/${TAG}_segcopy.log 2>&1      # copy rate of the column-tile access pattern of the y transforms
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_calib_fetch -- ./tools/micro/calib > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_calib_write -- ./tools/micro/calib > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_stats -- python3 bench.py $ARGS > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_stats.log
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_fetch -- python3 bench.py $ARGS > /dev/null 2> $O/${TAG}_fetch.log
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_write -- python3 bench.py $ARGS > /dev/null 2> $O/${TAG}_write.log
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/${TAG}_sq1 -- python3 bench.py $ARGS > /dev/null 2> $O/${TAG}_sq1.log
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/${TAG}_sq2 -- python3 bench.py $ARGS > /dev/null 2> $O/${TAG}_sq2.log
# keep only the csv files (the merge back is limited to 64 MiB)
find $O/${TAG}_* -type f ! -name '*.csv' ! -name '*.json' ! -name '*.jsonl' ! -name '*.log' -delete 2>/dev/null
du -sh $O/${TAG}_* | tail -12
