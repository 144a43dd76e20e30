#!/bin/bash
# rocprofv3 kernel stats AND HBM counters (--pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the side configurations of bench.py (C1 64^3, C2 256x128x128,
# C4 512x256x256, C5 1024^3), one run each:   bash tools/profile_configs.sh TAG [configs]   -> gpurun_out/TAG_{c1,c2,c4,c5}_{stats,fetch,write} + TAG_cX.json
TAG=${1:-r05}; CFGS=${2:-"c1 c2 c4 c5"}
export TMPDIR=/tmp
O=$PWD/gpurun_out
for c in $CFGS; do
  rm -rf $O/${TAG}_${c}_stats $O/${TAG}_${c}_fetch $O/${TAG}_${c}_write      # (a tag used before: its old files would be summed with the new ones)
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_${c}_stats -- python3 bench.py --skip-headline --configs $c --no-cpu > $O/${TAG}_${c}.json 2> $O/${TAG}_${c}.log
  if [ $c != c1 ]; then
    rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_${c}_fetch -- python3 bench.py --skip-headline --configs $c --no-cpu > /dev/null 2> $O/${TAG}_${c}_fetch.log
    rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_${c}_write -- python3 bench.py --skip-headline --configs $c --no-cpu > /dev/null 2> $O/${TAG}_${c}_write.log
  fi
  find $O/${TAG}_${c}_stats -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
  find $O/${TAG}_${c}_fetch $O/${TAG}_${c}_write -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
done
du -sh $O/${TAG}_c* | tail -16
