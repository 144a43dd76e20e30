#!/bin/bash
# rocprofv3 kernel stats of the four side configurations of bench.py (C1 64^3, C2 256x128x128, C4 512x256x256, C5 1024^3), one run each:
#   bash tools/profile_configs.sh TAG      -> gpurun_out/TAG_{c1,c2,c4,c5}_stats + TAG_cX.json (the configs object of that run)
TAG=${1:-r04}
export TMPDIR=/tmp
O=$PWD/gpurun_out
for c in c1 c2 c4 c5; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_${c}_stats -- python3 bench.py --skip-headline --configs $c > $O/${TAG}_${c}.json 2> $O/${TAG}_${c}.log
  find $O/${TAG}_${c}_stats -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
done
du -sh $O/${TAG}_c* | tail -8
