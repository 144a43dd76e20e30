# kernel stats (+ HBM and SQ counters for C4) of BASELINE.json configs[3] and [4] on one GPU (TAG = $1)
TAG=${1:-r03}
export TMPDIR=/tmp
O=$PWD/gpurun_out
bash tools/sq_op.sh ${TAG}_c4 --golden duct_smag_wm_imp1d --ng 512 256 256 --ops step --reps 3
rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_c5_stats -- python3 tools/cavity1024.py > $O/${TAG}_c5.log 2>&1
find $O/${TAG}_c5_stats -type f ! -name '*.csv' -delete 2>/dev/null
tail -3 $O/${TAG}_c5.log
