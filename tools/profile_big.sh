# kernel stats (+ HBM and SQ counters for C4) of BASELINE.json configs[3] and [4] on one GPU
export TMPDIR=/tmp
O=$PWD/gpurun_out
bash tools/sq_op.sh r02d_c4 --golden duct_smag_wm_imp1d --ng 512 256 256 --ops step --reps 3
rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02d_c5_stats -- python3 tools/cavity1024.py > $O/r02d_c5.log 2>&1
find $O/r02d_c5_stats -type f ! -name '*.csv' -delete 2>/dev/null
tail -3 $O/r02d_c5.log
