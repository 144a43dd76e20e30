# C1 (64^3 Taylor-Green) and C2 (256x128x128 wall-modelled channel) on one GPU: ms/step without per-kernel events + kernel trace (TAG = $1)
TAG=${1:-r03}
export TMPDIR=/tmp
O=$PWD/gpurun_out
python tools/opbench.py --golden tgv_ppp --ng 64 64 64 --ops step --reps 200 --noprof 2>&1 | tail -1
python tools/opbench.py --golden chan_smag_wm --ng 256 128 128 --ops step --reps 50 --noprof 2>&1 | tail -1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_c2_stats -- python3 tools/opbench.py --golden chan_smag_wm --ng 256 128 128 --ops step --reps 20 --noprof > $O/${TAG}_c2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/${TAG}_c1_stats -- python3 tools/opbench.py --golden tgv_ppp --ng 64 64 64 --ops step --reps 50 --noprof > $O/${TAG}_c1.log 2>&1
find $O/${TAG}_c* -type f ! -name '*.csv' ! -name '*.log' -delete 2>/dev/null
grep "ms/call" $O/${TAG}_c2.log $O/${TAG}_c1.log
