export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02g_c2_stats -- python3 tools/opbench.py --golden chan_smag_wm --ng 256 128 128 --ops step --reps 20 --noprof > $O/r02g_c2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02g_c1_stats -- python3 tools/opbench.py --golden tgv_ppp --ng 64 64 64 --ops step --reps 50 --noprof > $O/r02g_c1.log 2>&1
find $O/r02g_c* -type f ! -name '*.csv' ! -name '*.log' -delete 2>/dev/null
grep "ms/call" $O/r02g_c2.log $O/r02g_c1.log
