# the five BASELINE.json configurations on one GPU (ms/step, no per-kernel events)
python tools/opbench.py --golden tgv_ppp --ng 64 64 64 --ops step --reps 50 --noprof 2>&1 | tail -1
python tools/opbench.py --golden chan_smag_wm --ng 256 128 128 --ops step --reps 20 --noprof 2>&1 | tail -1
python tools/opbench.py --ops step --reps 5 --noprof 2>&1 | tail -1
python tools/opbench.py --golden duct_smag_wm_imp1d --ng 512 256 256 --ops step --reps 5 --noprof 2>&1 | tail -1
python tools/cavity1024.py 2>&1 | tail -3
