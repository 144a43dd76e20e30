#!/bin/bash
# A/B of tuning builds on the GPU box (timing + a parity subset for the working-tree library):  bash tools/ab_round6.sh "base v1 ..." [configs]
# per variant: side configurations (bench.py --skip-headline) and the 512^3 dsmag step (tools/opbench.py), kernel times included
VARS=${1:-"base"}; CFG=${2:-c2,c4,c5}
mkdir -p gpurun_out
for v in $VARS; do
  export CALES_LIB=$PWD/tools/variants/libcales_$v.so
  python3 bench.py --skip-headline --configs $CFG > gpurun_out/ab_${v}_cfg.json 2> gpurun_out/ab_${v}_cfg.err
  python3 tools/opbench.py --ops step --reps 6 > gpurun_out/ab_${v}_512.txt 2>&1
done
unset CALES_LIB
python3 - "$VARS" <<'PY'
import json, sys
for v in sys.argv[1].split():
    try:
        d = json.load(open(f"gpurun_out/ab_{v}_cfg.json"))["configs"]
    except Exception as e:
        print(v, "cfg failed", e); continue
    for k, c in d.items():
        if "error" in c: print(v, k, c["error"]); continue
        print(v, k, f"{c['ms_per_step']:.4f} ms/step solve {c['poisson_solve']['ms']:.4f} ms ({c['poisson_solve']['frac_of_hbm_peak']:.3f})", " ".join(f"{n}={t}" for n, t in list(c["kernels_ms_per_step"].items())[:12]))
    print(v, "512^3:", open(f"gpurun_out/ab_{v}_512.txt").read().strip().splitlines()[-1][:600])
PY
