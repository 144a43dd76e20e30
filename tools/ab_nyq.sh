mkdir -p gpurun_out
for rep in 1 2; do
python3 bench.py --steps 30 --warmup 5 --no-cpu --configs none > gpurun_out/nyq_on_$rep.json 2>gpurun_out/nyq_on_$rep.err
CALES_NO_NYQUIST_PACKING=1 python3 bench.py --steps 30 --warmup 5 --no-cpu --configs none > gpurun_out/nyq_off_$rep.json 2>gpurun_out/nyq_off_$rep.err
done
for P in 8 2; do
python3 tools/slabbench.py --ranks $P --ng 512 512 512 --steps 3 > gpurun_out/nyq_slab${P}_on.txt 2>&1
CALES_NO_NYQUIST_PACKING=1 python3 tools/slabbench.py --ranks $P --ng 512 512 512 --steps 3 > gpurun_out/nyq_slab${P}_off.txt 2>&1
done
python3 bench.py --skip-headline --configs c1,c2,c4 --no-cpu > gpurun_out/nyq_cfg_on.json 2>/dev/null
CALES_NO_NYQUIST_PACKING=1 python3 bench.py --skip-headline --configs c1,c2,c4 --no-cpu > gpurun_out/nyq_cfg_off.json 2>/dev/null
