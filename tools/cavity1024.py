#!/usr/bin/env python3
"""BASELINE.json configs[4] shape on ONE GPU: lid-driven cavity 1024^3 (fields of 8.6 GB: 64-bit offsets, tridiagonal tile with
16 planes per lane), two steps, divergence after projection. Needs ~160 GB of HBM and ~45 GB of host memory for the initial fields."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import load_golden
from cales_amd.hotpath import HotPath, initflow

free_gb = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) / 2 ** 20
print("host MemAvailable %.0f GB" % free_gb, flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if free_gb < 6 * 8 * (n + 2) ** 3 / 2 ** 30:
    print("not enough host memory for the initial fields: skipped"); sys.exit(0)
g, case = load_golden("cavity_nnn"); case.ng[:] = (n, n, n)
t0 = time.time()
h = HotPath(case); u, v, w, p = initflow(case)
h.upload(u, v, w, p); del u, v, w, p
h.startup(); dt = 0.5 * h.chkdt()
h.step(dt); h.sync(); t1 = time.time()
for _ in range(2):
    h.step(dt)
h.sync(); t2 = time.time()
print("cavity", case.ng, "dt %.3e" % dt, "ms/step %.2f" % (1e3 * (t2 - t1) / 2), "div", h.chkdiv(), "setup s %.1f" % (t1 - t0), flush=True)
h.profile_reset(); h.profile(True); h.step(dt); h.sync(); h.profile(False)
print("  " + "  ".join(f"{k}={v[1]:.2f}" for k, v in sorted(h.profile_stats().items(), key=lambda kv: -kv[1][1])[:14]))
h.close()
