#!/usr/bin/env python3
"""Full-size sanity runs of BASELINE.json configs[3] and [4] shapes on one GPU: a few steps, divergence after projection."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import load_golden
from cales_amd.hotpath import HotPath, initflow

for name, ng, nsteps in (("duct_smag_wm_imp1d", (512, 256, 256), 3), ("cavity_nnn", (1024, 1024, 512), 2)):
    g, case = load_golden(name); case.ng[:] = ng
    t0 = time.time()
    h = HotPath(case); u, v, w, p = initflow(case)
    h.upload(u, v, w, p); del u, v, w, p
    h.startup(); dt = 0.5 * h.chkdt()
    h.step(dt); h.sync(); t1 = time.time()
    for _ in range(nsteps):
        h.step(dt)
    h.sync(); t2 = time.time()
    print(name, ng, "dt %.3e" % dt, "ms/step %.2f" % (1e3 * (t2 - t1) / nsteps), "div", h.chkdiv(), "setup s %.1f" % (t1 - t0), flush=True)
    h.close()
