#!/usr/bin/env python3
"""ms/step with the projection folded into the momentum pass and with the separate pass: DNS channels 512 x 256 x 256, explicit and z-implicit diffusion (development aid)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from tests.test_gpu_golden import _nosgs_case
from cales_amd.hotpath import HotPath, initflow
for name, ng in (("chan_nosgs_imp1d", (512, 256, 256)), ("chan_nosgs", (512, 256, 256))):
    for mode in ("fold", "separate"):
        if mode == "separate": os.environ["CALES_UNFOLDED_MOM"] = "1"
        else: os.environ.pop("CALES_UNFOLDED_MOM", None)
        case = _nosgs_case(name, ng)
        h = HotPath(case); h.upload(*initflow(case)); h.startup(); dt = 0.5 * h.chkdt()
        for _ in range(3): h.step(dt)
        h.sync(); t0 = time.perf_counter()
        for _ in range(20): h.step(dt)
        h.sync(); t = (time.perf_counter() - t0) / 20
        print(name, ng, mode, f"{1e3 * t:.3f} ms/step", flush=True)
        h.close()
