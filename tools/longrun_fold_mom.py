#!/usr/bin/env python3
"""150 steps with the projection folded into the momentum pass (lazy last projection forced) against the separate correction pass: Taylor-Green, cavity, forced channel without subgrid model (development aid)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.test_gpu_golden import _nosgs_case
from cales_amd.hotpath import HotPath, initflow
os.environ["CALES_LAZY_PROJECTION"] = "1"
for name, ng, nsteps in (("tgv_ppp", (64, 64, 64), 150), ("cavity_nnn", (64, 48, 40), 150), ("chan_nosgs", (64, 32, 48), 150)):
    out = {}
    for mode in ("fold", "separate"):
        if mode == "separate": os.environ["CALES_UNFOLDED_MOM"] = "1"
        else: os.environ.pop("CALES_UNFOLDED_MOM", None)
        case = _nosgs_case(name, ng)
        h = HotPath(case); h.upload(*initflow(case)); h.startup(); dt = 0.5 * h.chkdt()
        for i in range(nsteps):
            h.step(dt)
            if (i + 1) % 50 == 0: h.chkdt(); h.chkdiv()
        out[mode] = h.download() + [h.chkdiv()[1]]
        h.close()
    errs = [np.abs(a - b).max() / max(np.abs(b).max(), 1e-300) for a, b in zip(out["fold"][:4], out["separate"][:4])]
    print(name, ng, nsteps, "rel diff u,v,w,p:", " ".join(f"{e:.1e}" for e in errs), "divmax", out["fold"][5], out["separate"][5])
