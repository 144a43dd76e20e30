#!/usr/bin/env python3
"""fold of the projection into the momentum pass against the separate pass: where the fields differ (development aid)
  python tools/folddbg_mom.py tgv_ppp 64 16 24 [ENV=VAL ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
name = sys.argv[1]; ng = tuple(int(x) for x in sys.argv[2:5])
for kv in sys.argv[5:]:
    k, v = kv.split("="); os.environ[k] = v
from tests.test_gpu_golden import _nosgs_case
from cales_amd.hotpath import HotPath, initflow
out = {}
for mode in ("fold", "separate"):
    if mode == "separate":
        os.environ["CALES_UNFOLDED_MOM"] = "1"
    case = _nosgs_case(name, ng)
    h = HotPath(case); u, v, w, p = initflow(case)
    rng = np.random.RandomState(2)
    for a in (u, v, w):
        a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
    h.upload(u, v, w, p); h.startup(); dt = 0.5 * h.chkdt()
    h.step(dt)
    out[mode] = h.download() + [h.get("pp")]
    h.close()
for nm, a, b in zip(("u", "v", "w", "p", "visct", "pp"), out["fold"], out["separate"]):
    d = np.abs(a - b); i = np.unravel_index(d.argmax(), d.shape)
    inner = d[1:-1, 1:-1, 1:-1].max()
    print(f"{nm}: max {d.max():.3e} at {i} (shape {a.shape}), interior max {inner:.3e}, cells > 1e-12: {(d > 1e-12).sum()}")
