#!/usr/bin/env python3
"""Folded projection (k_corr_strain_tile) against the separate correction pass on the same device, field by field after n steps: the two forms do the
same operations in the same order, so anything above round-off of the plane sums is a fault.   python tools/folddbg.py chan_dsmag 64 20 12 [nsteps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import load_golden
from cales_amd.hotpath import HotPath, initflow

name = sys.argv[1]; ng = tuple(int(x) for x in sys.argv[2:5]); nsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
out = {}
for mode in ("fold", "unfolded"):
    if mode == "unfolded": os.environ["CALES_UNFOLDED_CORREC"] = "1"
    else: os.environ.pop("CALES_UNFOLDED_CORREC", None)
    g, case = load_golden(name); case.ng[:] = ng
    h = HotPath(case); u, v, w, p = initflow(case)
    rng = np.random.RandomState(1)
    for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
    h.upload(u, v, w, p); h.startup(); dt = 0.5 * h.chkdt()
    for _ in range(nsteps): h.step(dt)
    out[mode] = h.download() + [h.get("pp")]; h.close()
for nm, a, b in zip(("u", "v", "w", "p", "visct", "pp"), out["fold"], out["unfolded"]):
    d = np.abs(a - b); sc = max(np.abs(b).max(), 1e-300)
    idx = np.unravel_index(d.argmax(), d.shape)
    din = np.abs(a[1:-1, 1:-1, 1:-1] - b[1:-1, 1:-1, 1:-1]).max() / sc
    print(f"{nm}: max rel diff {d.max() / sc:.3e} at {idx} (interior only {din:.3e})")
