#!/usr/bin/env python3
"""Development aid: time steps of one case with and without the x-ghost skipping of cales_step, step by step against the oracle.
   python tools/xskipdbg.py chan_dsmag 32 16 16 5 [ENV=VAL ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import load_golden, relerr
name, ng, ns = sys.argv[1], tuple(int(x) for x in sys.argv[2:5]), int(sys.argv[5])
for kv in sys.argv[6:]:
    k, v = kv.split("="); os.environ[k] = v
from cales_amd.hotpath import HotPath, initflow
from oracle.oracle import Oracle
g, case = load_golden(name); case.ng[:] = ng
o = Oracle(case, nthreads=8)
u, v, w, p = initflow(case)
rng = np.random.RandomState(1)
for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
os.environ.pop("CALES_XGHOSTS_IN_STEP", None)
ha = HotPath(case)
os.environ["CALES_XGHOSTS_IN_STEP"] = "1"
hb = HotPath(case)
for h in (ha, hb): h.upload(u, v, w, p); h.startup()
visct, pp = o.zeros(), o.zeros()
o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
dt = 0.5 * o.chkdt(visct, u, v, w)
I = (slice(1, -1),) * 3
for s in range(ns):
    ha.step(dt); hb.step(dt); o.step(dt, u, v, w, p, pp, visct)
    for tag, h in (("skip", ha), ("keep", hb)):
        gu, gv, gw, gp, gvis = h.download()
        e = [relerr(a[I], b[I]) for a, b in ((gu, u), (gv, v), (gw, w), (gvis, visct))]
        eg = [relerr(a, b) for a, b in ((gu, u), (gv, v), (gw, w), (gvis, visct))]
        d = np.abs(gvis[I] - visct[I]); loc = np.unravel_index(d.argmax(), d.shape)
        dw = np.abs(gw[I] - w[I]); locw = np.unravel_index(dw.argmax(), dw.shape)
        bad = np.argwhere(np.abs(gw - w) > 1e-12 * np.abs(w).max())
        if len(bad): print("   w mismatches:", len(bad), "i in", sorted(set(bad[:, 0].tolist())), "j in", sorted(set(bad[:, 1].tolist()))[:6], "k in", sorted(set(bad[:, 2].tolist())), "first", bad[0], gw[tuple(bad[0])], w[tuple(bad[0])])
        print(s, tag, "interior u,v,w,visct", " ".join("%.1e" % x for x in e), "| with ghosts", " ".join("%.1e" % x for x in eg), "| worst visct at", loc, "w at", locw)
