#!/usr/bin/env python3
"""Development aid: triply periodic boxes the size fuzzer flagged -- velocity difference to the oracle after two steps against the sensitivity of the
reference algorithm to its round-off-defined pressure constant C: eps |C| dt max(1/dx) / |u|max, default and CALES_KEEP_NULL_MODE=1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import load_golden, relerr
from cales_amd.hotpath import HotPath, initflow
from oracle.oracle import Oracle
sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(20, 58, 12), (74, 52, 26), (76, 46, 19), (10, 6, 12), (46, 74, 15)]
for ng in sizes:
    for keep in (0, 1):
        if keep: os.environ["CALES_KEEP_NULL_MODE"] = "1"
        else: os.environ.pop("CALES_KEEP_NULL_MODE", None)
        g, case = load_golden("tgv_ppp"); case.ng[:] = ng
        rng = np.random.RandomState(3)
        o = Oracle(case, nthreads=8); h = HotPath(case)
        u, v, w, p = initflow(case)
        for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
        h.upload(u, v, w, p); h.startup()
        visct, pp = o.zeros(), o.zeros()
        o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
        dt = 0.5 * o.chkdt(visct, u, v, w)
        Cmax = 0.
        for _ in range(2):
            h.step(dt); o.step(dt, u, v, w, p, pp, visct)
            Cmax = max(Cmax, abs(pp[1:-1, 1:-1, 1:-1].mean()))
        gu, gv, gw, gp, gvis = h.download()
        errs = [relerr(a, b) for a, b in ((gu, u), (gv, v), (gw, w))]
        dxi = max(ng[d] / case.l[d] for d in range(3)); umax = max(np.abs(a).max() for a in (u, v, w))
        unit = np.finfo(float).eps * Cmax * dt * dxi / umax
        print(ng, "keep" if keep else "pin ", "errs", " ".join("%.1e" % e for e in errs), "|C| of pp %.2e  mean p dev %.2e orc %.2e  eps|C|dt/dx/|u| = %.1e  ratio %.1f" %
              (Cmax, gp[1:-1, 1:-1, 1:-1].mean(), p[1:-1, 1:-1, 1:-1].mean(), unit, max(errs) / unit), flush=True)
        h.close()
