#!/usr/bin/env python3
"""One RK substep of a triply periodic case, operator by operator, device against oracle on the oracle's inputs (development aid):
   python tools/stage_ppp.py n1 n2 n3 [keep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ng = tuple(int(x) for x in sys.argv[1:4])
if len(sys.argv) > 4 and sys.argv[4] == "1": os.environ["CALES_KEEP_NULL_MODE"] = "1"
from tests.util import load_golden, relerr, RK, F
from cales_amd.hotpath import HotPath, initflow
from oracle.oracle import Oracle
g, case = load_golden("tgv_ppp"); case.ng[:] = ng
rng = np.random.RandomState(7)
h = HotPath(case); o = Oracle(case, nthreads=8)
u, v, w, p = initflow(case)
for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
visct, pp = o.zeros(), o.zeros()
o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
dt = 0.5 * o.chkdt(visct, u, v, w)
h.upload(u, v, w, p); h.set("visct", visct)
for irk in (1, 2, 3):
    dtrk = float(sum(RK[irk - 1])) * dt
    h.rk(irk, dt); o.rk(irk, dt, p, visct, u, v, w)
    print(irk, "rk", " ".join("%.1e" % relerr(h.get(k), a) for k, a in zip("uvw", (u, v, w))))
    for k, a in zip("uvw", (u, v, w)): h.set(k, a)
    h.bounduvw(False, False); o.bounduvw(u, v, w, False, False)
    h.fillps(1. / dtrk); o.fillps(1. / dtrk, u, v, w, pp)
    print(irk, "fillps %.1e" % relerr(h.get("pp")[1:-1, 1:-1, 1:-1], pp[1:-1, 1:-1, 1:-1]))
    h.set("pp", pp)
    h.solver(); o.solver(pp)
    a = h.get("pp")[1:-1, 1:-1, 1:-1]; b = pp[1:-1, 1:-1, 1:-1]
    print(irk, "solver: |pp|max %.3e  diff/|pp|max %.1e  mean-removed diff / mean-removed max %.1e  means %.6e %.6e" % (np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max(),
          np.abs((a - a.mean()) - (b - b.mean())).max() / np.abs(b - b.mean()).max(), a.mean(), b.mean()))
    # z profile of the plane means (the null column after the inverse transforms)
    za, zb = a.mean(axis=(0, 1)), b.mean(axis=(0, 1))
    print(irk, " plane-mean profile (minus its mean): max diff %.2e, range oracle %.2e device %.2e" % (np.abs((za - za.mean()) - (zb - zb.mean())).max(), np.ptp(zb), np.ptp(za)))
    h.set("pp", pp)
    h.boundp("pp", 0); o.boundp(pp, 0)
    h.correc(dtrk); o.correc(dtrk, pp, u, v, w)
    print(irk, "correc", " ".join("%.1e" % relerr(h.get(k), a_) for k, a_ in zip("uvw", (u, v, w))))
    for k, a_ in zip("uvw", (u, v, w)): h.set(k, a_)
    h.bounduvw(False, True); o.bounduvw(u, v, w, False, True)
    h.updatep(0.); o.updatep(0., pp, p); h.boundp("p", 0); o.boundp(p, 0)
    h.set("p", p)
h.close()
