#!/usr/bin/env python3
"""Stage-by-stage comparison GPU vs oracle for one case/size (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import load_golden, relerr, RK, F
from cales_amd.hotpath import HotPath, initflow
from oracle.oracle import Oracle
name, ng = sys.argv[1], tuple(int(x) for x in sys.argv[2:5])
g, case = load_golden(name); case.ng[:] = ng
rng = np.random.RandomState(0)
h = HotPath(case); o = Oracle(case, nthreads=8)
u, v, w, p = initflow(case)
for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
h.upload(u, v, w, p)
h.bounduvw(True, False); h.boundp("p", 0)
o.bounduvw(u, v, w, True, False); o.boundp(p, 0)
print("bounduvw", [("%.1e" % relerr(h.get(k), a)) for k, a in zip("uvw", (u, v, w))])
for iv in (1, 2, 3):
    print(" bc planes", iv, [("%.1e" % (np.abs(a - b).max() / max(1e-30, np.abs(b).max()))) for a, b in zip(h.bcvel_planes(iv), o.bcvel_planes(iv))])
visct = o.zeros(); o.cmpt_sgs(u, v, w, visct); h.cmpt_sgs()
print("cmpt_sgs", "%.1e" % relerr(h.get("visct")[1:-1, 1:-1, 1:-1], visct[1:-1, 1:-1, 1:-1]))
print("index_wm oracle", o.index_wm().ravel(order="F"))
h.close()
h = HotPath(case)
rhs = o.zeros(); rhs[1:-1, 1:-1, 1:-1] = rng.rand(*ng) - 0.5
dzf = o.grid()["dzf"][1:-1]
rhs[1:-1, 1:-1, 1:-1] -= (rhs[1:-1, 1:-1, 1:-1] * dzf).sum() / (dzf.sum() * ng[0] * ng[1])
ref = rhs.copy(order="F"); o.solver(ref)
h.set("pp", rhs); h.solver()
a = h.get("pp")[1:-1, 1:-1, 1:-1]; b = ref[1:-1, 1:-1, 1:-1]
print("solver", "%.1e" % (np.abs((a - a.mean()) - (b - b.mean())).max() / np.abs(b - b.mean()).max()))
# one rk substep
h.upload(u, v, w, p); h.set("visct", visct)
dt = 0.5 * o.chkdt(visct, u, v, w)
f = h.rk(1, dt); h.bulk_forcing()
uo, vo, wo = (F(a) for a in (u, v, w))
fo = o.rk(1, dt, p, visct, uo, vo, wo); o.bulk_forcing(fo, uo, vo, wo)
print("rk+forcing", [("%.1e" % relerr(h.get(k)[1:-1, 1:-1, 1:-1], a[1:-1, 1:-1, 1:-1])) for k, a in zip("uvw", (uo, vo, wo))], f, fo)
h.close()
