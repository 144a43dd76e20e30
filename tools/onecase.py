import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from tests.util import load_golden, relerr
from cales_amd.hotpath import HotPath, initflow
from oracle.oracle import Oracle
name, ng = sys.argv[1], tuple(int(x) for x in sys.argv[2:5])
g, case = load_golden(name); case.ng[:] = ng
rng = np.random.RandomState(0)
h = HotPath(case); o = Oracle(case, nthreads=8)
u, v, w, p = initflow(case)
for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
h.upload(u, v, w, p); h.startup()
visct, pp = o.zeros(), o.zeros()
o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
dt = 0.5 * o.chkdt(visct, u, v, w)
for s in range(2):
    h.step(dt); o.step(dt, u, v, w, p, pp, visct)
    gu, gv, gw, gp, gvis = h.download()
    print("   finite gpu/oracle:", [bool(np.isfinite(a).all()) for a in (gu, gv, gw, gvis)], [bool(np.isfinite(a).all()) for a in (u, v, w, visct)])
    print(os.environ.get("TAG", ""), "step", s, " ".join("%.1e" % relerr(a, b) for a, b in ((gu, u), (gv, v), (gw, w), (gvis, visct))), flush=True)
h.close()
