import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from tests.util import load_golden, relerr, oracle_steps
from cales_amd.hotpath import HotPath, initflow
name, ng = sys.argv[1], tuple(int(x) for x in sys.argv[2:5])
for kv in sys.argv[5:]:
    k, v = kv.split("="); os.environ[k] = v
g, case = load_golden(name); case.ng[:] = ng
for seed in (3, 4, 5, 6):
    u, v, w, p, dt = oracle_steps(case, 2, seed)
    h = HotPath(case)
    u0, v0, w0, p0 = initflow(case)
    rng = np.random.RandomState(seed)
    for a in (u0, v0, w0): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
    h.upload(u0, v0, w0, p0); h.startup()
    for _ in range(2): h.step(dt)
    gu, gv, gw, gp, gvis = h.download()
    print(name, ng, "seed", seed, " ".join("%.1e" % relerr(a, b) for a, b in ((gu, u), (gv, v), (gw, w))), "mean p dev %.3e orc %.3e" % (gp[1:-1,1:-1,1:-1].mean(), p[1:-1,1:-1,1:-1].mean()), flush=True)
    h.close()
