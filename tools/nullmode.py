#!/usr/bin/env python3
"""Singular (triply periodic) pressure problem at sizes whose n3 is not a power of two: device (default: p(n3) = 0 for the null mode; CALES_KEEP_NULL_MODE=1:
the reference's +eps pivots) against the oracle, per step and at the operator level (development aid behind tests/test_gpu_vs_oracle.py::test_triperiodic_*)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import load_golden, relerr
from cales_amd.hotpath import HotPath, initflow
from oracle.oracle import Oracle

def resid(case, o, p, rhs):
    ng = tuple(int(x) for x in case.ng)
    gr = o.grid(); dzf = gr["dzf"][1:-1]; dzc = gr["dzc"]
    dxi, dyi = ng[0] / case.l[0], ng[1] / case.l[1]
    c = p[1:-1, 1:-1, 1:-1]
    lap = ((p[2:, 1:-1, 1:-1] - 2 * c + p[:-2, 1:-1, 1:-1]) * dxi ** 2 + (p[1:-1, 2:, 1:-1] - 2 * c + p[1:-1, :-2, 1:-1]) * dyi ** 2 +
           ((p[1:-1, 1:-1, 2:] - c) / dzc[1:-1] - (c - p[1:-1, 1:-1, :-2]) / dzc[:-2]) / dzf)
    return np.abs(lap - rhs[1:-1, 1:-1, 1:-1]).max()

for ng in ((10, 6, 12), (46, 74, 15), (16, 16, 16)):
    for keep in (0, 1):
        if keep: os.environ["CALES_KEEP_NULL_MODE"] = "1"
        else: os.environ.pop("CALES_KEEP_NULL_MODE", None)
        g, case = load_golden("tgv_ppp"); case.ng[:] = ng
        rng = np.random.RandomState(7)
        o = Oracle(case, nthreads=8); h = HotPath(case)
        # operator level, dyadic right-hand side with zero sum: the x/y sums of the DC column are exact in any order
        rhs = o.zeros(); r = rng.randint(-2 ** 12, 2 ** 12, size=ng).astype(float)
        r[0, 0, :] -= r.sum(axis=(0, 1))
        rhs[1:-1, 1:-1, 1:-1] = r / 1024.
        ref = rhs.copy(order="F"); o.solver(ref); o.boundp(ref, 0)
        h.set("pp", rhs); h.solver(); h.boundp("pp", 0); got = h.get("pp")
        a, b = got[1:-1, 1:-1, 1:-1], ref[1:-1, 1:-1, 1:-1]
        print(f"{ng} keep={keep} solver: rel diff {np.abs(a - b).max() / np.abs(b).max():.2e}  after removing means {np.abs((a - a.mean()) - (b - b.mean())).max() / np.abs(b - b.mean()).max():.2e}"
              f"  mean dev {a.mean():.6e} orc {b.mean():.6e}  resid dev {resid(case, o, got, rhs):.2e} orc {resid(case, o, ref, rhs):.2e}", flush=True)
        # two steps
        u, v, w, p = initflow(case)
        for x in (u, v, w): x[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
        h.upload(u, v, w, p); h.startup()
        visct, pp = o.zeros(), o.zeros()
        o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
        dt = 0.5 * o.chkdt(visct, u, v, w)
        for _ in range(2):
            h.step(dt); o.step(dt, u, v, w, p, pp, visct)
        gu, gv, gw, gp, gvis = h.download()
        print(f"{ng} keep={keep} 2 steps: u,v,w {relerr(gu, u):.1e} {relerr(gv, v):.1e} {relerr(gw, w):.1e}  p mean dev {gp[1:-1,1:-1,1:-1].mean():.3e} orc {p[1:-1,1:-1,1:-1].mean():.3e}"
              f"  div dev {h.chkdiv()[1]:.1e} orc {o.chkdiv(u, v, w)[1]:.1e}", flush=True)
        h.close()
