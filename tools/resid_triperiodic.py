#!/usr/bin/env python3
"""Triply periodic Poisson solve at a size whose n3 is not a power of two: residual of the device solution and of the reference algorithm
(oracle) against the discrete Laplacian -- documents that their difference is the round-off-defined null mode of the reference."""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from tests.util import load_golden
from cales_amd.hotpath import HotPath
from oracle.oracle import Oracle
g, case = load_golden("tgv_ppp"); ng = (56, 70, 26); case.ng[:] = ng
o = Oracle(case, nthreads=8); h = HotPath(case)
rng = np.random.RandomState(0)
rhs = o.zeros(); rhs[1:-1, 1:-1, 1:-1] = rng.rand(*ng) - 0.5
gr = o.grid(); dzf = gr["dzf"][1:-1]; dzc = gr["dzc"]
rhs[1:-1, 1:-1, 1:-1] -= (rhs[1:-1, 1:-1, 1:-1] * dzf).sum() / (dzf.sum() * ng[0] * ng[1])
ref = rhs.copy(order="F"); o.solver(ref); o.boundp(ref, 0)
h.set("pp", rhs); h.solver(); h.boundp("pp", 0); got = h.get("pp")
dxi, dyi = ng[0] / case.l[0], ng[1] / case.l[1]
def resid(p):
    c = p[1:-1, 1:-1, 1:-1]
    lap = ((p[2:, 1:-1, 1:-1] - 2 * c + p[:-2, 1:-1, 1:-1]) * dxi ** 2 + (p[1:-1, 2:, 1:-1] - 2 * c + p[1:-1, :-2, 1:-1]) * dyi ** 2 +
           ((p[1:-1, 1:-1, 2:] - c) / dzc[1:-1] - (c - p[1:-1, 1:-1, :-2]) / dzc[:-2]) / dzf)
    return np.abs(lap - rhs[1:-1, 1:-1, 1:-1]).max()
print("residual of L_h p - r: device %.2e, oracle (reference algorithm) %.2e; mean(p): device %.2e, oracle %.2e" % (resid(got), resid(ref), got[1:-1,1:-1,1:-1].mean(), ref[1:-1,1:-1,1:-1].mean()))
