#!/usr/bin/env python3
"""Random small grid sizes, two steps, GPU against the oracle (development aid; the committed tests cover fixed sizes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import load_golden, relerr
from cales_amd.hotpath import HotPath, initflow
from oracle.oracle import Oracle

def case_lwm(name):
    return np.zeros(1, bool) if name.startswith("open:") else load_golden(name)[1].lwm != 0


rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
names = ["chan_dsmag", "chan_dsmag_wm", "chan_smag_wm", "tgv_dsmag_ppp", "cavity_nnn", "duct_smag_wm_imp1d", "couette_imp3d_ops", "chan_smag",
         "duct_dsmag_wm", "duct_dsmag", "cavity_dsmag", "duct_smag_wm", "tgv_ppp", "devchan_nd", "halfchan_imp1d",
         "open:0", "open:1", "open:4", "open:5", "open:7", "open:8"]      # 3-D implicit diffusion with open boundaries (tests/test_gpu_vs_oracle.py OPEN_SETS)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    name = names[trial % len(names)]
    ng = tuple(int(2 * rng.randint(2, 40)) for _ in range(2)) + (int(rng.randint(9, 80)),)      # ng(3) may be odd
    if trial % 3 == 2: ng = (int(2 ** rng.randint(4, 8)),) + ng[1:]      # power-of-two rows: the radix-8 x pass, fillps inside it, cales_step without x ghost updates
    if os.environ.get("FUZZ_POW2"):      # power-of-two rows AND y lines (radix-8 passes both ways, the packed modes 0 and n1/2 with periodic x and y), every chunking of the z tile
        ng = (int(2 ** rng.randint(6, 9)), int(2 ** rng.randint(4, 8)), int(rng.choice([rng.randint(9, 80), rng.randint(129, 600)])))
    if np.any(case_lwm(name)): ng = ng[:2] + (max(ng[2], 12),)
    if name.startswith("open:"):
        from tests.test_gpu_vs_oracle import OPEN_SETS, _open_case
        xs, ys, _ = OPEN_SETS[int(name[5:])]
        case = _open_case(xs, ys, ng); case.inivel = "uni"
    else:
        g, case = load_golden(name); case.ng[:] = ng
        if case.sgstype == "none" and case.cbcvel[0, 0, 0] != "P": case.cbcsgs[:, 0] = "D"      # see tests/test_gpu_golden.py
    if np.any(case.lwm != 0):      # a sampling height the reference accepts on this grid (sanity.f90:224-231)
        case.hwm = max(float(case.hwm), 1.6 * max(float(case.l[d]) / ng[d] for d in range(3) if case.lwm[:, d].any()))
    if case.inivel == "hcp": case.inivel = "poi"
    try:
        h = HotPath(case)
    except Exception as e:
        print(name, ng, "create refused:", str(e)[:80]); continue
    o = Oracle(case, nthreads=8)
    u, v, w, p = initflow(case)
    for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
    init0 = tuple(a.copy(order="F") for a in (u, v, w, p))
    h.upload(u, v, w, p); h.startup()
    visct, pp = o.zeros(), o.zeros()
    o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
    dt = 0.5 * o.chkdt(visct, u, v, w)
    for _ in range(2):
        h.step(dt); o.step(dt, u, v, w, p, pp, visct)
    gu, gv, gw, gp, gvis = h.download()
    errs = [relerr(a, b) for a, b in ((gu, u), (gv, v), (gw, w))] + [relerr(gvis, visct)]
    both_blew_up = not np.isfinite(u).all() and not np.isfinite(gu).all()      # unstable case: the reference blows up the same way
    # triply periodic boxes with n3 not a power of two: the yardstick is how far the reference algorithm's OWN result moves when the initial field moves
    # by one unit in the last place (tests/test_oracle_solver.py: 1e-10..1e-7 there, 1e-15 for well-posed boxes)
    illposed = name.startswith("tgv") and (ng[2] & (ng[2] - 1)) != 0
    bound = 1e-9
    if illposed and max(errs[:3]) >= 1e-9:
        sens = 0.
        for t in range(2):
            r2 = np.random.RandomState(100 + t)
            o2 = Oracle(case, nthreads=8)
            u2, v2, w2, p2 = (a.copy(order="F") for a in init0)
            for a in (u2, v2, w2): a *= 1. + np.finfo(float).eps * (r2.randint(0, 2, size=a.shape) * 2 - 1)
            vis2, pp2 = o2.zeros(), o2.zeros()
            o2.bounduvw(u2, v2, w2, True, False); o2.boundp(p2, 0); o2.cmpt_sgs(u2, v2, w2, vis2); o2.boundp(vis2, 1)
            for _ in range(2): o2.step(dt, u2, v2, w2, p2, pp2, vis2)
            sens = max(sens, max(relerr(a, b) for a, b in ((u2, u), (v2, v), (w2, w))))
        from tests.util import triperiodic_bounds
        bnd = triperiodic_bounds(case, (u, v, w), p, dt, 2, sens)
        ok = both_blew_up or (all(e < b for e, b in zip(errs[:3], bnd)) and max(bnd) < 1e-5 and errs[3] < 1e-6)
    else:
        ok = both_blew_up or (max(errs[:3]) < bound and errs[3] < 1e-6)
    bad += not ok
    print("OK " if ok else "BAD", name, ng, " ".join("%.1e" % e for e in errs), "div %.1e / oracle %.1e" % (h.chkdiv()[1], o.chkdiv(u, v, w)[1]), flush=True)
    h.close()
print("bad:", bad)
