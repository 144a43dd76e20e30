#!/usr/bin/env python3
"""Seconds per operator of the CPU oracle (oracle/cales_oracle.c) for several OpenMP team sizes on this host -- where the timed CPU baseline of
bench.py spends its step and what stops scaling.   python tools/oracle_prof.py 256 256 128 32 64 128"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle.oracle import Oracle

ng = tuple(int(x) for x in sys.argv[1:4]); teams = [int(x) for x in sys.argv[4:]] or [8]
case = bench.channel_case(ng, "dsmag")
for nth in teams:
    o = Oracle(case, nthreads=nth, team_sums=True)
    u, v, w, p = o.initflow(case.inivel, case.is_wallturb); visct, pp = o.zeros(), o.zeros()
    o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
    dt = 0.5 * o.chkdt(visct, u, v, w)
    o.step(dt, u, v, w, p, pp, visct)

    def T(f, n=2):
        t = time.perf_counter()
        for _ in range(n):
            f()
        return (time.perf_counter() - t) / n
    r = {"step": T(lambda: o.step(dt, u, v, w, p, pp, visct)), "cmpt_sgs": T(lambda: o.cmpt_sgs(u, v, w, visct)), "solver": T(lambda: o.solver(pp)),
         "rk": T(lambda: o.rk(1, dt, p, visct, u, v, w)), "bounduvw": T(lambda: o.bounduvw(u, v, w, True, False)), "boundp": T(lambda: o.boundp(p, 0)),
         "fillps": T(lambda: o.fillps(1 / dt, u, v, w, pp)), "correc": T(lambda: o.correc(dt, pp, u, v, w)), "updatep": T(lambda: o.updatep(0., pp, p)),
         "bulk_mean": T(lambda: o.bulk_mean(u, "f")), "chkdt": T(lambda: o.chkdt(visct, u, v, w))}
    print(nth, " ".join(f"{k}={v:.4f}" for k, v in r.items()), flush=True)
    o.close()
