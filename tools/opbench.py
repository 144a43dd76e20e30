#!/usr/bin/env python3
"""Times single operators of the hot path at a given size (development aid; also handy under rocprofv3).
  python tools/opbench.py --ng 512 512 512 --sgs dsmag --ops cmpt_sgs solver mom --reps 5"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ng", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--sgs", default="dsmag")
    ap.add_argument("--ops", nargs="+", default=["cmpt_sgs"])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--noprof", action="store_true", help="no per-kernel events (what a production step looks like)")
    ap.add_argument("--golden", default=None, help="take the case (BCs, forcing, sgs, impdiff) from tests/golden/<name>.npz instead of the bench channel")
    a = ap.parse_args()
    import bench
    from cales_amd.hotpath import HotPath, initflow
    if a.golden:
        from tests.util import load_golden
        _, case = load_golden(a.golden); case.ng[:] = a.ng
    else:
        case = bench.channel_case(a.ng, a.sgs)
    h = HotPath(case)
    h.upload(*initflow(case)); h.startup()
    dt = 0.5 * h.chkdt()
    h.step(dt)
    fns = {"cmpt_sgs": h.cmpt_sgs, "solver": h.solver, "mom": h.mom, "step": lambda: h.step(dt), "rk": lambda: h.rk(1, dt),
           "correc": lambda: h.correc(dt), "fillps": lambda: h.fillps(1. / dt), "bounduvw": lambda: h.bounduvw(True, False)}
    for op in a.ops:
        fns[op](); h.sync()
        h.profile_reset(); h.profile(not a.noprof)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fns[op]()
        h.sync(); t = (time.perf_counter() - t0) / a.reps
        h.profile(False)
        st = h.profile_stats()
        print(f"{op}: {1e3 * t:.3f} ms/call  " + "  ".join(f"{k}={v[1] / a.reps:.3f}" for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:24]))
    h.close()


if __name__ == "__main__":
    main()
