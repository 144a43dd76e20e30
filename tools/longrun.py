#!/usr/bin/env python3
"""Sanity run: many steps of the bench case at a reduced size; prints divergence, bulk velocity, dpdl and max visct."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ng", type=int, nargs=3, default=[128, 128, 64])
    ap.add_argument("--sgs", default="dsmag")
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    from cales_amd.hotpath import HotPath, initflow
    case = bench.channel_case(a.ng, a.sgs)
    h = HotPath(case); h.upload(*initflow(case)); h.startup()
    for it in range(1, a.steps + 1):
        dt = 0.5 * h.chkdt() if it % 10 == 1 else dt
        h.step(dt)
        if it % 50 == 0 or it == 1:
            dv = h.chkdiv(); vis = h.get("visct")[1:-1, 1:-1, 1:-1]
            print(f"step {it:5d} dt {dt:.3e} divmax {dv[1]:.2e} ubulk {h.bulk_mean('u', 'f'):.12f} dpdl {h.dpdl()} visct max {vis.max():.3e} min {vis.min():.3e} finite {np.isfinite(vis).all()}")
    h.close()


if __name__ == "__main__":
    main()
