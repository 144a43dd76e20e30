#!/usr/bin/env python3
"""Random sizes, k chunks, switches and slab counts for the projection folded into the momentum pass (k_momrk<.., CORR>, no subgrid model): cales_step with
the fold against cales_step with the separate correction pass (CALES_UNFOLDED_MOM) on the same device, all fields incl. ghost cells to 1e-12; with P > 1
the emulated slabs (fold on every slab) against the one-rank run with the separate pass.    python tools/fuzz_fold_mom.py SEED NTRIALS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import relerr
from tests.test_gpu_golden import _nosgs_case
from cales_amd.hotpath import HotPath, initflow

POOL = ["CALES_LAZY_PROJECTION", "CALES_LAZY_PROJECTION", "CALES_UNMERGED_BC", "CALES_XGHOSTS_IN_STEP", "CALES_WIDE_OFFSETS", "CALES_UNFUSED_FILLPS", "CALES_UNFUSED_MEAN",
        "CALES_UNFUSED_FORCING", "CALES_KEEP_LAST_RHS", "CALES_FFT_GENERIC", "CALES_GAUSSEL_MARCH"]
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0


def perturbed(case, ng):
    u, v, w, p = initflow(case)
    r2 = np.random.RandomState(1)
    for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (r2.rand(*ng) - 0.5)
    return u, v, w, p


def run(case, ng, nsteps, unfolded):
    if unfolded: os.environ["CALES_UNFOLDED_MOM"] = "1"
    else: os.environ.pop("CALES_UNFOLDED_MOM", None)
    h = HotPath(case)
    h.upload(*perturbed(case, ng)); h.startup(); dt = 0.5 * h.chkdt()
    h.profile(True)
    for _ in range(nsteps): h.step(dt)
    out = h.download() + [h.get("pp")]
    h.profile(False); ncorr = (h.profile_stats().get("correc_updatep", (0, 0.))[0] + h.profile_stats().get("correc", (0, 0.))[0]); h.close()
    return out, dt, ncorr


for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    name = ("tgv_ppp", "cavity_nnn", "chan_nosgs", "halfchan_nosgs", "chan_nosgs_imp1d", "halfchan_imp1d")[trial % 6]
    P = int(rng.choice([1, 1, 1, 2, 3, 4]))
    n2l = int(rng.randint(2, 22)); n2l += (n2l * P) % 2      # (ng(2) even, sanity.f90)
    n1 = int(rng.choice([16, 32, 64, 128, 192])) if rng.rand() < 0.6 else 2 * int(rng.randint(4, 80))
    ng = (n1, n2l * P if P > 1 else 2 * int(rng.randint(2, 24)), int(rng.randint(3, 50)))
    for k in POOL + ["CALES_KCHUNK"]: os.environ.pop(k, None)
    chosen = sorted(set(str(k) for k in rng.choice(POOL, size=rng.randint(0, 5), replace=False)))
    for k in chosen: os.environ[k] = "1"
    if rng.rand() < 0.5: os.environ["CALES_KCHUNK"] = str(int(rng.randint(2, 14)))
    case = _nosgs_case(name, ng)
    nsteps = int(rng.randint(1, 4))
    try:
        ref, dt, nref = run(case, ng, nsteps, True)
        if P == 1:
            got, _, ngot = run(case, ng, nsteps, False)
            errs = [relerr(a, b) for a, b in zip(got, ref)]
            folded = ngot == (1 if 'CALES_LAZY_PROJECTION' in chosen else nsteps) and nref == 3 * nsteps
        else:
            from cales_amd.decomp import run_loopback
            os.environ.pop("CALES_UNFOLDED_MOM", None)

            def body(h, r):
                h.upload_global(*perturbed(case, ng)); h.startup()
                h.profile(True)
                for _ in range(nsteps): h.step(dt)
                out = h.download() + [h.get("pp")]
                h.profile(False)
                return out[:5] + [h.lo, h.n, (h.profile_stats().get("correc_updatep", (0, 0.))[0] + h.profile_stats().get("correc", (0, 0.))[0]), out[5]]
            res = run_loopback(case, P, body)
            folded = all(r_[7] == nsteps for r_ in res)      # (several slabs: the third substep's projection is never left pending, CALES_LAZY_PROJECTION is ignored)
            errs = []
            for q in range(5):
                e = 0.
                for r_ in res:
                    j0 = r_[5][1] - 1; n2 = r_[6][1]
                    e = max(e, relerr(r_[q][:, 1:-1, :], ref[q][:, j0 + 1:j0 + n2 + 1, :]) if q != 3 else 0.)
                errs.append(e)
            # p and pp on the slabs (the p + pp store of k_momrk<CORR = 1>, the P / scr1 swap, the ghost rows riding with the prediction's bounduvw): the
            # difference to the one-rank field over ALL slabs, its mean over the interior removed (singular mode), ghost cells in x and z included
            for q, a in ((3, 3), (8, 5)):
                d = np.concatenate([r_[q][:, 1:-1, :] - ref[a][:, r_[5][1]:r_[5][1] + r_[6][1], :] for r_ in res], axis=1)
                d = d - d[1:-1, :, 1:-1].mean()
                e = np.abs(d).max() / max(np.abs(ref[a] - ref[a][1:-1, 1:-1, 1:-1].mean()).max(), 1e-300)
                if q == 3: errs[3] = e
                else: errs.append(e)
    except Exception as e:
        print(trial, name, ng, "P", P, chosen, "exception:", repr(e)[:200]); bad += 1; continue
    tol = [1e-10 if P > 1 else 1e-12] * len(errs)
    if P > 1 and len(tol) > 5: tol[5] = 1e-9      # pp on slabs: div(u*) / dtrk amplifies the last digits the slab-wise bulk-mean sums move (2e-10 seen on the half channel)
    if len(tol) > 5 and name == "tgv_ppp": tol[5] = 1e-10      # pp: with periodic z its round-off-defined constant moves the last digits (the velocity and p do not see it)
    ok = all(e < t for e, t in zip(errs, tol))
    print(trial, name, ng, "P", P, "kchunk", os.environ.get("CALES_KCHUNK"), chosen, "steps", nsteps, "folded" if folded else "NOT-FOLDED", "OK" if ok else "BAD", " ".join(f"{e:.1e}" for e in errs), flush=True)
    bad += 0 if ok else 1
print("bad:", bad)
sys.exit(1 if bad else 0)
