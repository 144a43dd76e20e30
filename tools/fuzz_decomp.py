#!/usr/bin/env python3
"""Random sizes and rank counts: emulated y-slab ranks on one GPU against the single-rank run (development aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import load_golden, relerr
from tests.test_gpu_decomp import _single, _case
from cales_amd.decomp import run_loopback

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
names = ["chan_dsmag", "chan_dsmag_wm", "chan_smag_wm", "tgv_dsmag_ppp", "cavity_nnn", "duct_smag_wm_imp1d", "couette_imp3d_ops", "cavity_imp3d", "devchan_nd", "halfchan_imp1d",
         "duct_dsmag_wm", "duct_dsmag", "cavity_dsmag", "duct_smag_wm", "tgv_ppp", "chan_smag"]
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    name = names[trial % len(names)]
    P = int(rng.choice([2, 3, 4]))
    ng = (int(2 * rng.randint(4, 40)), int(2 * P * rng.randint(2, 8)), int(2 * rng.randint(5, 40)))
    if trial % 3 == 2: ng = (int(2 ** rng.randint(4, 8)),) + ng[1:]      # power-of-two rows (see fuzz_sizes.py)
    if os.environ.get("FUZZ_POW2"):      # power-of-two rows AND y lines: radix-8 passes both ways; periodic x and y then pack the modes 0 and n1/2 (k_gaussel_nyq)
        P = int(rng.choice([2, 4, 8]))
        ng = (int(2 ** rng.randint(6, 9)), int(max(2 * P, 2 ** rng.randint(4, 8))), int(rng.choice([2 * rng.randint(5, 40), 2 * rng.randint(65, 300)])))
    if os.environ.get("FUZZ_SWITCHES"):      # three to five run-time switches at once (the single-rank run takes the same ones)
        pool = ["CALES_UNFUSED_RK", "CALES_UNFUSED_CORREC", "CALES_UNFOLDED_CORREC", "CALES_UNFOLDED_MOM", "CALES_LAZY_PROJECTION", "CALES_UNFUSED_FORCING", "CALES_UNFUSED_FILLPS", "CALES_UNFUSED_MEAN", "CALES_GAUSSEL_MARCH", "CALES_NO_NYQUIST_PACKING",
                "CALES_DSMAG_XGHOSTS", "CALES_WIDE_OFFSETS", "CALES_UNMERGED_BC", 
                "CALES_XGHOSTS_IN_STEP", "CALES_FFT_GENERIC", "CALES_HELMHOLTZ_Z_PER_COLUMN", "CALES_UNFUSED_IMP_RHS", "CALES_OVERLAP",
                "CALES_LOOPBACK_EVENTS", "CALES_DSMAG_REFERENCE_SEQUENCE", "CALES_SMAG_REFERENCE_SEQUENCE"]
        for k in pool: os.environ.pop(k, None)
        chosen = [str(k) for k in rng.choice(pool, size=rng.randint(3, 6), replace=False)]
        for k in chosen: os.environ[k] = "1"
        print("   switches:", " ".join(c[6:] for c in chosen), flush=True)
    try:
        case = _case(name, ng)
        if np.any(case.lwm != 0):      # a sampling height the reference accepts on this grid and slab (sanity.f90:224-231)
            case.hwm = max(float(case.hwm), 1.6 * max(float(case.l[d]) / ng[d] for d in range(3) if case.lwm[:, d].any()))
        u, v, w, p, visct, dt, div, dpdl = _single(case, 2)
    except Exception as e:
        print(name, ng, P, "refused:", str(e)[:80]); continue

    def body(h, r):
        h.upload_initial(); h.startup()
        dtr = 0.5 * h.chkdt()
        for _ in range(2):
            h.step(dtr)
        return h.download() + [h.lo, h.n]
    try:
        res = run_loopback(case, P, body)
    except Exception as e:
        if "sanity.f90" in str(e):      # a case the reference itself refuses on this grid / rank count
            print("SKIP", name, ng, P, str(e)[:100]); continue
        print("BAD", name, ng, P, "loopback failed:", str(e)[:100]); bad += 1; continue
    worst = 0.
    for r, (ur, vr, wr, pr, visr, lo, n) in enumerate(res):
        j0 = lo[1] - 1; sl = slice(j0 + 1, j0 + n[1] + 1)
        for a, b in ((ur, u), (vr, v), (wr, w), (visr, visct)):
            worst = max(worst, relerr(a[:, 1:-1, :], b[:, sl, :]))
    ok = worst < 1e-9
    bad += not ok
    print("OK " if ok else "BAD", name, ng, "P", P, "%.1e" % worst, flush=True)
print("bad:", bad)
