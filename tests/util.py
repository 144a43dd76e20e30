"""Shared helpers for the tests: golden fixtures -> Case, comparison norms."""
import os

import numpy as np

from cales_amd.nml import parse_text

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RK = [(32. / 60., 0.), (25. / 60., -17. / 60.), (45. / 60., -25. / 60.)]      # reference src/param.f90:27-29
FULL_CASES = ["tgv_ppp", "tgv_dsmag_ppp", "chan_smag_wm", "chan_smag", "chan_dsmag", "chan_dsmag_wm", "duct_smag_wm", "duct_smag_wm_imp1d",
              "cavity_nnn", "devchan_nd", "halfchan_imp1d", "duct_dsmag_wm", "duct_dsmag", "cavity_dsmag"]


# cases DERIVED from a golden file's input.nml (compared with the oracle only, no reference-made stage vectors carry these names)
DERIVED = {"duct_smag": ("duct_smag_wm", lambda case: case.lwm.fill(0))}      # static Smagorinsky duct with no-slip walls, no wall model


def load_golden(name):
    edit = None
    if name in DERIVED:
        name, edit = DERIVED[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    case = parse_text(str(g["input_nml"]))
    case.impdiff = int(g["impdiff"])
    if edit:
        edit(case)
    return g, case


def F(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


def relerr(a, b):
    """L-infinity error scaled by the field maximum."""
    a = np.asarray(a); b = np.asarray(b)
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


def triperiodic_solve_scipy(o, case, rhs):
    """The reference's solve of a triply periodic pressure problem (solver.f90:20-80 with gaussel_periodic / dgtsv_homebrewed, :109-179, +eps pivots,
    one operation at a time) evaluated with scipy's FFTs instead of the oracle's own: a second, independent evaluation of the SAME algorithm.
    Columns are solved with the reference's sequential order, vectorised over the columns (numpy rounds every operation, no contraction)."""
    import scipy.fft as sf
    eps = np.finfo(float).eps
    ng = tuple(int(x) for x in case.ng); n = ng[2]
    lam, a, b, c, nrm = o.solver_operands(0)
    hx = np.minimum(np.arange(ng[0]), ng[0] - np.arange(ng[0])); hy = np.minimum(np.arange(ng[1]), ng[1] - np.arange(ng[1]))
    L = lam[np.ix_(hx, hy)].reshape(-1)      # eigenvalue of complex mode (kx, ky) = that of the half-complex entries of wavenumbers min(k, n - k)

    def dgtsv(m, bb, p):
        d = np.zeros_like(bb); z = 1. / (bb[0] + eps); d[0] = c[0] * z; p[0] = p[0] * z
        for l in range(1, m):
            z = 1. / (bb[l] - a[l] * d[l - 1] + eps); d[l] = c[l] * z; p[l] = (p[l] - a[l] * p[l - 1]) * z
        for l in range(m - 2, -1, -1):
            p[l] = p[l] - d[l] * p[l + 1]

    def periodic(P):
        bb = b[:, None] + L[None, :]
        p1 = P[:n - 1].copy(); dgtsv(n - 1, bb, p1)
        p2 = np.zeros((n - 1, P.shape[1])); p2[0] = -a[0]; p2[n - 2] = -c[n - 2]; dgtsv(n - 1, bb, p2)
        pn = (P[n - 1] - c[n - 1] * p1[0] - a[n - 1] * p1[n - 2]) / (bb[n - 1] + c[n - 1] * p2[0] + a[n - 1] * p2[n - 2] + eps)
        out = np.empty_like(P); out[n - 1] = pn; out[:n - 1] = p1 + p2 * pn
        return out

    X = sf.fft2(rhs[1:-1, 1:-1, 1:-1], axes=(0, 1)).reshape(-1, n).T.copy()
    Y = periodic(X.real.copy()) + 1j * periodic(X.imag.copy())
    return sf.ifft2(Y.T.reshape(ng[0], ng[1], n), axes=(0, 1)).real


def perturbed_tgv_rhs(o, case, seed=7):
    """div(u*)/dt of the Taylor-Green field with 2 % noise (what the first pressure solve of the fuzzers' cases sees); returns (pp, u, v, w, p, dt)."""
    from cales_amd.hotpath import initflow
    ng = tuple(int(x) for x in case.ng)
    rng = np.random.RandomState(seed)
    u, v, w, p = initflow(case)
    for x in (u, v, w):
        x[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
    visct, pp = o.zeros(), o.zeros()
    o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
    dt = 0.5 * o.chkdt(visct, u, v, w)
    o.fillps(1. / dt, u, v, w, pp)
    return pp, u, v, w, p, dt


def oracle_steps(case, nsteps, seed, ulp_seed=None, nthreads=8):
    """`nsteps` steps of the oracle from the perturbed initial field of the fuzzers / test_time_steps (2 % noise, RandomState(seed)); with `ulp_seed` every
    initial velocity value is moved by ONE unit in the last place, random sign. Returns (u, v, w, p, dt)."""
    from cales_amd.hotpath import initflow
    from oracle.oracle import Oracle
    ng = tuple(int(x) for x in case.ng)
    o = Oracle(case, nthreads=nthreads)
    rng = np.random.RandomState(seed)
    u, v, w, p = initflow(case)
    for a in (u, v, w):
        a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
    if ulp_seed is not None:
        r2 = np.random.RandomState(ulp_seed)
        for a in (u, v, w):
            a *= 1. + np.finfo(float).eps * (r2.randint(0, 2, size=a.shape) * 2 - 1)
    visct, pp = o.zeros(), o.zeros()
    o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
    dt = 0.5 * o.chkdt(visct, u, v, w)
    for _ in range(nsteps):
        o.step(dt, u, v, w, p, pp, visct)
    o.close()
    return u, v, w, p, dt


def one_ulp_sensitivity(case, nsteps, seed, trials=2):
    """How far the reference algorithm's own result (the oracle's) moves when every initial velocity value moves by one unit in the last place:
    the largest relative change of u, v, w over `trials` random sign patterns. ~1e-15 for well-posed cases."""
    base = oracle_steps(case, nsteps, seed)
    worst = 0.
    for t in range(trials):
        pert = oracle_steps(case, nsteps, seed, ulp_seed=100 + t)
        worst = max(worst, max(relerr(pert[i], base[i]) for i in range(3)))
    return worst, base


def triperiodic_bounds(case, fields, p, dt, nsteps, sens):
    """Per-field bounds (relative to each field's own maximum, as relerr measures) for time steps of a triply periodic box whose pressure the reference
    algorithm returns as C + p' with a round-off-defined constant C (solver.f90:160-178 on a grid whose dzf is not exactly uniform): (1) four times the
    algorithm's own response to one unit in the last place (`sens`, one_ulp_sensitivity); (2) the digits C costs: p' is carried with an absolute error of
    eps |C| that depends on the summation order of the transforms (two CPU evaluations of the same algorithm differ by 2-200 eps |C| per solve,
    tests/test_oracle_solver.py; 100 here), and every one of the 3 nsteps projections moves the velocity by dt grad p'. |C| is taken from the mean of the
    accumulated pressure."""
    eps = np.finfo(float).eps
    C = abs(float(np.asarray(p)[1:-1, 1:-1, 1:-1].mean()))
    dxi = max(float(case.ng[d]) / float(case.l[d]) for d in range(3))
    return [1e-9 + 4. * sens + 100. * eps * C * dt * dxi * 3 * nsteps / max(float(np.abs(f).max()), 1e-300) for f in fields]
