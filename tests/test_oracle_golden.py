"""The CPU restatement (oracle/cales_oracle.c) against golden vectors produced by the reference's
own compiled operators (tests/golden/gen_golden.py). Each operator is fed the GOLDEN input of its
stage, so errors do not accumulate. Tolerance 1e-13 relative; most stages are
bit-identical because the oracle keeps the reference's expression order."""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle
from tests.util import FULL_CASES, RK, F, load_golden, relerr

TOL = 1e-13


@pytest.mark.parametrize("name", FULL_CASES + ["couette_imp3d_ops"])
def test_setup_products(name):
    g, case = load_golden(name)
    o = Oracle(case)
    grid = o.grid()
    for k in ("dzc", "dzf", "zc", "zf"):
        assert relerr(grid[k], g["g_" + k]) < 1e-15, k
    assert (o.index_wm() == g["par_index_wm"]).all()
    assert (o.cbcvel() == g["par_cbcvel_after_initbc"]).all()
    for a, b in zip(o.rhsbp(), (g["rhsbp_x"], g["rhsbp_y"], g["rhsbp_z"])):
        assert np.abs(a - b).max() <= 1e-15 * max(1., np.abs(b).max())
    # parsed namelist == what the reference's read_input produced
    for k in ("ng", "l", "gr", "cfl", "dtmax", "dt_f", "visci", "bcvel", "bcpre", "bcsgs", "bforce", "velf", "hwm",
              "is_forced", "lwm", "cbcpre", "cbcsgs", "nstep", "icheck", "isave", "stop_type"):
        assert np.all(np.asarray(getattr(case, k)) == g["par_" + k]), k
    assert case.inivel == str(g["par_inivel"]) and case.sgstype == str(g["par_sgstype"])
    assert np.allclose(case.dl, g["par_dl"], rtol=1e-16, atol=0) and case.visc == float(g["par_visc"])


def test_grids():
    g = np.load(__import__("os").path.join(__import__("tests.util", fromlist=["GOLD"]).GOLD, "grids.npz"))
    import ctypes as C
    from oracle.oracle import build
    lib = C.CDLL(build())
    q = 0
    while f"g{q}_spec" in g.files:
        gtype, gr, n3, lz = g[f"g{q}_spec"]
        n3 = int(n3)
        out = [np.zeros(n3 + 2) for _ in range(4)]
        lib.o_initgrid(int(gtype), n3, C.c_double(gr), C.c_double(lz), *[a.ctypes.data_as(C.c_void_p) for a in out])
        for a, k in zip(out, ("dzc", "dzf", "zc", "zf")):
            assert relerr(a, g[f"g{q}_{k}"]) < 4e-16, (q, k)
        q += 1
    assert q == 8


@pytest.mark.parametrize("name", FULL_CASES)
def test_initflow(name):
    g, case = load_golden(name)
    o = Oracle(case)
    u, v, w, p = o.initflow(case.inivel, case.is_wallturb)
    for a, k in zip((u, v, w, p), "uvwp"):
        ref = g["if_" + k]
        assert np.abs(a - ref).max() <= 2e-15 * max(1., np.abs(ref).max()), k


@pytest.mark.parametrize("name", FULL_CASES)
def test_startup_and_substeps(name):
    g, case = load_golden(name)
    o = Oracle(case)
    imp = case.impdiff
    # start-up: bounduvw, boundp, cmpt_sgs, boundp(visct), chkdt   (main.f90:370-375,395)
    u, v, w, p = (F(g["s0raw_" + k]) for k in "uvwp")
    o.bounduvw(u, v, w, True, False); o.boundp(p, 0)
    for a, k in zip((u, v, w, p), "uvwp"):
        assert relerr(a, g["s0_" + k]) < TOL, ("s0", k)
    visct = o.zeros()
    o.cmpt_sgs(F(g["s0_u"]), F(g["s0_v"]), F(g["s0_w"]), visct)
    assert relerr(visct[1:-1, 1:-1, 1:-1], g["s0_visct_nobc"][1:-1, 1:-1, 1:-1]) < TOL
    o.boundp(visct, 1)
    assert relerr(visct, g["s0_visct"]) < TOL
    if np.any(case.lwm != 0):
        for iv, nm in ((1, "bcu"), (2, "bcv"), (3, "bcw")):
            for a, d in zip(o.bcvel_planes(iv), "xyz"):
                _cmp_wm_planes(a, g[f"s0_{nm}_{d}"], case, iv, d)
    s0 = [F(g["s0_" + k]) for k in ("u", "v", "w", "p", "visct")]
    assert abs(o.chkdt(s0[4], s0[0], s0[1], s0[2]) / float(g["dt_cfl"]) - 1) < 1e-14
    m = o.mom(s0[0], s0[1], s0[2], s0[4])
    names = ("dudt", "dvdt", "dwdt") + (("dudtd", "dvdtd", "dwdtd") if imp else ())
    for a, k in zip(m, names):
        assert relerr(a, g["m_" + k]) < TOL, k
    assert abs(o.bulk_mean(s0[0], "f") - float(g["mean_u_f"])) <= 1e-15 * max(1, abs(float(g["mean_u_f"])))
    d0 = o.chkdiv(s0[0], s0[1], s0[2])
    assert abs(d0[1] / g["div0"][1] - 1) < 1e-14 and abs(d0[0] - g["div0"][0]) < 1e-12 * max(1, abs(g["div0"][0]))

    dt = float(g["dt"])
    prev = dict(u=g["s0_u"], v=g["s0_v"], w=g["s0_w"], p=g["s0_p"], visct=g["s0_visct"])
    for irk in (1, 2, 3):
        K = f"r{irk}_"
        dtrk = (RK[irk - 1][0] + RK[irk - 1][1]) * dt; dtrki = dtrk ** (-1)
        alpha = -.5 * case.visc * dtrk if imp else 0.
        # rk + bulk_forcing
        u, v, w = F(prev["u"]), F(prev["v"]), F(prev["w"])
        f = o.rk(irk, dt, F(prev["p"]), F(prev["visct"]), u, v, w)
        o.bulk_forcing(f, u, v, w)
        assert np.abs(f - g[K + "s1_f"]).max() < 1e-14
        for a, k in zip((u, v, w), "uvw"):
            assert relerr(a, g[K + "s1_" + k]) < TOL, (K, "s1", k)
        nxt = "s1"
        if imp == 2:
            u, v, w = (F(g[K + "s1_" + k]) for k in "uvw")
            for iv, q in ((1, u), (2, v), (3, w)):
                o.updt_rhs_b_velz(iv, alpha, q)
            for a, k in zip((u, v, w), "uvw"):
                assert relerr(a, g[K + "s1a_" + k]) < TOL, (K, "s1a", k)
            nxt = "s1b"
        # bounduvw (wall model updated)
        sfx = "_orc" if nxt == "s1b" else ""
        u, v, w = (F(g[K + nxt + "_" + k + sfx]) for k in "uvw")
        o.bounduvw(u, v, w, True, False)
        for a, k in zip((u, v, w), "uvw"):
            assert relerr(a, g[K + "s2_" + k]) < TOL, (K, "s2", k)
        # fillps + boundary r.h.s.
        pp = o.zeros()
        o.fillps(dtrki, F(g[K + "s2_u"]), F(g[K + "s2_v"]), F(g[K + "s2_w"]), pp)
        o.updt_rhs_b_p(pp)
        n = o.n
        assert relerr(pp[1:-1, 1:-1, 1:-1], g[K + "s3_pp"][1:-1, 1:-1, 1:-1]) < TOL
        # boundp(pp) from the stored solver output
        pp = F(g[K + "s5_pp"]); pp[0, :, :] = 0; pp[-1, :, :] = 0; pp[:, 0, :] = 0; pp[:, -1, :] = 0; pp[:, :, 0] = 0; pp[:, :, -1] = 0
        o.boundp(pp, 0)
        assert relerr(pp, g[K + "s5_pp"]) < TOL
        # correc + bounduvw(is_correc)
        u, v, w = (F(g[K + "s2_" + k]) for k in "uvw")
        o.correc(dtrk, F(g[K + "s5_pp"]), u, v, w)
        if irk == 1:
            for a, k in zip((u, v, w), "uvw"):
                assert relerr(a, g[K + "s6_" + k]) < TOL, (K, "s6", k)
        o.bounduvw(u, v, w, True, True)
        for a, k in zip((u, v, w), "uvw"):
            assert relerr(a, g[K + "s7_" + k]) < TOL, (K, "s7", k)
        # updatep + boundp
        p = F(prev["p"])
        o.updatep(alpha, F(g[K + "s5_pp"]), p); o.boundp(p, 0)
        assert relerr(p, g[K + "s8_p"]) < TOL
        # sgs
        visct = F(prev["visct"])
        o.cmpt_sgs(F(g[K + "s7_u"]), F(g[K + "s7_v"]), F(g[K + "s7_w"]), visct); o.boundp(visct, 1)
        assert relerr(visct, g[K + "s9_visct"]) < (TOL if case.sgstype != "dsmag" else 1e-11), (K, "s9")
        prev = dict(u=g[K + "s7_u"], v=g[K + "s7_v"], w=g[K + "s7_w"], p=g[K + "s8_p"], visct=g[K + "s9_visct"])
    assert abs(o.chkdt(F(prev["visct"]), F(prev["u"]), F(prev["v"]), F(prev["w"])) / float(g["dt_cfl_end"]) - 1) < 1e-14


def _cmp_wm_planes(a, ref, case, ivel, d):
    """bc planes: compare where the reference defines them (wall-model loops do not touch every ghost entry)."""
    assert np.abs(a - ref).max() <= 1e-13 * max(1., np.abs(ref).max()), (ivel, d)


def test_imp3d_operators():
    g, case = load_golden("couette_imp3d_ops")
    o = Oracle(case)
    s0 = [F(g["s0_" + k]) for k in ("u", "v", "w", "p", "visct")]
    for a, k in zip(o.mom(s0[0], s0[1], s0[2], s0[4]), ("dudt", "dvdt", "dwdt", "dudtd", "dvdtd", "dwdtd")):
        assert relerr(a, g["m_" + k]) < TOL, k
    u, v, w = s0[0].copy(order="F"), s0[1].copy(order="F"), s0[2].copy(order="F")
    for irk in (1, 2):
        o.rk(irk, float(g["dt"]), s0[3], s0[4], u, v, w)
        for a, k in zip((u, v, w), "uvw"):
            assert relerr(a, g[f"r{irk}_s1_{k}"]) < TOL
    p = s0[3].copy(order="F")
    o.updatep(float(g["upd_alpha"]), F(g["upd_pp"]), p)
    assert relerr(p, g["upd_p"]) < TOL
    assert abs(o.chkdt(s0[4], u, v, w) / float(g["dt_cfl"]) - 1) < 1e-14


@pytest.mark.parametrize("name", FULL_CASES)
def test_plane_statistics(name):
    """The plane statistics of the reference's default out1d.h90 (out1d_single_point_chan, src/output.f90:509-1061: 27 single-point
    sums, 38 budget sums, 6 divergence measures per z plane), produced by the reference's own routine on the golden end-of-step state:
    the C restatement o_stats_chan and the numpy restatement oracle/stats_np.py reproduce them (same terms, the plane sums in another
    order: 1e-13 of each column's largest entry; the two max-norm leakage columns exactly)."""
    from oracle import stats_np
    g, case = load_golden(name)
    o = Oracle(case)
    u, v, w, p, vis = (F(g[k]) for k in ("r3_s7_u", "r3_s7_v", "r3_s7_w", "r3_s8_p", "r3_s9_visct"))
    st = o.stats_chan(u, v, w, p, vis)
    gr = o.grid()
    dx, dy = float(case.dl[0]), float(case.dl[1])
    bud = stats_np.budget_terms(u, v, w, p, dx, dy, gr["dzc"], gr["dzf"], float(case.l[0]), float(case.l[1]))
    leak = stats_np.leakage_terms(u, v, w, dx, dy, gr["dzf"], float(case.l[0]), float(case.l[1]))
    for got, ref, nm in ((st, g["st_chan"], "single-point"), (bud, g["st_budget"], "budget"), (leak, g["st_leak"], "leakage")):
        assert got.shape == ref.shape, nm
        scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-300)
        assert (np.abs(got - ref) <= 1e-13 * scale).all(), (nm, np.abs(got - ref).max(axis=1) / scale[:, 0])      # measured: 4e-16, 3e-14, 1e-15


@pytest.mark.parametrize("name", FULL_CASES)
def test_solver_operands_and_z_sweep(name):
    """eigenvalues, tridmatrix (initsolver.f90:66-169) and the tridiagonal sweep gaussel / gaussel_periodic / dgtsv_homebrewed with its
    `+eps` pivots (solver.f90:82-179), by the reference's own routines (compiled from their lines by oracle/ref/Makefile): the restatement
    reproduces the operands to round-off of the eigenvalue formula and the sweep bit for bit."""
    g, case = load_golden(name)
    o = Oracle(case)
    lam, a, b, c, _ = o.solver_operands(0)
    lam_ref = g["sol_lamx"][:, None] * case.dli[0] ** 2 + g["sol_lamy"][None, :] * case.dli[1] ** 2
    assert np.abs(lam - lam_ref).max() <= 4e-16 * np.abs(lam_ref).max()
    for x, k in ((a, "sol_a"), (b, "sol_b"), (c, "sol_c")):
        assert np.array_equal(x, g[k]), k
    pz = o.zeros(); pz[1:-1, 1:-1, 1:-1] = g["sol_gz_in"]
    o.solver_zsweep(pz)
    out = pz[1:-1, 1:-1, 1:-1]
    if np.array_equal(lam, lam_ref):
        assert np.array_equal(out, g["sol_gz_out"])
    else:       # eigenvalues differing in the last bit move the singular/near-singular columns by that much
        assert relerr(out, g["sol_gz_out"]) < 1e-10
    if case.impdiff == 2:       # solver_gaussel_z == a transposition around gaussel: bit-identical to the reference's routine
        for irk in (1, 2, 3):
            for k in "uvw":
                assert np.array_equal(g[f"r{irk}_s1b_{k}"], g[f"r{irk}_s1b_{k}_orc"]), (irk, k)


END_CASES = ["chan_dsmag_p2", "chan_smag_p2", "duct_dsmag_p2", "tgv_ppp_p2", "chan_dsmag_x64", "chan_dsmag_x128", "tgv_dsmag_ppp_x64", "chan_smag_wm_x64", "duct_smag_wm_x64", "duct_smag_wm_imp1d_x64",
             "chan_nosgs_x64", "cavity_nnn_x64", "halfchan_imp1d_x64", "duct_dsmag_x64", "tgv_ppp_x64"]


@pytest.mark.parametrize("name", END_CASES)
def test_step_at_power_of_two_rows(name):
    """End-of-step state made by the reference's modules at power-of-two row lengths (gen_golden.py END_ONLY): the oracle's whole step against it."""
    g, case = load_golden(name)
    o = Oracle(case, nthreads=4)
    u, v, w, p = (F(g["s0raw_" + k]) for k in "uvwp")
    visct, pp = o.zeros(), o.zeros()
    o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
    assert abs(o.chkdt(visct, u, v, w) / float(g["dt_cfl"]) - 1) < 1e-12
    dpdl = o.step(float(g["dt"]), u, v, w, p, pp, visct)
    for a, k in zip((u, v, w), "uvw"):
        assert relerr(a, g["r3_s7_" + k]) < 1e-12, k
    # (the pressure's constant is the zero mode of the singular solve through the +eps pivot, solver.f90:165: defined by round-off, it moves with the
    #  summation order of the run that made the vector -- compared after removing the mean where it does)
    pg = g["r3_s8_p"]
    assert relerr(p, pg) < 1e-11 or relerr(p - p[1:-1, 1:-1, 1:-1].mean(), pg - pg[1:-1, 1:-1, 1:-1].mean()) < 1e-11
    assert relerr(visct, g["r3_s9_visct"]) < 1e-10
    assert np.abs(dpdl - g["dpdl"]).max() < 1e-10 * max(1., np.abs(g["dpdl"]).max())


def test_manifest_lists_every_golden_file():
    """tests/golden/manifest.json (written by gen_golden.py) names every case and carries the digest of every vector file as committed."""
    import hashlib, json, os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(here, "manifest.json")))
    files = sorted(f for f in os.listdir(here) if f.endswith(".npz"))
    assert sorted(man["files"]) == files
    assert set(c + ".npz" for c in man["cases"]) | {"grids.npz", "outstats.npz"} == set(files)
    for f in files:
        assert hashlib.sha256(open(os.path.join(here, f), "rb").read()).hexdigest()[:16] == man["files"][f], f


OUTSTATS_CASES = ["tgv_ppp", "chan_smag_wm", "chan_dsmag", "duct_smag_wm", "duct_smag_wm_imp1d", "duct_dsmag", "cavity_nnn", "devchan_nd", "halfchan_imp1d",
                  # power-of-two rows (the cases whose step leaves the x ghost columns alone): 16..64-point profiles
                  "chan_dsmag_p2", "chan_smag_p2", "duct_dsmag_p2", "tgv_ppp_p2"]


def printed_equal(mine, printed, what, floor=0.):
    """`printed` = numbers read back from a file the reference wrote with E16.7e3, i.e. 0.ddddddd x 10^e: half a unit of the seventh digit is
    between 5e-8 (mantissa near 1) and 5e-7 (mantissa near 0.1) of the value; sums that cancel to round-off are compared on the scale of the column
    or, where the whole column is round-off (the plane mean of w in a closed box), on `floor` = 1e-13 of the scale of the field that was summed."""
    mine, printed = np.asarray(mine, float), np.asarray(printed, float)
    scale = np.abs(printed).max() if printed.size else 0.
    assert np.all(np.abs(mine - printed) <= 5.1e-7 * np.abs(printed) + 1e-13 * scale + floor + 1e-300), (what, np.abs(mine - printed).max(), scale)


def outstats_of(name, fn):
    """(printed by the reference, computed by `fn(kind, ...)`) for the profiles and duct statistics of one golden case's end-of-step state"""
    g, case = load_golden(name)
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "outstats.npz"))
    u, v, w = F(g["r3_s7_u"]), F(g["r3_s7_v"]), F(g["r3_s7_w"])
    return g, case, G, (u, v, w)


@pytest.mark.parametrize("name", OUTSTATS_CASES)
def test_profiles_and_duct_statistics_against_reference(name):
    """out1d, out1d_chan, out2d_duct (src/output.f90:50-163, 317-507): the oracle against what the reference's own routines print (their lines are
    compiled by oracle/ref/Makefile; tests/golden/gen_golden.py --case outstats) for the end-of-step state of the golden cases."""
    g, case, G, (u, v, w) = outstats_of(name, None)
    o = Oracle(case)
    for key, (idir, fld, dzc) in dict(u_z=(3, u, 0), v_y=(2, v, 0), w_x=(1, w, 1), w_z=(3, w, 1), u_y=(2, u, 0)).items():
        printed_equal(o.out1d(idir, fld, bool(dzc)), G[f"{name}__out1d_{key}"][1], key, floor=1e-13 * np.abs(fld).max())
    vmax = max(np.abs(a).max() for a in (u, v, w))
    printed_equal(o.out1d_chan(u, v, w).T, G[name + "__out1d_chan"][:, 1:], "out1d_chan", floor=1e-13 * max(vmax, vmax ** 2))
    n2, n3 = int(case.ng[1]), int(case.ng[2])
    duct = G[name + "__out2d_duct"]                                   # rows (y, z, 9 values), j fastest
    printed_equal(o.out2d_duct(u, v, w).reshape(9, n2 * n3, order="F").T, duct[:, 2:], "out2d_duct", floor=1e-13 * max(vmax, vmax ** 2))
    # the coordinates the files carry: cell centres in y (uniform) and z (the grid's zc)
    dl2 = float(case.l[1]) / n2
    printed_equal(np.tile((np.arange(n2) + 0.5) * dl2, n3), duct[:, 0], "y")
    printed_equal(np.repeat(o.grid()["zc"][1:-1], n2), duct[:, 1], "z")


@pytest.mark.parametrize("name,ng", [("chan_dsmag", (48, 40, 36)), ("duct_smag_wm_imp1d", (32, 24, 40))])
def test_team_sums_agree_with_the_reference_order(name, ng):
    """Oracle(team_sums=True), the mode bench.py's CPU baseline times: the bulk mean and the total divergence are summed plane by plane over the OpenMP
    team instead of cell by cell on one thread (a serial sum over 1.3e8 cells three times a step would idle the host). Same value for every team size,
    and equal to the reference-order result to a few units in the last place -- after two steps the fields agree to 1e-12."""
    from cales_amd.hotpath import initflow
    g, case = load_golden(name)
    case.ng[:] = ng
    out = []
    for team, nth in ((False, 4), (True, 1), (True, 3), (True, 8)):
        o = Oracle(case, nthreads=nth, team_sums=team)
        u, v, w, p = initflow(case)
        rng = np.random.RandomState(3)
        for a in (u, v, w):
            a[1:-1, 1:-1, 1:-1] += 0.02 * (rng.rand(*ng) - 0.5)
        visct, pp = o.zeros(), o.zeros()
        o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
        dt = 0.5 * o.chkdt(visct, u, v, w)
        mean = o.bulk_mean(u, "f")
        for _ in range(2):
            o.step(dt, u, v, w, p, pp, visct)
        out.append((mean, u, v, w, visct, o.chkdiv(u, v, w)))
        o.close()
    ref = out[0]
    assert all(abs(x[0] - ref[0]) <= 1e-14 * abs(ref[0]) for x in out[1:])
    for x in out[1:]:
        for q in range(1, 5):
            assert relerr(x[q], ref[q]) < (1e-12 if q < 4 else 1e-9)
    # one value for every team size: bit for bit
    for x in out[2:]:
        assert x[0] == out[1][0] and all(np.array_equal(x[q], out[1][q]) for q in range(1, 5)) and x[5] == out[1][5]
