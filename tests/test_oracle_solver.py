"""The one boundary where parity is unpinned by the reference (FFTW r2r + solver.f90, not
buildable here): the restated transforms are checked against scipy.fft (FFTW conventions,
reference src/fft.f90:192-245) and the solve by the discrete identity L_h(solve(r)) = r built
with the pinned fillps/correc stencils (correc -> chkdiv ~ 0 is also in the golden files)."""
import numpy as np
import pytest
import scipy.fft as sf

from oracle.oracle import R2R, Oracle
from tests.util import F, load_golden


@pytest.mark.parametrize("n", [8, 12, 15, 20, 64, 90])
def test_r2r_kinds_match_fftw_definitions(n):
    g, case = load_golden("tgv_ppp")
    o = Oracle(case)
    rng = np.random.RandomState(n)
    x = rng.rand(n) - 0.5
    X = sf.rfft(x)
    hc = np.concatenate([X.real, X.imag[1:(n + 1) // 2][::-1]])
    assert np.allclose(o.r2r(R2R["R2HC"], x), hc, atol=1e-13)
    assert np.allclose(o.r2r(R2R["HC2R"], hc), n * x, atol=1e-12)
    assert np.allclose(o.r2r(R2R["REDFT10"], x), sf.dct(x, type=2), atol=1e-13)
    assert np.allclose(o.r2r(R2R["REDFT01"], x), sf.dct(x, type=3), atol=1e-13)
    assert np.allclose(o.r2r(R2R["RODFT10"], x), sf.dst(x, type=2), atol=1e-13)
    assert np.allclose(o.r2r(R2R["RODFT01"], x), sf.dst(x, type=3), atol=1e-13)
    assert np.allclose(o.r2r(R2R["REDFT11"], x), sf.dct(x, type=4), atol=1e-13)
    assert np.allclose(o.r2r(R2R["RODFT11"], x), sf.dst(x, type=4), atol=1e-13)
    assert np.allclose(o.r2r(R2R["REDFT00"], x), sf.dct(x, type=1), atol=1e-13)
    assert np.allclose(o.r2r(R2R["RODFT00"], x), sf.dst(x, type=1), atol=1e-13)


def laplacian(o, case, p):
    """7-point Laplacian consistent with fillps(correc(.)) (reference fillps.f90:40-44, correc.f90:45-66)."""
    g = o.grid(); dzci = 1 / g["dzc"]; dzfi = 1 / g["dzf"]
    dxi, dyi = case.dli[0], case.dli[1]
    c = p[1:-1, 1:-1, 1:-1]
    lap = (p[2:, 1:-1, 1:-1] - 2 * c + p[:-2, 1:-1, 1:-1]) * dxi ** 2 + (p[1:-1, 2:, 1:-1] - 2 * c + p[1:-1, :-2, 1:-1]) * dyi ** 2
    n3 = c.shape[2]
    k = np.arange(1, n3 + 1)
    lap += ((p[1:-1, 1:-1, 2:] - c) * dzci[k] - (c - p[1:-1, 1:-1, :-2]) * dzci[k - 1]) * dzfi[k]
    return lap


@pytest.mark.parametrize("name", ["tgv_ppp", "chan_smag", "duct_smag_wm", "cavity_nnn", "devchan_nd"])
def test_laplacian_identity(name):
    g, case = load_golden(name)
    o = Oracle(case)
    rng = np.random.RandomState(3)
    r = o.zeros(); r[1:-1, 1:-1, 1:-1] = rng.rand(*o.n) - 0.5
    singular = not np.any(case.cbcpre == "D")
    if singular:   # compatibility: zero volume-weighted mean
        dzf = o.grid()["dzf"][1:-1]
        r[1:-1, 1:-1, 1:-1] -= (r[1:-1, 1:-1, 1:-1] * dzf).sum() / (dzf.sum() * o.n[0] * o.n[1])
    p = r.copy(order="F")
    o.solver(p); o.boundp(p, 0)
    res = laplacian(o, case, p) - r[1:-1, 1:-1, 1:-1]
    assert np.abs(res).max() < 1e-11 * max(1., np.abs(r).max() * case.dli.max() ** 2 * 0 + 1)


@pytest.mark.parametrize("ivel", [1, 2, 3])
def test_helmholtz_3d_identity(ivel):
    """3-D implicit diffusion (main.f90:423-491): (1 + alpha L_h) applied to the oracle's solution gives back the r.h.s.;
    L_h = periodic second differences in x,y + the tridiagonal a,b,c of the component (initsolver.f90:100-169)."""
    g, case = load_golden("couette_imp3d_ops")
    case.ng[:] = (12, 10, 14)
    o = Oracle(case)
    n1, n2, n3 = case.ng
    nz = n3 - 1 if ivel == 3 else n3              # w lives on the faces: the wall face is not an unknown (solver.f90:49)
    _, a, b, c, _ = o.solver_operands(ivel)
    rng = np.random.RandomState(ivel)
    rhs = o.zeros(); rhs[1:-1, 1:-1, 1:nz + 1] = rng.rand(n1, n2, nz) - 0.5
    q = rhs.copy(order="F")
    alpha = -0.37
    o.solver_helmholtz(ivel, alpha, q)
    x = q[1:-1, 1:-1, 1:nz + 1]
    dxi2, dyi2 = (n1 / case.l[0]) ** 2, (n2 / case.l[1]) ** 2
    lap = (np.roll(x, -1, 0) - 2 * x + np.roll(x, 1, 0)) * dxi2 + (np.roll(x, -1, 1) - 2 * x + np.roll(x, 1, 1)) * dyi2
    lz = b[None, None, :nz] * x
    lz[:, :, 1:] += a[None, None, 1:nz] * x[:, :, :-1]
    lz[:, :, :-1] += c[None, None, :nz - 1] * x[:, :, 1:]
    back = x + alpha * (lap + lz)
    assert np.abs(back - rhs[1:-1, 1:-1, 1:nz + 1]).max() < 1e-12


def test_plane_statistics_known_answers():
    """o_stats_chan (first block of out1d_single_point_chan, output.f90:509-700) on fields with known plane averages:
    u = a + b z gives <u> = a + b zc, <u^2>, <du/dz> = b, spanwise vorticity b; visct = const with that shear gives the modelled <uw>."""
    g, case = load_golden("chan_dsmag")
    o = Oracle(case)
    gr = o.grid(); zc = gr["zc"]
    a, b, nu = 0.7, 1.3, 0.25
    u = o.zeros(); v = o.zeros(); w = o.zeros(); p = o.zeros(); vis = o.zeros()
    u[:, :, :] = (a + b * zc)[None, None, :]; p[:] = -2.; vis[:] = nu
    st = o.stats_chan(u, v, w, p, vis)
    k = np.arange(1, o.n[2] + 1)
    assert np.allclose(st[0], a + b * zc[k], rtol=1e-13) and np.allclose(st[3], (a + b * zc[k]) ** 2, rtol=1e-13)
    assert np.allclose(st[13], -2.) and np.allclose(st[14], 4.) and np.allclose(st[25], nu)
    assert np.allclose(st[26], b, rtol=1e-12) and np.allclose(st[16], b, rtol=1e-12) and np.allclose(st[19], b * b, rtol=1e-12)
    assert np.allclose(st[24], -nu * b, rtol=1e-12)                       # -visct (du/dz + dw/dx) at the cell edge
    for q in (1, 2, 4, 5, 6, 15, 17, 21, 22, 23):
        assert np.abs(st[q]).max() < 1e-13


def test_plane_budgets_known_answers():
    """oracle/stats_np.py (second and third block of out1d_single_point_chan, output.f90:700-1055) on fields with known plane values:
    u = a + b z (shear), v = c x-independent sine in x? no: v = s sin(2 pi x / lx), w = 0, p = p0."""
    from oracle import stats_np
    g, case = load_golden("chan_dsmag")
    o = Oracle(case); gr = o.grid(); zc = gr["zc"]; n1, n2, n3 = o.n
    lx, ly = float(case.l[0]), float(case.l[1]); dx, dy = lx / n1, ly / n2
    a, b, s0, p0 = 0.7, 1.3, 0.4, -2.
    u = o.zeros(); v = o.zeros(); w = o.zeros(); p = o.zeros()
    u[:, :, :] = (a + b * zc)[None, None, :]
    xc = (np.arange(n1 + 2) - 0.5) * dx
    v[:, :, :] = (s0 * np.sin(2 * np.pi * xc / lx))[:, None, None]
    p[:] = p0
    bt = stats_np.budget_terms(u, v, w, p, dx, dy, gr["dzc"], gr["dzf"], lx, ly)
    lk = stats_np.leakage_terms(u, v, w, dx, dy, gr["dzf"], lx, ly)
    k = np.arange(1, n3 + 1)
    assert np.allclose(bt[0], a + b * zc[k], rtol=1e-13) and np.allclose(bt[8], p0) and np.allclose(bt[24], p0)
    for q in (2, 6, 28):                                   # du/dz at the edge, its 4-point cell-centre average, the split term
        assert np.allclose(bt[q], b, rtol=1e-12), q
    assert np.allclose(bt[31], b * b, rtol=1e-12) and np.allclose(bt[10], b * b, rtol=1e-12)      # (du/dz)^2; uu dissipation = that alone
    dvdx2 = (s0 * 2 * np.sin(np.pi * dx / lx) / dx) ** 2 / 2          # plane mean of the squared one-cell difference of the sine
    assert np.allclose(bt[32], dvdx2, rtol=1e-12) and np.allclose(bt[14], dvdx2, rtol=1e-12)
    assert np.allclose(bt[26], b * p0, rtol=1e-12)                    # pressure-strain of uw: <du/dz> p
    for q in (4, 5, 9, 13, 15, 16, 17, 18, 19, 20, 21, 22, 23, 27, 33, 35, 36, 37):
        assert np.abs(bt[q]).max() < 1e-13, q
    assert np.abs(lk).max() < 1e-13


@pytest.mark.parametrize("ivel", [1, 2, 3])
def test_helmholtz_3d_identity_walls(ivel):
    """The same identity in a box with no-slip walls on all six faces (face-centred RODFT00 along the component's own direction,
    RODFT10/01 along the others, find_fft/eigenvalues with c_or_f, initsolver.f90:66-98, fft.f90:192-245): (1 + alpha L_h) x = r with
    the discrete Laplacian of the staggered component -- ghost = -interior for a cell-centred direction, wall faces = 0 for its own."""
    g, case = load_golden("cavity_nnn")
    case.ng[:] = (12, 10, 14); case.impdiff = 1
    o = Oracle(case)
    n = [int(x) for x in case.ng]
    nn = list(n); nn[ivel - 1] -= 1                       # the wall face of the component is not an unknown
    _, a, b, c, _ = o.solver_operands(ivel)
    rng = np.random.RandomState(10 + ivel)
    rhs = o.zeros(); rhs[1:nn[0] + 1, 1:nn[1] + 1, 1:nn[2] + 1] = rng.rand(*nn) - 0.5
    q = rhs.copy(order="F")
    alpha = -0.37
    o.solver_helmholtz(ivel, alpha, q)
    x = q[1:nn[0] + 1, 1:nn[1] + 1, 1:nn[2] + 1]

    def d2(x, axis, own, h2i):
        pad = [(1, 1) if ax == axis else (0, 0) for ax in range(3)]
        e = np.pad(x, pad)                                # zero wall faces (own direction)
        if not own:                                       # cell-centred: ghost = -first interior value (wall value 0 half a cell away)
            idx0 = [slice(None)] * 3; idx1 = [slice(None)] * 3
            idx0[axis] = 0; idx1[axis] = 1; e[tuple(idx0)] = -e[tuple(idx1)]
            idx0[axis] = -1; idx1[axis] = -2; e[tuple(idx0)] = -e[tuple(idx1)]
        sl = lambda s_: tuple(s_ if ax == axis else slice(None) for ax in range(3))
        return (e[sl(slice(2, None))] - 2 * e[sl(slice(1, -1))] + e[sl(slice(0, -2))]) * h2i
    lap = d2(x, 0, ivel == 1, (n[0] / case.l[0]) ** 2) + d2(x, 1, ivel == 2, (n[1] / case.l[1]) ** 2)
    nz = nn[2]
    lz = b[None, None, :nz] * x
    lz[:, :, 1:] += a[None, None, 1:nz] * x[:, :, :-1]
    lz[:, :, :-1] += c[None, None, :nz - 1] * x[:, :, 1:]
    back = x + alpha * (lap + lz)
    assert np.abs(back - rhs[1:nn[0] + 1, 1:nn[1] + 1, 1:nn[2] + 1]).max() < 1e-12


@pytest.mark.parametrize("pair,ivel", [("DN", 1), ("DN", 2), ("ND", 2), ("NN", 2), ("DD", 1), ("DD", 2)])
def test_helmholtz_3d_identity_open_x(pair, ivel):
    """Open boundaries in x (inflow / outflow): (1 + alpha L_h) x = r for the BC pairs whose transform set is an exact diagonalisation --
    every cell-centred pair (ivel = 2: REDFT10/01, RODFT10/01, REDFT11, RODFT11) and the face-centred DD (RODFT00) and DN (RODFT01/10: u = 0 on
    the inflow face, mirror about the outflow face). The face-centred NN (REDFT00 with eigenvalues of period n) and ND (REDFT10/01 with
    half-integer eigenvalues) of find_fft / eigenvalues are NOT exact inverses in the reference (residual of a few per cent); the
    restatement follows the reference there too and those two are not asserted here."""
    g, case = load_golden("devchan_nd")
    n1, n2, n3 = 12, 10, 14
    case.ng[:] = (n1, n2, n3); case.impdiff = 1; case.bcvel[:] = 0.
    for iv in range(3):
        case.cbcvel[0, 0, iv] = pair[0]; case.cbcvel[1, 0, iv] = pair[1]
    o = Oracle(case)
    own = ivel == 1
    cut = 1 if (own and pair == "DD") else 0
    nz = n3 - 1 if ivel == 3 else n3
    _, a, b, c, _ = o.solver_operands(ivel)
    rng = np.random.RandomState(5)
    rhs = o.zeros(); rhs[1:n1 - cut + 1, 1:-1, 1:nz + 1] = rng.rand(n1 - cut, n2, nz) - 0.5
    q = rhs.copy(order="F"); alpha = -0.37
    o.solver_helmholtz(ivel, alpha, q)
    x = q[1:n1 - cut + 1, 1:-1, 1:nz + 1]
    e = np.pad(x, [(1, 1), (0, 0), (0, 0)])
    if own:      # faces: Dirichlet face value 0 (not an unknown), Neumann face = mirror about it
        e[0] = 0. if pair[0] == "D" else e[2]
        e[-1] = 0. if pair[1] == "D" else e[-3]
    else:        # cell centres: ghost = -/+ first interior value
        e[0] = -e[1] if pair[0] == "D" else e[1]
        e[-1] = -e[-2] if pair[1] == "D" else e[-2]
    dxi2, dyi2 = (n1 / case.l[0]) ** 2, (n2 / case.l[1]) ** 2
    lap = (e[2:] - 2 * e[1:-1] + e[:-2]) * dxi2 + (np.roll(x, -1, 1) - 2 * x + np.roll(x, 1, 1)) * dyi2
    lz = b[None, None, :nz] * x
    lz[:, :, 1:] += a[None, None, 1:nz] * x[:, :, :-1]
    lz[:, :, :-1] += c[None, None, :nz - 1] * x[:, :, 1:]
    back = x + alpha * (lap + lz)
    assert np.abs(back - rhs[1:n1 - cut + 1, 1:-1, 1:nz + 1]).max() < 1e-12


@pytest.mark.parametrize("ng", [(10, 6, 12), (46, 74, 15), (24, 20, 18), (16, 12, 9), (16, 32, 48)])
def test_triperiodic_reference_solution_is_defined_only_to_eps_times_its_constant(ng):
    """Why the device cannot be held to 1e-10 against the reference algorithm on triply periodic boxes whose n3 is not a power of two -- shown on
    the CPU alone. initgrid's default-real arithmetic (initgrid.f90:63) leaves dzf non-uniform at 1e-7; the last pivot of the zero-eigenvalue
    column is then +-eps exactly and its numerator 1e-8, so the reference's pressure is C + p' with a constant C of 1e4..1e6 (solver.f90:109-150).
    Two CORRECT evaluations of that same algorithm -- the oracle (its own mixed-radix transforms) and tests.util.triperiodic_solve_scipy (scipy's
    pocketfft, the same sequential column solves) -- agree on C to 1e-4, yet their p' differ by 10..100 eps |C| / range(p'): the round-off of the
    inverse transforms scales with C. That is 3e-9..1e-7 here, above the 1e-10 asked; FFTW (the reference's own transforms) is a third summation
    order. The device with CALES_KEEP_NULL_MODE=1 is held to the same bound (tests/test_gpu_vs_oracle.py::test_triperiodic_reference_null_mode)."""
    from tests.util import perturbed_tgv_rhs, triperiodic_solve_scipy
    g, case = load_golden("tgv_ppp"); case.ng[:] = ng
    o = Oracle(case, nthreads=4)
    pp = perturbed_tgv_rhs(o, case)[0]
    a = triperiodic_solve_scipy(o, case, pp)
    ref = pp.copy(order="F"); o.solver(ref); b = ref[1:-1, 1:-1, 1:-1]
    C = b.mean(); rng_ = np.ptp(b - C)
    assert abs(C) > 1e4 and abs(a.mean() - C) < 1e-4 * abs(C)                 # the same huge constant ...
    d = np.abs((a - a.mean()) - (b - C)).max() / np.abs(b - C).max()
    unit = np.finfo(float).eps * abs(C) / rng_
    assert 2. * unit < d < 200. * unit, (d, unit)                             # ... and a p' that depends on the summation order at eps |C|
    assert d > 1e-10                                                          # i.e. above the bar that well-posed cases meet


def test_triperiodic_power_of_two_is_well_posed():
    """The same comparison where the grid arithmetic is exact (n3 a power of two): no constant, both evaluations agree to round-off."""
    from tests.util import perturbed_tgv_rhs, triperiodic_solve_scipy
    g, case = load_golden("tgv_ppp"); case.ng[:] = (16, 16, 16)
    o = Oracle(case, nthreads=4)
    pp = perturbed_tgv_rhs(o, case)[0]
    a = triperiodic_solve_scipy(o, case, pp)
    ref = pp.copy(order="F"); o.solver(ref); b = ref[1:-1, 1:-1, 1:-1]
    assert abs(b.mean()) < 1. and np.abs((a - a.mean()) - (b - b.mean())).max() < 1e-13 * np.abs(b - b.mean()).max()


@pytest.mark.parametrize("ng,lo,hi", [((74, 52, 26), 3e-9, 1e-6), ((46, 74, 15), 1e-10, 1e-7), ((20, 58, 12), 5e-11, 1e-7), ((32, 32, 16), 0., 1e-13), ((24, 20, 32), 0., 1e-13)])
def test_triperiodic_steps_move_with_one_unit_in_the_last_place(ng, lo, hi):
    """Conditioning of the reference algorithm itself, CPU only: two steps of the triply periodic box from an initial field whose every velocity value
    is moved by ONE unit in the last place. Where n3 is a power of two the result moves by ~1e-15; where it is not (float32 grid arithmetic,
    initgrid.f90:63 -> an incompatible singular pressure problem whose +eps pivot turns 1e-8 into a constant of 1e3..1e5, solver.f90:160-178) it
    moves by 1e-10..1e-7 of the velocity scale. No implementation can agree with the reference more closely than the reference agrees with itself:
    the GPU tests and the fuzzers hold the device to a small multiple of THIS number for such boxes (tests/util.py one_ulp_sensitivity)."""
    from tests.util import load_golden, one_ulp_sensitivity
    g, case = load_golden("tgv_ppp"); case.ng[:] = ng
    sens, _ = one_ulp_sensitivity(case, 2, seed=3)
    assert lo <= sens < hi, sens
