"""CPU-side checks of the product (no GPU needed): the C-ABI library loads and exports every symbol
include/cales.h declares; its host-only helpers reproduce the reference's set-up (golden vectors); the
namelist reader mirrors read_input (reference src/param.f90:88-224)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cales_amd import capi
from cales_amd.nml import NamelistError, parse_text
from oracle.oracle import Oracle
from tests.util import FULL_CASES, GOLD, load_golden, relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cales.h")).read()
    declared = set(re.findall(r"\b(cales_[a-z_0-9]+)\s*\(", hdr)) - {"cales_halo_cb", "cales_alltoall_cb", "cales_allreduce_cb"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = capi.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_single_precision_library_host_side():
    """libcales_hip_sp.so (-DCALES_SINGLE, the reference's -D_SINGLE_PRECISION): exports the same symbols, reports four-byte reals, and its
    host-only entries take float arguments: initgrid against the reference-made grids within single precision, check_case through the float struct.
    (The precision is process-wide in the Python host -- CALES_PRECISION=single --, so this binds the second library by hand.)"""
    import ctypes as C
    path = os.path.join(ROOT, "cales_amd", "libcales_hip_sp.so")
    assert os.path.exists(path), "build with python -c 'import __graft_entry__ as g; g.build()'"
    if os.environ.get("CALES_NO_TORCH") != "1":
        import torch  # noqa: F401  (one HIP runtime per process: see capi.lib)
    L = C.CDLL(path)
    for sname in capi.SYMBOLS:
        assert hasattr(L, sname), sname
    L.cales_real_size.restype = C.c_int
    assert L.cales_real_size() == 4 and capi.lib().cales_real_size() == 8
    g = np.load(os.path.join(GOLD, "grids.npz"))
    L.cales_initgrid.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float] + [C.c_void_p] * 4
    q = 0
    while f"g{q}_spec" in g.files:
        gtype, gr, n3, lz = g[f"g{q}_spec"]
        out = [np.zeros(int(n3) + 2, dtype=np.float32) for _ in range(4)]
        assert L.cales_initgrid(int(gtype), int(n3), float(gr), float(lz), *[a.ctypes.data_as(C.c_void_p) for a in out]) == 0
        for a, k in zip(out, ("dzc", "dzf", "zc", "zf")):
            assert np.abs(a - g[f"g{q}_{k}"]).max() < 2e-5 * np.abs(g[f"g{q}_{k}"]).max(), (q, k)
        q += 1
    assert q > 0


def test_no_cpu_fallback_and_no_oracle_in_product():
    """cales_create must fail loudly without a HIP device; nothing under cales_amd/ touches oracle/."""
    for root, _, files in os.walk(os.path.join(ROOT, "cales_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".f90", "Makefile")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle" not in txt.replace("the oracle", "").replace("against the oracle", "") or f.endswith(".hip"), (root, f)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g, case = load_golden("tgv_ppp")
    from cales_amd.hotpath import CalesError, HotPath
    with pytest.raises(CalesError):
        HotPath(case)


def test_initgrid_matches_reference():
    from cales_amd.hotpath import initgrid
    g = np.load(os.path.join(GOLD, "grids.npz"))
    q = 0
    while f"g{q}_spec" in g.files:
        gtype, gr, n3, lz = g[f"g{q}_spec"]
        out = initgrid(int(gtype), int(n3), float(gr), float(lz))
        for k in ("dzc", "dzf", "zc", "zf"):
            assert relerr(out[k], g[f"g{q}_{k}"]) < 4e-16, (q, k)
        q += 1


@pytest.mark.parametrize("name", FULL_CASES)
def test_initflow_matches_reference(name):
    from cales_amd.hotpath import initflow
    g, case = load_golden(name)
    for a, k in zip(initflow(case), "uvwp"):
        ref = g["if_" + k]
        assert np.abs(a - ref).max() <= 2e-15 * max(1., np.abs(ref).max()), k


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libcales_ref.so")),
                    reason="oracle/_ref (the compiled reference) is only present in the build container")
@pytest.mark.parametrize("inivel,wallturb", [("hdc", "F"), ("hdc", "T"), ("pdc", "T"), ("hcp", "F")])
def test_pressure_driven_profiles_match_compiled_reference(inivel, wallturb, tmp_path):
    """initial fields without a committed golden, against the reference's own initflow (oracle/_ref) where it is built;
    one case per process: the reference reads input.nml from the working directory"""
    import re
    import subprocess
    import sys
    from cales_amd.hotpath import initflow
    from cales_amd.nml import parse_text
    g, _ = load_golden("halfchan_imp1d")
    text = str(g["input_nml"])
    text = re.sub(r"inivel = .*", f"inivel = '{inivel}'", text)
    text = re.sub(r"is_wallturb = .*", f"is_wallturb = {wallturb}", text)
    text = re.sub(r"bforce\(1:3\) = .*", "bforce(1:3) = 0.7, 0., 0.", text)
    text = re.sub(r"is_forced\(1:3\) = .*", "is_forced(1:3) = F, F, F", text)
    open(tmp_path / "input.nml", "w").write(text)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from oracle.ref.refpy import Ref; "
            "u, v, w, p = Ref(0).initflow(); np.savez('out.npz', u=u, v=v, w=w, p=p)" % root)
    subprocess.run([sys.executable, "-c", code], cwd=tmp_path, check=True, timeout=300)
    ref = np.load(tmp_path / "out.npz")
    case = parse_text(text)
    for a, k in zip(initflow(case), "uvwp"):
        assert np.abs(a - ref[k]).max() <= 2e-15 * max(1., np.abs(ref[k]).max()), k
    assert np.abs(ref["u"]).max() > 0.


def test_unknown_initial_field_is_refused():
    from cales_amd.hotpath import CalesError, initflow
    g, case = load_golden("chan_smag")
    case.inivel = "nonsense"
    with pytest.raises(CalesError):
        initflow(case)


@pytest.mark.parametrize("kind", ["log", "hcl", "tbl"])
def test_noisy_initial_fields(kind):
    """log-law / temporal-boundary-layer profiles with +-5 % noise (initflow.f90:60-91,285-315). The reference's noise comes from the
    Fortran run-time's random_number (compiler-specific), so only the deterministic part and the statistics are checked: plane
    means follow the profile, the bulk velocity is ubulk after set_mean, the noise is bounded and decomposition-independent."""
    import ctypes as C
    from cales_amd import capi
    from cales_amd.hotpath import _p, initflow
    g, case = load_golden("chan_smag")
    case.ng[:] = (32, 16, 24); case.inivel = kind; case.is_wallturb = False
    u, v, w, p = initflow(case)
    ui = u[1:-1, 1:-1, 1:-1]
    prof = ui.mean(axis=(0, 1))
    amp = np.abs(ui - prof).max()
    assert 0. < amp < 0.06 and np.abs(v[1:-1, 1:-1, 1:-1]).max() <= 0.05 and np.abs(w[1:-1, 1:-1, 1:-1]).max() <= 0.05
    if kind != "tbl":
        o = Oracle(case); gr = o.grid()
        assert abs((ui * gr["dzf"][1:-1]).sum() / (gr["dzf"][1:-1].sum() * 32 * 16) - 1.) < 1e-12           # ubulk = velf = 1
        assert prof[0] < prof[3] < prof[11]                                                                  # log layer grows from the wall
    else:
        assert prof[0] > 0.9 and prof[-1] < 0.1
    # rows of a 2-rank decomposition == rows of the global field
    for r in range(2):
        cs = capi.make_case(case, 2, r)
        loc = [np.zeros((34, 10, 26), order="F") for _ in range(4)]
        assert capi.lib().cales_initflow_slab(C.byref(cs), kind.encode(), 0, *[_p(a) for a in loc]) == 0
        assert np.array_equal(loc[0][1:-1, 1:-1, 1:-1], u[1:-1, 1 + 8 * r:9 + 8 * r, 1:-1])


def test_check_case_rules(monkeypatch):
    """the rules of reference src/sanity.f90:115-274 that bound what the kernels must support"""
    from cales_amd.hotpath import CalesError, check_case
    g, case = load_golden("chan_smag_wm")
    check_case(case)
    bad = case.copy(); bad.cbcpre[:, 2] = "D"                  # velocity DD needs pressure NN
    with pytest.raises(CalesError):
        check_case(bad)
    bad = case.copy(); bad.is_forced[2] = True                 # forcing along a non-periodic direction
    with pytest.raises(CalesError):
        check_case(bad)
    bad = case.copy(); bad.cbcvel[0, 2, 0] = "N"               # wall-model faces must be all-Dirichlet
    with pytest.raises(CalesError):
        check_case(bad)
    bad = case.copy(); bad.bcpre[0, 0] = 1.                    # x,y pressure BC values must be zero
    with pytest.raises(CalesError):
        check_case(bad)
    ok = case.copy(); ok.impdiff = 1                           # 3-D implicit diffusion: periodic or wall pairs in x and y
    if all(ch == "P" for ch in ok.cbcvel[:, :2, :].ravel()):
        check_case(ok)
    ok = load_golden("duct_smag_wm")[1]; ok.impdiff = 1; ok.lwm[:] = 0      # walls in y (no wall model)
    check_case(ok)
    # beyond the reference's -D_IMPDIFF limits (sanity.f90:233-252: no NN pair, zero BC values in x and y): refused unless CALES_IMP3D_OPEN=1
    mv = ok.copy(); mv.bcvel[0, 1, 0] = 0.3                    # a moving wall in y
    ok2 = load_golden("devchan_nd")[1]; ok2.impdiff = 1; ok2.cbcsgs[:, 0] = "D"      # an open boundary (inflow / outflow in x)
    for c_ in (mv, ok2):
        with pytest.raises(CalesError):
            check_case(c_)
    monkeypatch.setenv("CALES_IMP3D_OPEN", "1")
    check_case(mv); check_case(ok2)
    bad = ok2.copy(); bad.cbcvel[:, 1, :] = "D"; bad.cbcvel[1, 1, 0] = "N"; bad.cbcvel[1, 1, 2] = "N"; bad.cbcpre[:, 1] = "N"; bad.cbcsgs[:, 1] = "D"
    bad.ng[1] = 9                                              # ND/DN pairs across y go through half-length lines: even ng(2) only
    with pytest.raises(CalesError):
        check_case(bad)
    with pytest.raises(CalesError):
        check_case(case, nranks=5)                             # ng(2) not divisible


def test_namelist_reader():
    g, case = load_golden("tgv_ppp")
    text = str(g["input_nml"])
    # the reference's examples/dns/* close &les with a backslash: accepted
    alt = re.sub(r"(?s)(&les.*?)\n/", lambda m: m.group(1) + "\n" + chr(92), text, count=1)
    assert alt.count(chr(92)) == 1
    c2 = parse_text(alt)
    assert c2.sgstype == case.sgstype and (c2.lwm == case.lwm).all() and c2.hwm == case.hwm
    assert c2.dt_f == -1.0                                     # default, param.f90:124
    assert (case.cbcvel == "P").all() and case.cbcvel.shape == (2, 3, 3)
    with pytest.raises(NamelistError):
        parse_text("&dns\nng(1:3) = 4,4,4\n/\n")               # &les missing (param.f90:144-150)
    with pytest.raises(NamelistError):
        parse_text(text.replace("gtype = 1", "gtype = 1, bogus = 3"))
    c3 = parse_text(text.replace("ng(1:3) = 12, 10, 8", "ng(1:3) = 3*16 ! repeat count"))
    assert (c3.ng == 16).all()
