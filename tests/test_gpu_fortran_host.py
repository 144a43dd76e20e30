"""The Fortran host (cales_amd/fortran/cales, ISO_C_BINDING over the C-ABI) run as a user would run the
reference: `input.nml` in the working directory, `fld.bin` / `time.out` / `forcing.out` / `grid.bin` out.
Checks the checkpoint byte layout (reference src/load.f90:20-153, utils/read_binary_data/python/
read_restart_file.py:43-56), equality with the Python host driving the same library, and restart equivalence."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from tests.util import F, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cales_amd", "fortran", "cales")
EXE_MPI = os.path.join(ROOT, "cales_amd", "fortran", "cales_mpi")
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"


def _nml(name, **subs):
    g, case = load_golden(name)
    text = str(g["input_nml"])
    for k, v in subs.items():
        text, n = re.subn(rf"(?m)\b{k}\s*=\s*[^,\n]+", f"{k} = {v}", text, count=1)
        assert n == 1, k
    return text


def _run(tmp, text, args=(), cmd=None, env=None):
    os.makedirs(tmp, exist_ok=True)
    open(os.path.join(tmp, "input.nml"), "w").write(text)
    r = subprocess.run([*(cmd or [EXE]), *args], cwd=tmp, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _read_fld(path, ng):
    data = np.fromfile(path, dtype=np.float64)
    n = int(np.prod(ng))
    assert data.size == 4 * n + 2                                   # (4*N+2)*sizeof(rp), load.f90:44-52
    flds = [data[q * n:(q + 1) * n].reshape(ng, order="F") for q in range(4)]
    return flds, data[-2], int(round(data[-1]))


@pytest.mark.skipif(not os.path.exists(EXE), reason="Fortran host not built (amdflang absent)")
@pytest.mark.parametrize("name", ["tgv_ppp", "chan_smag_wm", "duct_smag_wm_imp1d", "couette_imp3d_ops"])
def test_fortran_host_equals_python_host(tmp_path, name):
    from cales_amd.hotpath import HotPath, initflow
    from cales_amd.nml import parse_text
    text = _nml(name, nstep=4, icheck=2, iout0d=2, iout1d=4, iout2d=4, iout3d=4, isave=100000)
    text = re.sub(r"stop_type\(1:3\) = .*", "stop_type(1:3) = T, F, F", text)
    imp = int(load_golden(name)[1].impdiff)           # the reference's build switches are a run-time argument of the host
    out = _run(str(tmp_path), text, args=(str(imp),) if imp else ())
    assert "*** Fim ***" in out
    case = parse_text(text); case.impdiff = imp
    ng = tuple(int(x) for x in case.ng)
    (u, v, w, p), time, istep = _read_fld(os.path.join(tmp_path, "fld.bin"), ng)
    assert istep == 4
    # same loop with the Python host (main.f90:395-396,523-527)
    h = HotPath(case)
    h.upload(*initflow(case)); h.startup()
    dt = min(case.cfl * h.chkdt(), case.dtmax); t = 0.
    for s in range(1, 5):
        t += dt; h.step(dt)
        if s % 2 == 0:
            dt = min(case.cfl * h.chkdt(), case.dtmax)
    gu, gv, gw, gp, _ = h.download()
    for a, b in ((u, gu), (v, gv), (w, gw), (p, gp)):
        assert np.array_equal(a, b[1:-1, 1:-1, 1:-1])               # same library, same sequence: bit-identical
    assert abs(time - t) < 1e-15 * max(1., t)
    tout = np.loadtxt(os.path.join(tmp_path, "time.out"))
    assert tout.shape == (2, 3) and tout[-1, 0] == 4.
    if case.is_forced.any():
        fo = np.loadtxt(os.path.join(tmp_path, "forcing.out"))
        assert fo.shape == (2, 7) and abs(fo[-1, 4] - 1.) < 1e-10      # bulk velocity held at velf = 1
    grid = np.fromfile(os.path.join(tmp_path, "grid.bin"))
    g, _ = load_golden(name)
    assert np.allclose(grid[:ng[2]], g["g_dzc"][1:-1], rtol=1e-15)
    # plane statistics written at iout1d (out1d.h90 -> out1d_single_point_chan): zc, zf, 27 columns, dzc, dzf per plane + raw .bin;
    # only for channels (walls in z, periodic x and y): the reference's out1d.h90 is a per-case include
    is_chan = bool((case.cbcpre[:, :2] == "P").all() and (case.cbcvel[:, 2, 2] == "D").all())
    # ducts along x (walls in y and z) get the cross-stream maps of out2d_duct instead (src/out1d.h90:37): one text file in the reference's format
    is_duct = bool((case.cbcpre[:, 0] == "P").all() and (case.cbcvel[:, 1, 1] == "D").all() and (case.cbcvel[:, 2, 2] == "D").all())
    assert os.path.exists(os.path.join(tmp_path, "velstats_fld_0000004.out")) == (is_chan or is_duct)
    assert os.path.exists(os.path.join(tmp_path, "velstats_fld_0000000.out")) == (is_chan or is_duct)      # initial field, main.f90:377-395
    if is_duct:
        from tests.test_oracle_golden import printed_equal
        txt = np.loadtxt(os.path.join(tmp_path, "velstats_fld_0000004.out"))
        assert txt.shape == (ng[1] * ng[2], 11)
        printed_equal(h.out2d_duct().reshape(9, ng[1] * ng[2], order="F").T, txt[:, 2:], "out2d_duct as written by the host", floor=1e-13)
        printed_equal(np.tile((np.arange(ng[1]) + 0.5) * float(case.l[1]) / ng[1], ng[2]), txt[:, 0], "y")
    if is_chan:
        st = h.stats_chan()
        txt = np.loadtxt(os.path.join(tmp_path, "velstats_fld_0000004.out"))
        assert txt.shape == (ng[2], 31) and np.array_equal(txt[:, 2:29], st.T)
        assert np.array_equal(np.fromfile(os.path.join(tmp_path, "velstats_fld_0000004.bin")).reshape((27, ng[2]), order="F"), st)
        bud, leak = h.stats_chan_budgets()
        assert np.array_equal(np.fromfile(os.path.join(tmp_path, "velstats_fld_0000004_reystr_budget.bin")).reshape((38, ng[2]), order="F"), bud)
        assert np.array_equal(np.loadtxt(os.path.join(tmp_path, "velstats_fld_0000004_leakage.out"))[:, 2:8], leak.T)
    # plane and volume dumps (out2d.h90 / out3d.h90 defaults) with their log lines (output.f90:244-272)
    sl = np.fromfile(os.path.join(tmp_path, "vex_slice_fld_0000004.bin")).reshape((ng[0], ng[2]), order="F")
    assert np.array_equal(sl, gu[1:-1, ng[1] // 2, 1:-1])
    vol = np.fromfile(os.path.join(tmp_path, "pre_fld_0000004.bin")).reshape(ng, order="F")
    assert np.array_equal(vol, gp[1:-1, 1:-1, 1:-1])
    logl = open(os.path.join(tmp_path, "log_visu_3d.out")).read().splitlines()
    # five lines for the initial field (main.f90:377-395), five at step 4
    assert len(logl) == 10 and logl[0].split()[:2] == ["vex_fld_0000000.bin", "Velocity_X"] and int(logl[0].split()[-1]) == 0
    assert logl[5].split()[:2] == ["vex_fld_0000004.bin", "Velocity_X"] and int(logl[5].split()[-1]) == 4
    u0 = initflow(case)[0]
    assert np.array_equal(np.fromfile(os.path.join(tmp_path, "vex_fld_0000000.bin")).reshape(ng, order="F"), u0[1:-1, 1:-1, 1:-1])
    h.close()


@pytest.mark.skipif(not os.path.exists(EXE), reason="Fortran host not built (amdflang absent)")
def test_restart_equivalence(tmp_path):
    """4 steps in one go == 2 steps, checkpoint, restart, 2 more (the RK history is not saved; harmless because
    rkcoeff(2,1) = 0, src/param.f90:27-29)."""
    base = re.sub(r"stop_type\(1:3\) = .*", "stop_type(1:3) = T, F, F", _nml("chan_smag", nstep=4, icheck=2, iout0d=0, isave=100000))
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    _run(a, base)
    _run(b, base.replace("nstep = 4", "nstep = 2"))
    _run(b, base.replace("restart = F", "restart = T"))
    ng = (10, 12, 8)
    fa, ta, ia = _read_fld(os.path.join(a, "fld.bin"), ng)
    fb, tb, ib = _read_fld(os.path.join(b, "fld.bin"), ng)
    assert ia == ib == 4 and abs(ta - tb) < 1e-14
    for x, y in zip(fa[:3], fb[:3]):
        assert np.abs(x - y).max() < 1e-12 * max(1., np.abs(x).max())
    assert np.abs((fa[3] - fa[3].mean()) - (fb[3] - fb[3].mean())).max() < 1e-10


def _ngpu():
    import ctypes as C
    from cales_amd import capi
    n = C.c_int(0)
    return n.value if capi.lib().cales_device_count(C.byref(n)) == 0 else 0


@pytest.mark.skipif(not os.path.exists(EXE_MPI), reason="MPI Fortran host not built (amdflang or mpif.h absent)")
@pytest.mark.parametrize("launcher", ["singleton", "mpiexec"])
def test_fortran_mpi_host_one_rank(tmp_path, launcher):
    """The MPI host (one rank per GPU; src/main.f90:135-144, src/initmpi.f90:34-206) with ONE rank and the communicator forced on:
    device selection, token of cales_comm_unique_id carried by MPI_Bcast, cales_comm_init_rccl, the library's own RCCL exchanges,
    slab-wise checkpoint I/O -- and the same bytes in fld.bin as the serial host."""
    if launcher == "mpiexec" and not os.path.exists(MPIEXEC):
        pytest.skip("mpiexec not found")
    text = re.sub(r"stop_type\(1:3\) = .*", "stop_type(1:3) = T, F, F", _nml("chan_dsmag", nstep=3, icheck=2, iout0d=1, iout1d=3, iout2d=0, iout3d=3, isave=100000))
    a, b = str(tmp_path / "serial"), str(tmp_path / "mpi")
    _run(a, text)
    env = dict(os.environ, CALES_FORCE_COMM="1")
    out = _run(b, text, cmd=[EXE_MPI] if launcher == "singleton" else [MPIEXEC, "-n", "1", EXE_MPI], env=env)
    assert "RCCL communicator of" in out and "*** Fim ***" in out
    for f in ("fld.bin", "velstats_fld_0000003.bin", "vez_fld_0000003.bin", "grid.bin"):
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    assert np.array_equal(np.loadtxt(os.path.join(a, "forcing.out")), np.loadtxt(os.path.join(b, "forcing.out")))


@pytest.mark.skipif(not os.path.exists(EXE_MPI), reason="MPI Fortran host not built (amdflang or mpif.h absent)")
@pytest.mark.parametrize("name,P", [("chan_dsmag", 2), ("duct_smag_wm", 2)])
def test_fortran_mpi_host_ranks(tmp_path, name, P):
    """P MPI ranks on P GPUs (RCCL refuses two ranks on one device, so this needs a multi-GPU node and is skipped on the one-GPU box):
    y-slab run == serial run to the tolerance of tests/test_gpu_decomp.py, checkpoint assembled by the ranks in the reference's layout."""
    if _ngpu() < P or not os.path.exists(MPIEXEC):
        pytest.skip(f"needs {P} GPUs and mpiexec")
    g, case = load_golden(name)
    text = re.sub(r"stop_type\(1:3\) = .*", "stop_type(1:3) = T, F, F", _nml(name, nstep=3, icheck=2, iout0d=1, iout1d=0, iout2d=0, iout3d=0, isave=100000))
    text = re.sub(r"ng\(1:3\) = .*", "ng(1:3) = 32, 24, 16", text)
    a, b = str(tmp_path / "serial"), str(tmp_path / "mpi")
    _run(a, text); out = _run(b, text, cmd=[MPIEXEC, "-n", str(P), EXE_MPI])
    assert "*** Fim ***" in out
    fa, ta, ia = _read_fld(os.path.join(a, "fld.bin"), (32, 24, 16)); fb, tb, ib = _read_fld(os.path.join(b, "fld.bin"), (32, 24, 16))
    assert ia == ib == 3 and abs(ta - tb) < 1e-14
    for x, y in zip(fa[:3], fb[:3]):
        assert np.abs(x - y).max() < 1e-10 * max(1., np.abs(x).max())
    assert np.abs((fa[3] - fa[3].mean()) - (fb[3] - fb[3].mean())).max() < 1e-9 * max(1., np.abs(fa[3]).max())
