"""The single-precision build (libcales_hip_sp.so = the same sources with -DCALES_SINGLE; the reference's -D_SINGLE_PRECISION,
src/precision.f90:11-20): every real of the C-ABI is a float. The precision is a process-wide choice (CALES_PRECISION=single), so the
runs happen in a worker process (tests/_single_worker.py); they are held to the FP64 oracle within single-precision tolerances:
1e-5 of the velocity scale on the velocities after three steps, 5e-4 on the pressure and the eddy viscosity (the Poisson solve and the
dynamic coefficient amplify the rounding of their inputs), divergence at the rounding level of the velocity gradients."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests.util import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(specs):
    env = dict(os.environ, CALES_PRECISION="single")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_single_worker.py"), *specs], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def _check(res):
    for spec, r in res.items():
        assert r["dtype"] == "float32", spec
        assert r["dt_rel"] < 1e-5, (spec, r)
        tie = spec.startswith("duct_smag")      # static Smagorinsky between four walls: the nearest-wall choice of diagonal cells hangs on rounding
        for k in "uvw":
            assert r[k] < (2e-4 if tie else 1e-5), (spec, k, r)
        assert r["p"] < (2e-3 if tie else 5e-4), (spec, r)
        if tie:
            assert r["visct_frac"] < 0.05, (spec, r)
        else:
            assert r["visct"] < 5e-4 and r["visct0"] < 5e-5, (spec, r)
        assert r["divmax"] < 2e-5 * max(1., r["divscale"]), (spec, r)


def test_single_precision_goldens():
    """The reference-made initial states of the golden cases (all BC sets, both SGS models, wall model, implicit diffusion 1-D and 3-D)."""
    _check(_worker([f"{n}:0:1" for n in ("tgv_ppp", "chan_smag", "chan_smag_wm", "chan_dsmag", "chan_dsmag_wm", "tgv_dsmag_ppp", "halfchan_imp1d",
                                         "duct_smag_wm", "duct_smag_wm_imp1d", "duct_dsmag_wm", "cavity_nnn", "cavity_dsmag", "devchan_nd")]))


def test_single_precision_production_kernels():
    """Sizes that take the radix-8 transforms, the in-LDS tridiagonal tile and the marching tile kernels with their k chunks and XCD bands."""
    _check(_worker(["chan_dsmag:128x64x64:1", "chan_smag_wm:128x128x32:1", "cavity_nnn:64x32x32:1", "chan_dsmag_wm:64x64x32:1", "tgv_ppp:64x64x64:1"]))


def test_single_precision_slab_ranks():
    """Several ranks (loopback on one GPU): float staging buffers, float exchanges."""
    _check(_worker(["chan_dsmag:64x32x32:2", "cavity_nnn:32x24x12:4", "chan_smag_wm:32x24x16:3"]))


EXE = os.path.join(ROOT, "cales_amd", "fortran", "cales")
EXE_SP = os.path.join(ROOT, "cales_amd", "fortran", "cales_sp")


@pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(EXE_SP)), reason="Fortran hosts not built (amdflang absent)")
def test_fortran_host_single_precision(tmp_path):
    """cales_sp = the Fortran host compiled with -D_SINGLE_PRECISION against libcales_hip_sp.so: its checkpoint holds (4 N + 2) reals of four
    bytes (load.f90:44-52 with rp = sp) and agrees with the FP64 host's within single precision."""
    from tests.test_gpu_fortran_host import _nml, _read_fld, _run
    text = _nml("chan_smag_wm", nstep=4, icheck=2, iout0d=2, iout1d=100000, iout2d=100000, iout3d=100000, isave=100000)
    text = re.sub(r"stop_type\(1:3\) = .*", "stop_type(1:3) = T, F, F", text)
    a, b = str(tmp_path / "dp"), str(tmp_path / "sp")
    _run(a, text); out = _run(b, text, cmd=[EXE_SP])
    assert "*** Fim ***" in out
    ng = tuple(int(x) for x in load_golden("chan_smag_wm")[1].ng)
    fd, td, _ = _read_fld(os.path.join(a, "fld.bin"), ng)
    raw = np.fromfile(os.path.join(b, "fld.bin"), dtype=np.float32)
    n = int(np.prod(ng))
    assert raw.size == 4 * n + 2 and int(round(raw[-1])) == 4
    assert abs(raw[-2] / td - 1.) < 1e-5
    for q in range(3):
        x = raw[q * n:(q + 1) * n].reshape(ng, order="F")
        assert np.abs(x - fd[q]).max() < 1e-5 * max(np.abs(f).max() for f in fd[:3]), q
