"""Host-side mirror of the reference operator interface for the per-step hot path.

`HotPath` owns one `cales_ctx` (one GPU, one HIP stream) and exposes the routines the reference driver
calls inside its time loop, under their reference names and argument meaning
(src/main.f90:417-507: rk, bulk_forcing, bounduvw, fillps, updt_rhs_b, solver, boundp, correc, updatep,
cmpt_sgs; src/main.f90:523-537: chkdt, chkdiv). All arithmetic runs in libcales_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np

from . import capi
from .nml import Case

RKCOEFF = ((32. / 60., 0.), (25. / 60., -17. / 60.), (45. / 60., -25. / 60.))   # src/param.f90:27-29
SMALL = np.finfo(capi.np_real).eps * 10 ** ((6 if capi.SINGLE else 15) // 2)     # src/param.f90:24: epsilon(1._rp)*10**(precision(1._rp)/2)


class CalesError(RuntimeError):
    pass


REAL = capi.np_real      # numpy dtype of rp: float64, or float32 with CALES_PRECISION=single (capi.py)


def _p(a: np.ndarray):
    if a.dtype != REAL or not (a.flags.f_contiguous or a.ndim == 1):
        raise ValueError(f"fields must be Fortran-ordered {np.dtype(REAL).name} arrays")
    return a.ctypes.data_as(C.c_void_p)


class _In:
    """An input array in the library's precision (a converted copy when the caller's dtype differs, e.g. float64 fields handed to the
    single-precision build); keeps the copy alive for the duration of the call."""
    def __init__(self, a):
        a = np.asarray(a)
        self.a = a if a.dtype == REAL and (a.flags.f_contiguous or a.ndim == 1) else np.asfortranarray(a, dtype=REAL)
        self.p = self.a.ctypes.data_as(C.c_void_p)


def initgrid(gtype: int, n: int, gr: float, lz: float) -> Dict[str, np.ndarray]:
    """src/initgrid.f90:15 (host helper of the library)."""
    out = [np.zeros(n + 2, dtype=REAL) for _ in range(4)]
    rc = capi.lib().cales_initgrid(int(gtype), int(n), float(gr), float(lz), *[_p(a) for a in out])
    if rc:
        raise CalesError("cales_initgrid failed")
    return dict(dzc=out[0], dzf=out[1], zc=out[2], zf=out[3])


def initflow(case: Case) -> Tuple[np.ndarray, ...]:
    """src/initflow.f90:17, deterministic kinds; returns global haloed u,v,w,p."""
    cs = capi.make_case(case)
    shape = tuple(int(x) + 2 for x in case.ng)
    u, v, w, p = (np.zeros(shape, order="F", dtype=REAL) for _ in range(4))
    rc = capi.lib().cales_initflow(C.byref(cs), case.inivel.encode(), int(case.is_wallturb), _p(u), _p(v), _p(w), _p(p))
    if rc == 2:
        raise CalesError(f"inivel='{case.inivel}' relies on the Fortran RNG stream and is not offered")
    if rc:
        raise CalesError("ERROR: invalid name for initial velocity field")
    return u, v, w, p


def check_case(case: Case, nranks: int = 1) -> None:
    """The a-priori input checks of src/sanity.f90:33-67; raises like the reference aborts."""
    cs = capi.make_case(case, nranks, 0)
    buf = C.create_string_buffer(512)
    if capi.lib().cales_check_case(C.byref(cs), buf, 512):
        raise CalesError("*** Simulation aborted due to errors in the input file *** " + buf.value.decode())


class HotPath:
    def __init__(self, case: Case, nranks: int = 1, rank: int = 0, stream: int | None = None):
        self.case = case
        self.L = capi.lib()
        self.cs = capi.make_case(case, nranks, rank)
        h = C.c_void_p()
        rc = self.L.cales_create(C.byref(self.cs), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc:
            raise CalesError(f"cales_create failed ({rc}): {self.L.cales_last_error(None).decode()}")
        self.h = h
        n = (C.c_int32 * 3)(); lo = (C.c_int32 * 3)()
        self.L.cales_local_size(self.h, n, lo)
        self.n = tuple(n); self.lo = tuple(lo)
        self.shape = tuple(x + 2 for x in self.n)

    # -- plumbing
    def _chk(self, rc: int):
        if rc:
            raise CalesError(self.L.cales_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.cales_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def zeros(self) -> np.ndarray:
        return np.zeros(self.shape, order="F", dtype=REAL)

    def set(self, name: str, a: np.ndarray):
        a = _In(a)
        self._chk(self.L.cales_set_field(self.h, capi.FIELDS[name], a.p))

    def get(self, name: str) -> np.ndarray:
        a = self.zeros()
        self._chk(self.L.cales_get_field(self.h, capi.FIELDS[name], _p(a)))
        return a

    def upload(self, u, v, w, p):
        a = [_In(x) for x in (u, v, w, p)]
        self._chk(self.L.cales_upload_state(self.h, *[x.p for x in a]))

    def download(self):
        out = [self.zeros() for _ in range(5)]
        self._chk(self.L.cales_download_state(self.h, *[_p(a) for a in out]))
        return out

    def bcvel_planes(self, ivel: int):
        n = self.n
        x = np.zeros((n[1] + 2, n[2] + 2, 2), order="F", dtype=REAL); y = np.zeros((n[0] + 2, n[2] + 2, 2), order="F", dtype=REAL)
        z = np.zeros((n[0] + 2, n[1] + 2, 2), order="F", dtype=REAL)
        self._chk(self.L.cales_get_bcvel(self.h, ivel, _p(x), _p(y), _p(z)))
        return x, y, z

    def sync(self):
        self._chk(self.L.cales_sync(self.h))

    # -- operators (reference names)
    def bounduvw(self, is_updt_wm=True, is_correc=False):
        self._chk(self.L.cales_bounduvw(self.h, int(is_updt_wm), int(is_correc)))

    def boundp(self, field="p", which=0):
        self._chk(self.L.cales_boundp(self.h, capi.FIELDS[field], int(which)))

    def mom(self):
        self._chk(self.L.cales_mom(self.h))

    def rk(self, irk: int, dt: float) -> np.ndarray:
        self._chk(self.L.cales_rk(self.h, int(irk), float(dt)))
        f = np.zeros(3, dtype=REAL)
        self._chk(self.L.cales_get_forcing(self.h, _p(f)))
        return f

    def rk_par(self, rkpar, dt: float) -> np.ndarray:
        """rk(rkpar, ..., dt, ..., f) with the caller's coefficients (src/rk.f90:17)"""
        rp = np.ascontiguousarray(rkpar, dtype=REAL); f = np.zeros(3, dtype=REAL)
        self._chk(self.L.cales_rk_par(self.h, _p(rp), float(dt), _p(f)))
        return f

    def bulk_forcing(self):
        self._chk(self.L.cales_bulk_forcing(self.h))

    def bulk_mean(self, field="u", c_or_f="f") -> float:
        m = capi.c_real(0.)
        self._chk(self.L.cales_bulk_mean(self.h, capi.FIELDS[field], 1 if c_or_f == "f" else 0, C.byref(m)))
        return m.value

    def fillps(self, dtrki: float):
        self._chk(self.L.cales_fillps(self.h, float(dtrki)))

    def updt_rhs_b(self):
        self._chk(self.L.cales_updt_rhs_b(self.h))

    def solver(self):
        self._chk(self.L.cales_solver(self.h))

    def helmholtz_z(self, ivel: int, alpha: float):
        self._chk(self.L.cales_helmholtz_z(self.h, int(ivel), float(alpha)))

    def helmholtz(self, ivel: int, alpha: float):
        """3-D implicit diffusion of one velocity component (impdiff = 1; main.f90:423-491)."""
        self._chk(self.L.cales_helmholtz(self.h, int(ivel), float(alpha)))

    def correc(self, dtrk: float):
        self._chk(self.L.cales_correc(self.h, float(dtrk)))

    def updatep(self, alpha: float = 0.):
        self._chk(self.L.cales_updatep(self.h, float(alpha)))

    def cmpt_sgs(self):
        self._chk(self.L.cales_cmpt_sgs(self.h))

    def chkdt(self) -> float:
        d = capi.c_real(0.)
        self._chk(self.L.cales_chkdt(self.h, C.byref(d)))
        return d.value

    def chkdiv(self) -> Tuple[float, float]:
        a, b = capi.c_real(0.), capi.c_real(0.)
        self._chk(self.L.cales_chkdiv(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats_chan(self) -> np.ndarray:
        """First block of out1d_single_point_chan (src/output.f90:509-700): 27 plane statistics per z plane, shape (27, n3)."""
        buf = np.zeros((27, self.n[2]), order="F", dtype=REAL)
        self._chk(self.L.cales_out1d_single_point_chan(self.h, _p(buf)))
        return buf

    def out1d(self, field: str, idir: int, use_dzc: bool = False) -> np.ndarray:
        """out1d (src/output.f90:50-163): profile of `field` along idir averaged over the other two directions (this rank's sums)."""
        buf = np.zeros(self.n[idir - 1], dtype=capi.np_real)
        self._chk(self.L.cales_out1d(self.h, capi.FIELDS[field], int(idir), int(use_dzc), _p(buf)))
        return buf

    def out1d_chan(self) -> np.ndarray:
        """out1d_chan (src/output.f90:317-405): (7, n3) um, vm, wm, u2, v2, w2, uw per plane."""
        buf = np.zeros((7, self.n[2]), dtype=capi.np_real, order="F")
        self._chk(self.L.cales_out1d_chan(self.h, _p(buf)))
        return buf

    def out2d_duct(self) -> np.ndarray:
        """out2d_duct (src/output.f90:406-507): (9, n2, n3) um, vm, wm, u2, v2, w2, uv, uw, vw at the cell centres of every (j, k)."""
        buf = np.zeros((9, self.n[1], self.n[2]), dtype=capi.np_real, order="F")
        self._chk(self.L.cales_out2d_duct(self.h, _p(buf)))
        return buf

    def stats_chan_budgets(self):
        """Second and third block of out1d_single_point_chan (src/output.f90:700-1055): (38, n3) budget sums and (6, n3) divergence measures."""
        b = np.zeros((38, self.n[2]), order="F", dtype=REAL); l = np.zeros((6, self.n[2]), order="F", dtype=REAL)
        self._chk(self.L.cales_out1d_chan_budgets(self.h, _p(b), _p(l)))
        return b, l

    def step(self, dt: float):
        """Three RK substeps, src/main.f90:417-508, queued without host synchronisation."""
        self._chk(self.L.cales_step(self.h, float(dt)))

    def dpdl(self) -> np.ndarray:
        d = np.zeros(3, dtype=REAL)
        self._chk(self.L.cales_get_dpdl(self.h, _p(d)))
        return d

    def startup(self):
        """src/main.f90:370-375: ghost cells and eddy viscosity of the initial state."""
        self.bounduvw(True, False); self.boundp("p", 0); self.cmpt_sgs(); self.boundp("visct", 1)

    def describe_plan(self) -> Dict[str, str]:
        """The path the next cales_step takes (struct StepPlan): {'projection': 'in_strain_rate_pass', 'fillps': 'in_x_transform', ...}."""
        buf = C.create_string_buffer(2048)
        rc = self.L.cales_describe_plan(self.h, buf, 2048)
        if rc not in (0, 2):
            self._chk(rc)
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if "=" in kv)

    def calibrate(self, reps: int = 3) -> Dict[str, float]:
        """Read-only / write-only / copy streams over this context's own fields (cales_calibrate): GB/s on THIS box."""
        g = np.zeros(3, dtype=REAL); nb = C.c_int64(0)
        self._chk(self.L.cales_calibrate(self.h, int(reps), _p(g), C.byref(nb)))
        return {"read_GBps": float(g[0]), "write_GBps": float(g[1]), "copy_GBps": float(g[2]), "bytes_per_stream": int(nb.value), "launches": int(reps)}

    # -- measurement
    def profile(self, on: bool):
        self._chk(self.L.cales_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._chk(self.L.cales_profile_reset(self.h))

    def profile_stats(self) -> Dict[str, Tuple[int, float]]:
        out = {}
        for i in range(self.L.cales_profile_count(self.h)):
            name = C.create_string_buffer(64); calls = C.c_int64(0); ms = capi.c_real(0.)
            self.L.cales_profile_get(self.h, i, name, 64, C.byref(calls), C.byref(ms))
            out[name.value.decode()] = (calls.value, ms.value)
        return out

    def device_info(self) -> Tuple[str, int]:
        name = C.create_string_buffer(128); b = C.c_int64(0)
        self._chk(self.L.cales_device_info(self.h, name, 128, C.byref(b)))
        return name.value.decode(), b.value
