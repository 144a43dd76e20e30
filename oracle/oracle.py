"""TEST INFRASTRUCTURE: ctypes front-end of the CPU restatement (oracle/cales_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Fields are numpy float64 arrays of shape (n1+2, n2+2, n3+2), Fortran order.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcales_oracle.so")

SGS = {"none": 0, "smag": 1, "dsmag": 2}


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("cales_oracle.c", "cales_oracle.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcales_oracle.so"])
    return _LIB


class OParams(C.Structure):
    _fields_ = [("ng", C.c_int * 3), ("l", C.c_double * 3), ("gtype", C.c_int), ("gr", C.c_double),
                ("visci", C.c_double), ("cbcvel", C.c_char * 18), ("cbcpre", C.c_char * 6),
                ("cbcsgs", C.c_char * 6), ("bcvel", C.c_double * 18), ("bcpre", C.c_double * 6),
                ("bcsgs", C.c_double * 6), ("bforce", C.c_double * 3), ("is_forced", C.c_int * 3),
                ("velf", C.c_double * 3), ("sgstype", C.c_int), ("lwm", C.c_int * 6), ("hwm", C.c_double),
                ("impdiff", C.c_int), ("nthreads", C.c_int)]


def _chars(a) -> bytes:
    return "".join(np.asarray(a).ravel(order="F").tolist()).encode()


def _p(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.f_contiguous, "expect Fortran-ordered float64"
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, case, nthreads: int = 1, team_sums: bool = False):
        self.lib = C.CDLL(build())
        L = self.lib
        L.o_create.restype = C.c_void_p
        L.o_bulk_mean.restype = C.c_double
        L.o_chkdt.restype = C.c_double
        L.o_initflow.restype = C.c_int
        self.case = case
        p = OParams()
        p.ng[:] = [int(x) for x in case.ng]
        p.l[:] = [float(x) for x in case.l]
        p.gtype, p.gr, p.visci = int(case.gtype), float(case.gr), float(case.visci)
        p.cbcvel, p.cbcpre, p.cbcsgs = _chars(case.cbcvel), _chars(case.cbcpre), _chars(case.cbcsgs)
        p.bcvel[:] = case.bcvel.ravel(order="F").tolist()
        p.bcpre[:] = case.bcpre.ravel(order="F").tolist()
        p.bcsgs[:] = case.bcsgs.ravel(order="F").tolist()
        p.bforce[:] = case.bforce.tolist()
        p.is_forced[:] = [int(x) for x in case.is_forced]
        p.velf[:] = case.velf.tolist()
        p.sgstype = SGS[case.sgstype]
        p.lwm[:] = [int(x) for x in case.lwm.ravel(order="F")]
        p.hwm = float(case.hwm)
        p.impdiff = int(case.impdiff)
        p.nthreads = int(nthreads)
        self.params = p
        self.h = C.c_void_p(L.o_create(C.byref(p)))
        if team_sums:      # timing runs: sums over all cells by planes over the team instead of cell by cell on one thread (cales_oracle.h)
            L.o_set_team_sums(self.h, 1)
        self.n = tuple(int(x) for x in case.ng)
        self.shape = tuple(x + 2 for x in self.n)

    def close(self):
        if self.h:
            self.lib.o_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def zeros(self) -> np.ndarray:
        a = np.zeros(self.shape, dtype=np.float64, order="F")
        if self.params.nthreads > 1 and a.nbytes > (1 << 26):      # large field, several threads: pages placed by the threads that will use them
            self.lib.o_first_touch(self.h, _p(a))
        return a

    def grid(self):
        n3 = self.n[2]
        g = [np.zeros(n3 + 2) for _ in range(4)]
        self.lib.o_get_grid(self.h, *[_p(a) for a in g])
        return dict(dzc=g[0], dzf=g[1], zc=g[2], zf=g[3])

    def index_wm(self) -> np.ndarray:
        iw = np.zeros(6, dtype=np.int32)
        self.lib.o_get_index_wm(self.h, iw.ctypes.data_as(C.c_void_p))
        return iw.reshape((2, 3), order="F")

    def cbcvel(self) -> np.ndarray:
        b = C.create_string_buffer(18)
        self.lib.o_get_cbcvel(self.h, b)
        return np.array(list(b.raw.decode()), dtype="U1").reshape((2, 3, 3), order="F")

    def rhsbp(self):
        n = self.n
        x = np.zeros((n[1], n[2], 2), order="F"); y = np.zeros((n[0], n[2], 2), order="F"); z = np.zeros((n[0], n[1], 2), order="F")
        self.lib.o_get_rhsbp(self.h, _p(x), _p(y), _p(z))
        return x, y, z

    def bcvel_planes(self, ivel: int):
        n = self.n
        x = np.zeros((n[1] + 2, n[2] + 2, 2), order="F"); y = np.zeros((n[0] + 2, n[2] + 2, 2), order="F")
        z = np.zeros((n[0] + 2, n[1] + 2, 2), order="F")
        self.lib.o_get_bcvel(self.h, int(ivel), _p(x), _p(y), _p(z))
        return x, y, z

    def solver_operands(self, which: int = 0):
        n = self.n
        lam = np.zeros((n[0], n[1]), order="F"); a = np.zeros(n[2]); b = np.zeros(n[2]); c = np.zeros(n[2])
        nf = C.c_double(0.)
        self.lib.o_get_solver(self.h, which, _p(lam), _p(a), _p(b), _p(c), C.byref(nf))
        return lam, a, b, c, nf.value

    # ---- operators
    def initflow(self, inivel: str, is_wallturb: bool):
        u, v, w, p = (self.zeros() for _ in range(4))
        rc = self.lib.o_initflow(self.h, inivel.encode(), int(is_wallturb), _p(u), _p(v), _p(w), _p(p))
        if rc:
            raise ValueError(f"initial field {inivel!r} is not restated in the oracle")
        return u, v, w, p

    def bounduvw(self, u, v, w, is_updt_wm=True, is_correc=False):
        self.lib.o_bounduvw(self.h, int(is_updt_wm), int(is_correc), _p(u), _p(v), _p(w))

    def boundp(self, p, which=0):
        self.lib.o_boundp(self.h, int(which), _p(p))

    def mom(self, u, v, w, visct):
        n = self.n
        out = [np.zeros(n, order="F") for _ in range(6)]
        self.lib.o_mom(self.h, _p(u), _p(v), _p(w), _p(visct), *[_p(a) for a in out])
        return out

    def rk(self, irk, dt, p, visct, u, v, w):
        f = np.zeros(3)
        self.lib.o_rk(self.h, int(irk), C.c_double(dt), _p(p), _p(visct), _p(u), _p(v), _p(w), _p(f))
        return f

    def bulk_forcing(self, f, u, v, w):
        f = np.ascontiguousarray(f, dtype=np.float64)
        self.lib.o_bulk_forcing(self.h, _p(f), _p(u), _p(v), _p(w))

    def bulk_mean(self, p, c_or_f="f") -> float:
        return self.lib.o_bulk_mean(self.h, 1 if c_or_f == "f" else 0, _p(p))

    def fillps(self, dtrki, u, v, w, pp):
        self.lib.o_fillps(self.h, C.c_double(dtrki), _p(u), _p(v), _p(w), _p(pp))

    def updt_rhs_b_p(self, pp):
        self.lib.o_updt_rhs_b_p(self.h, _p(pp))

    def updt_rhs_b_velz(self, ivel, alpha, q):
        self.lib.o_updt_rhs_b_velz(self.h, int(ivel), C.c_double(alpha), _p(q))

    def updt_rhs_b_vel(self, ivel, alpha, q):
        """Boundary terms of all three directions for the 3-D implicit step (main.f90:424-431)."""
        self.lib.o_updt_rhs_b_vel(self.h, int(ivel), C.c_double(alpha), _p(q))

    def solver(self, pp):
        self.lib.o_solver(self.h, _p(pp))

    def solver_zsweep(self, pp):
        self.lib.o_solver_zsweep(self.h, _p(pp))

    def solver_gaussel_z(self, ivel, alpha, q):
        self.lib.o_solver_gaussel_z(self.h, int(ivel), C.c_double(alpha), _p(q))

    def solver_helmholtz(self, ivel, alpha, q):
        """(1 + alpha L) q = q* of main.f90:423-491 (3-D implicit diffusion) with the component's own transform kinds."""
        if self.lib.o_solver_helmholtz(self.h, int(ivel), C.c_double(alpha), _p(q)):
            raise ValueError("o_solver_helmholtz failed")

    def correc(self, dtrk, pp, u, v, w):
        self.lib.o_correc(self.h, C.c_double(dtrk), _p(pp), _p(u), _p(v), _p(w))

    def updatep(self, alpha, pp, p):
        self.lib.o_updatep(self.h, C.c_double(alpha), _p(pp), _p(p))

    def cmpt_sgs(self, u, v, w, visct):
        self.lib.o_cmpt_sgs(self.h, _p(u), _p(v), _p(w), _p(visct))

    def chkdt(self, visct, u, v, w) -> float:
        return self.lib.o_chkdt(self.h, _p(visct), _p(u), _p(v), _p(w))

    def chkdiv(self, u, v, w):
        a, b = C.c_double(0.), C.c_double(0.)
        self.lib.o_chkdiv(self.h, _p(u), _p(v), _p(w), C.byref(a), C.byref(b))
        return a.value, b.value

    def stats_chan(self, u, v, w, p, visct) -> np.ndarray:
        """first block of out1d_single_point_chan (output.f90:509-700): (27, n3) plane statistics"""
        buf = np.zeros((27, self.n[2]), order="F")
        self.lib.o_stats_chan(self.h, _p(u), _p(v), _p(w), _p(p), _p(visct), _p(buf))
        return buf

    def out1d(self, idir: int, p, use_dzc: bool = False) -> np.ndarray:
        """out1d (output.f90:50-163): profile of p along idir averaged over the other two directions"""
        buf = np.zeros(self.n[idir - 1])
        self.lib.o_out1d(self.h, int(idir), int(use_dzc), _p(p), _p(buf))
        return buf

    def out1d_chan(self, u, v, w) -> np.ndarray:
        """out1d_chan (output.f90:317-405): (7, n3) um, vm, wm, u2, v2, w2, uw"""
        buf = np.zeros((7, self.n[2]), order="F")
        self.lib.o_out1d_chan(self.h, _p(u), _p(v), _p(w), _p(buf))
        return buf

    def out2d_duct(self, u, v, w) -> np.ndarray:
        """out2d_duct (output.f90:406-507): (9, n2, n3) um, vm, wm, u2, v2, w2, uv, uw, vw at the cell centres of every (j, k)"""
        buf = np.zeros((9, self.n[1], self.n[2]), order="F")
        self.lib.o_out2d_duct(self.h, _p(u), _p(v), _p(w), _p(buf))
        return buf

    def step(self, dt, u, v, w, p, pp, visct):
        dpdl = np.zeros(3)
        self.lib.o_step(self.h, C.c_double(dt), _p(u), _p(v), _p(w), _p(p), _p(pp), _p(visct), _p(dpdl))
        return dpdl

    def r2r(self, kind: int, x: np.ndarray) -> np.ndarray:
        y = np.array(x, dtype=np.float64, copy=True)
        self.lib.o_r2r(int(kind), int(y.size), _p(np.asfortranarray(y)) if False else y.ctypes.data_as(C.c_void_p), 1)
        return y


R2R = dict(R2HC=0, HC2R=1, REDFT00=3, REDFT01=4, REDFT10=5, REDFT11=6, RODFT00=7, RODFT01=8, RODFT10=9, RODFT11=10)
