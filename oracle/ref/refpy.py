"""TEST INFRASTRUCTURE: ctypes front-end of oracle/_ref/libcales_ref*.so (the reference's own
compiled modules, see oracle/ref/Makefile). Exists only where /root/reference was present at
build time. One case per process: the reference keeps `save`d state (rk.f90:36-41, sgs.f90:51-57).
`Ref(...)` must be constructed with the current directory holding the case's input.nml.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.normpath(os.path.join(_HERE, "..", "_ref"))
VARIANT = {0: "libcales_ref.so", 1: "libcales_ref_imp.so", 2: "libcales_ref_imp1d.so"}


def available(impdiff: int = 0) -> bool:
    return os.path.exists(os.path.join(REFDIR, VARIANT[impdiff]))


def _p(a):
    assert a.dtype == np.float64 and (a.flags.f_contiguous or a.ndim == 1)
    return a.ctypes.data_as(C.c_void_p)


class Ref:
    def __init__(self, impdiff: int = 0):
        self.lib = C.CDLL(os.path.join(REFDIR, VARIANT[impdiff]))
        ist = C.c_int(0)
        self.lib.ref_init(C.byref(ist))
        iv = np.zeros(40, np.int32)
        self.lib.ref_get_ints(iv.ctypes.data_as(C.c_void_p))
        rv = np.zeros(64)
        self.lib.ref_get_reals(_p(rv))
        cv = C.create_string_buffer(230)
        self.lib.ref_get_chars(cv)
        self.ints, self.reals, self.chars = iv, rv, cv.raw
        self.n = tuple(int(x) for x in iv[:3])
        self.shape = tuple(x + 2 for x in self.n)

    def params(self) -> dict:
        iv, rv, cv = self.ints, self.reals, self.chars.decode()
        return dict(
            ng=iv[0:3].copy(), gtype=int(iv[3]), nstep=int(iv[4]), restart=bool(iv[5]), is_overwrite_save=bool(iv[6]),
            nsaves_max=int(iv[7]), icheck=int(iv[8]), iout0d=int(iv[9]), iout1d=int(iv[10]), iout2d=int(iv[11]),
            iout3d=int(iv[12]), isave=int(iv[13]), stop_type=iv[14:17].astype(bool), is_forced=iv[17:20].astype(bool),
            is_wallturb=bool(iv[20]), lwm=iv[23:29].reshape((2, 3), order="F").copy(),
            index_wm=iv[29:35].reshape((2, 3), order="F").copy(),
            l=rv[0:3].copy(), dl=rv[3:6].copy(), dli=rv[6:9].copy(), gr=rv[9], cfl=rv[10], dtmax=rv[11], dt_f=rv[12],
            visci=rv[13], visc=rv[14], time_max=rv[15], tw_max=rv[16], bforce=rv[17:20].copy(), velf=rv[20:23].copy(),
            hwm=rv[23], bcvel=rv[24:42].reshape((2, 3, 3), order="F").copy(), bcpre=rv[42:48].reshape((2, 3), order="F").copy(),
            bcsgs=rv[48:54].reshape((2, 3), order="F").copy(),
            cbcvel_after_initbc=np.array(list(cv[0:18]), dtype="U1").reshape((2, 3, 3), order="F"),
            cbcpre=np.array(list(cv[18:24]), dtype="U1").reshape((2, 3), order="F"),
            cbcsgs=np.array(list(cv[24:30]), dtype="U1").reshape((2, 3), order="F"),
            inivel=cv[30:130].strip(), sgstype=cv[130:230].strip())

    def zeros(self):
        return np.zeros(self.shape, order="F")

    def grid(self):
        g = [np.zeros(self.n[2] + 2) for _ in range(4)]
        self.lib.ref_get_grid(*[_p(a) for a in g])
        return dict(dzc=g[0], dzf=g[1], zc=g[2], zf=g[3])

    def rhsbp(self):
        n = self.n
        x = np.zeros((n[1], n[2], 2), order="F"); y = np.zeros((n[0], n[2], 2), order="F"); z = np.zeros((n[0], n[1], 2), order="F")
        self.lib.ref_get_rhsbp(_p(x), _p(y), _p(z))
        return x, y, z

    def bcvel_planes(self, ivel):
        n = self.n
        x = np.zeros((n[1] + 2, n[2] + 2, 2), order="F"); y = np.zeros((n[0] + 2, n[2] + 2, 2), order="F")
        z = np.zeros((n[0] + 2, n[1] + 2, 2), order="F")
        self.lib.ref_get_bcvel(int(ivel), _p(x), _p(y), _p(z))
        return x, y, z

    def initflow(self):
        u, v, w, p = (self.zeros() for _ in range(4))
        self.lib.ref_initflow(_p(u), _p(v), _p(w), _p(p))
        return u, v, w, p

    def bounduvw(self, u, v, w, is_updt_wm=True, is_correc=False):
        self.lib.ref_bounduvw(int(is_updt_wm), int(is_correc), _p(u), _p(v), _p(w))

    def boundp(self, p, which=0):
        self.lib.ref_boundp(int(which), _p(p))

    def mom(self, u, v, w, visct):
        out = [np.zeros(self.n, order="F") for _ in range(6)]
        self.lib.ref_mom(_p(u), _p(v), _p(w), _p(visct), *[_p(a) for a in out])
        return out

    def rk(self, irk, dt, p, visct, u, v, w):
        f = np.zeros(3)
        self.lib.ref_rk(int(irk), C.c_double(dt), _p(p), _p(visct), _p(u), _p(v), _p(w), _p(f))
        return f

    def bulk_forcing(self, f, u, v, w):
        self.lib.ref_bulk_forcing(_p(np.ascontiguousarray(f, dtype=np.float64)), _p(u), _p(v), _p(w))

    def bulk_mean(self, p, c_or_f="f"):
        m = C.c_double(0.)
        self.lib.ref_bulk_mean(1 if c_or_f == "f" else 0, _p(p), C.byref(m))
        return m.value

    def fillps(self, dtrki, u, v, w, pp):
        self.lib.ref_fillps(C.c_double(dtrki), _p(u), _p(v), _p(w), _p(pp))

    def updt_rhs_b_p(self, pp):
        self.lib.ref_updt_rhs_b_p(_p(pp))

    def updt_rhs_b_velz(self, ivel, alpha, q):
        self.lib.ref_updt_rhs_b_velz(int(ivel), C.c_double(alpha), _p(q))

    def correc(self, dtrk, pp, u, v, w):
        self.lib.ref_correc(C.c_double(dtrk), _p(pp), _p(u), _p(v), _p(w))

    def updatep(self, alpha, pp, p):
        self.lib.ref_updatep(C.c_double(alpha), _p(pp), _p(p))

    def cmpt_sgs(self, u, v, w, visct):
        self.lib.ref_cmpt_sgs(_p(u), _p(v), _p(w), _p(visct))

    def chkdt(self, visct, u, v, w):
        d = C.c_double(0.)
        self.lib.ref_chkdt(_p(visct), _p(u), _p(v), _p(w), C.byref(d))
        return d.value

    def chkdiv(self, u, v, w):
        a, b = C.c_double(0.), C.c_double(0.)
        self.lib.ref_chkdiv(_p(u), _p(v), _p(w), C.byref(a), C.byref(b))
        return a.value, b.value

    def out1d_single_point_chan(self, u, v, w, p, visct, fname="velstats_ref"):
        """the reference's plane statistics of the current directory's case: (27, n3), (38, n3), (6, n3) arrays read back from the
        .bin files the routine writes (src/output.f90:683-699, 990-1055)"""
        b = fname.encode()
        self.lib.ref_out1d_single_point_chan(b, C.c_int(len(b)), _p(u), _p(v), _p(w), _p(p), _p(visct))
        n3 = self.n[2]
        rd = lambda suffix, nv: np.fromfile(fname + suffix + ".bin").reshape((nv, n3), order="F")
        return rd("", 27), rd("_reystr_budget", 38), rd("_leakage", 6)

    def out1d(self, idir, p, use_dzc=False, fname="out1d_ref.out"):
        """the reference's out1d (src/output.f90:50-163): (coordinate, profile) as printed, 8 significant digits"""
        b = fname.encode()
        self.lib.ref_out1d(b, C.c_int(len(b)), C.c_int(int(idir)), C.c_int(int(use_dzc)), _p(p))
        t = np.loadtxt(fname, ndmin=2)
        return t[:, 0], t[:, 1]

    def out1d_chan(self, u, v, w, fname="out1d_chan_ref.out"):
        """out1d_chan (src/output.f90:317-405): the file's columns (z, um, vm, wm, u2, v2, w2, uw) as an (n3, 8) array"""
        b = fname.encode()
        self.lib.ref_out_uvw(C.c_int(0), b, C.c_int(len(b)), _p(u), _p(v), _p(w))
        return np.loadtxt(fname, ndmin=2)

    def out2d_duct(self, u, v, w, fname="out2d_duct_ref.out"):
        """out2d_duct (src/output.f90:406-507): the file's rows (y, z, um, vm, wm, u2, v2, w2, uv, uw, vw), j fastest, as an (n2 n3, 11) array"""
        b = fname.encode()
        self.lib.ref_out_uvw(C.c_int(1), b, C.c_int(len(b)), _p(u), _p(v), _p(w))
        return np.loadtxt(fname, ndmin=2)

    # ---- plain-arithmetic routines of initsolver.f90:66-169 and solver.f90:82-179
    def eigenvalues(self, n, cbc2, c_or_f):
        lam = np.zeros(int(n))
        self.lib.ref_eigenvalues(C.c_int(int(n)), "".join(cbc2).encode(), c_or_f.encode(), _p(lam))
        return lam

    def tridmatrix(self, cbc2, n, dzi, dzci, dzfi, c_or_f):
        a, b, c = (np.zeros(int(n)) for _ in range(3))
        self.lib.ref_tridmatrix("".join(cbc2).encode(), C.c_int(int(n)), C.c_double(dzi), _p(np.ascontiguousarray(dzci)), _p(np.ascontiguousarray(dzfi)),
                                c_or_f.encode(), _p(a), _p(b), _p(c))
        return a, b, c

    def gaussel(self, p, a, b, c, n, nh=0, lambdaxy=None, periodic=False):
        """in place on p (Fortran order, halo nh); n = unknowns per column"""
        nx, ny, nz = (s_ - 2 * nh for s_ in p.shape)
        lam = lambdaxy if lambdaxy is not None else np.zeros((nx, ny), order="F")
        self.lib.ref_gaussel(C.c_int(int(periodic)), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(int(n)), C.c_int(nh), _p(a), _p(b), _p(c), _p(p),
                             C.c_int(int(lambdaxy is not None)), _p(np.asfortranarray(lam)))

    def finalize(self):
        self.lib.ref_finalize()
