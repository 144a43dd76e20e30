"""ctypes binding of libcales_hip.so (C-ABI declared in include/cales.h).

There is no CPU fallback: if the HIP library has not been built, importing the symbols raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Working precision rp of the reference (src/precision.f90:11-20), a build-time choice there (-D_SINGLE_PRECISION) and a process-wide one
# here: CALES_PRECISION=single loads libcales_hip_sp.so (the same sources built with -DCALES_SINGLE), and every real of the C-ABI --
# fields, case parameters, scalars -- is a float.
PRECISION = os.environ.get("CALES_PRECISION", "double")
if PRECISION not in ("double", "single"):
    raise ValueError("CALES_PRECISION must be 'double' or 'single'")
SINGLE = PRECISION == "single"
c_real = C.c_float if SINGLE else C.c_double
np_real = np.float32 if SINGLE else np.float64
LIB_PATH = os.environ.get("CALES_LIB", os.path.join(_HERE, "libcales_hip_sp.so" if SINGLE else "libcales_hip.so"))   # CALES_LIB: tuning builds only

SGS = {"none": 0, "smag": 1, "dsmag": 2}
FIELDS = dict(u=0, v=1, w=2, p=3, pp=4, visct=5, dudt=6, dvdt=7, dwdt=8, dudto=9, dvdto=10, dwdto=11,
              dudtd=12, dvdtd=13, dwdtd=14)

# every symbol include/cales.h declares (tests/test_capi_symbols.py checks the list against the header)
SYMBOLS = ["cales_real_size", "cales_initgrid", "cales_initflow", "cales_check_case", "cales_create", "cales_destroy", "cales_last_error",
           "cales_sync", "cales_local_size", "cales_upload_state", "cales_download_state", "cales_set_field",
           "cales_get_field", "cales_get_bcvel", "cales_bounduvw", "cales_boundp", "cales_mom", "cales_rk",
           "cales_bulk_forcing", "cales_get_forcing", "cales_bulk_mean", "cales_fillps", "cales_updt_rhs_b",
           "cales_solver", "cales_helmholtz_z", "cales_helmholtz", "cales_correc", "cales_updatep", "cales_cmpt_sgs", "cales_chkdt",
           "cales_chkdiv", "cales_out1d_single_point_chan", "cales_out1d_chan_budgets", "cales_out1d", "cales_out1d_chan", "cales_out2d_duct", "cales_step", "cales_get_dpdl", "cales_profile_enable", "cales_profile_reset",
           "cales_profile_count", "cales_profile_get", "cales_device_info", "cales_comm_buffer_doubles", "cales_set_comm",
           "cales_initflow_slab", "cales_comm_unique_id", "cales_comm_init_rccl", "cales_comm_selftest",
           "cales_device_count", "cales_set_device", "cales_set_comm_overlap", "cales_rk_par", "cales_describe_plan", "cales_calibrate"]


class CalesCase(C.Structure):
    """struct cales_case of include/cales.h."""
    _fields_ = [("ng", C.c_int32 * 3), ("l", c_real * 3), ("gtype", C.c_int32), ("gr", c_real),
                ("visci", c_real), ("cbcvel", C.c_char * 18), ("cbcpre", C.c_char * 6), ("cbcsgs", C.c_char * 6),
                ("bcvel", c_real * 18), ("bcpre", c_real * 6), ("bcsgs", c_real * 6),
                ("bforce", c_real * 3), ("is_forced", C.c_int32 * 3), ("velf", c_real * 3),
                ("sgstype", C.c_int32), ("lwm", C.c_int32 * 6), ("hwm", c_real), ("impdiff", C.c_int32),
                ("nranks", C.c_int32), ("rank", C.c_int32)]


def _chars(a) -> bytes:
    return "".join(np.asarray(a).ravel(order="F").tolist()).encode()


def make_case(case, nranks: int = 1, rank: int = 0) -> CalesCase:
    """cales_amd.nml.Case -> struct cales_case (Fortran storage order kept)."""
    p = CalesCase()
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
    if case.sgstype not in SGS:
        raise ValueError("ERROR: unknown SGS model" if case.sgstype != "amd" else "ERROR: AMD model not yet implemented")
    p.sgstype = SGS[case.sgstype]
    p.lwm[:] = [int(x) for x in case.lwm.ravel(order="F")]
    p.hwm = float(case.hwm)
    p.impdiff = int(case.impdiff)
    p.nranks, p.rank = int(nranks), int(rank)
    return p


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(make -C cales_amd/csrc). The CaLES hot path has no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime under the same soname (libamdhip64.so.7). A process
        # must hold ONE runtime: whichever is loaded first wins, and torch cannot initialise on top of the system
        # copy. So torch (when installed) is imported before libcales_hip.so is opened; set CALES_NO_TORCH=1 for a
        # torch-free process (e.g. alongside the Fortran driver).
        if os.environ.get("CALES_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.cales_real_size.restype = C.c_int; L.cales_real_size.argtypes = []
        if L.cales_real_size() != C.sizeof(c_real):
            raise RuntimeError(f"{LIB_PATH} is built for {8 * L.cales_real_size()}-bit reals, CALES_PRECISION={PRECISION}")
        L.cales_last_error.restype = C.c_char_p
        L.cales_last_error.argtypes = [C.c_void_p]
        L.cales_create.argtypes = [C.POINTER(CalesCase), C.c_void_p, C.POINTER(C.c_void_p)]
        dp = C.c_void_p
        for name, args in {
            "cales_destroy": [C.c_void_p], "cales_sync": [C.c_void_p], "cales_local_size": [C.c_void_p, dp, dp],
            "cales_upload_state": [C.c_void_p, dp, dp, dp, dp], "cales_download_state": [C.c_void_p, dp, dp, dp, dp, dp],
            "cales_set_field": [C.c_void_p, C.c_int, dp], "cales_get_field": [C.c_void_p, C.c_int, dp],
            "cales_get_bcvel": [C.c_void_p, C.c_int, dp, dp, dp],
            "cales_bounduvw": [C.c_void_p, C.c_int, C.c_int], "cales_boundp": [C.c_void_p, C.c_int, C.c_int],
            "cales_mom": [C.c_void_p], "cales_rk": [C.c_void_p, C.c_int, c_real], "cales_bulk_forcing": [C.c_void_p],
            "cales_get_forcing": [C.c_void_p, dp], "cales_bulk_mean": [C.c_void_p, C.c_int, C.c_int, dp],
            "cales_fillps": [C.c_void_p, c_real], "cales_updt_rhs_b": [C.c_void_p], "cales_solver": [C.c_void_p],
            "cales_helmholtz_z": [C.c_void_p, C.c_int, c_real], "cales_helmholtz": [C.c_void_p, C.c_int, c_real], "cales_correc": [C.c_void_p, c_real],
            "cales_updatep": [C.c_void_p, c_real], "cales_cmpt_sgs": [C.c_void_p], "cales_chkdt": [C.c_void_p, dp],
            "cales_chkdiv": [C.c_void_p, dp, dp], "cales_out1d_single_point_chan": [C.c_void_p, C.c_void_p], "cales_out1d_chan_budgets": [C.c_void_p, C.c_void_p, C.c_void_p], "cales_out1d": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], "cales_out1d_chan": [C.c_void_p, C.c_void_p], "cales_out2d_duct": [C.c_void_p, C.c_void_p], "cales_step": [C.c_void_p, c_real], "cales_get_dpdl": [C.c_void_p, dp],
            "cales_profile_enable": [C.c_void_p, C.c_int], "cales_profile_reset": [C.c_void_p],
            "cales_profile_count": [C.c_void_p], "cales_profile_get": [C.c_void_p, C.c_int, C.c_char_p, C.c_int, dp, dp],
            "cales_device_info": [C.c_void_p, C.c_char_p, C.c_int, dp],
            "cales_initgrid": [C.c_int, C.c_int, c_real, c_real, dp, dp, dp, dp],
            "cales_initflow": [C.POINTER(CalesCase), C.c_char_p, C.c_int, dp, dp, dp, dp],
            "cales_check_case": [C.POINTER(CalesCase), C.c_char_p, C.c_int],
            "cales_comm_buffer_doubles": [C.c_void_p, dp],
            "cales_set_comm": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64],
            "cales_initflow_slab": [C.POINTER(CalesCase), C.c_char_p, C.c_int, dp, dp, dp, dp],
            "cales_comm_unique_id": [C.c_void_p], "cales_comm_init_rccl": [C.c_void_p, C.c_void_p], "cales_comm_selftest": [],
            "cales_device_count": [C.POINTER(C.c_int)], "cales_set_device": [C.c_int],
            "cales_set_comm_overlap": [C.c_void_p, C.c_void_p, C.c_void_p],
            "cales_rk_par": [C.c_void_p, dp, c_real, dp],
            "cales_describe_plan": [C.c_void_p, C.c_char_p, C.c_int], "cales_calibrate": [C.c_void_p, C.c_int, dp, dp],
        }.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = None if name == "cales_destroy" else C.c_int
        _lib = L
    return _lib
