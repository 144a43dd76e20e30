"""Shared helpers for the tests: golden fixtures -> Case, comparison norms."""
import os

import numpy as np

from cales_amd.nml import parse_text

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RK = [(32. / 60., 0.), (25. / 60., -17. / 60.), (45. / 60., -25. / 60.)]      # reference src/param.f90:27-29
FULL_CASES = ["tgv_ppp", "tgv_dsmag_ppp", "chan_smag_wm", "chan_smag", "chan_dsmag", "chan_dsmag_wm", "duct_smag_wm", "duct_smag_wm_imp1d",
              "cavity_nnn", "devchan_nd", "halfchan_imp1d", "duct_dsmag_wm", "duct_dsmag", "cavity_dsmag"]


def load_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    case = parse_text(str(g["input_nml"]))
    case.impdiff = int(g["impdiff"])
    return g, case


def F(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


def relerr(a, b):
    """L-infinity error scaled by the field maximum."""
    a = np.asarray(a); b = np.asarray(b)
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale
