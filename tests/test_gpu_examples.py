"""Every example case the reference ships (examples/*/*/input.nml, collected as data by tests/golden/gen_examples.py) through
the device path with the grid shrunk: create, deterministic initial field, start-up, three time steps -- finite fields,
divergence at round-off after the projection, bulk velocity held where the case forces it."""
import json
import os

import numpy as np
import pytest

from cales_amd.nml import parse_text

pytestmark = pytest.mark.gpu
EXAMPLES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "examples.json")))


def _shrink(n):
    # keep the factors of the shipped size (144 = 2^4 3^2, 80 = 2^4 5, 72 = 2^3 3^2 ...) but stay small
    for d in (16, 8, 4, 2):
        if n % d == 0 and n // d >= 16:
            return n // d
    return n


@pytest.mark.parametrize("name", sorted(EXAMPLES))
def test_example_case_runs(name):
    from cales_amd.hotpath import HotPath, initflow
    case = parse_text(EXAMPLES[name])
    case.ng[:] = [_shrink(int(x)) for x in case.ng]
    if case.sgstype == "none" and case.cbcvel[0, 0, 0] != "P":
        case.cbcsgs[:, 0] = "D"                         # the two developing_* examples ship an entry sanity.f90:191-203 rejects
    if np.any(case.lwm != 0):
        case.hwm = max(float(case.hwm), 1.6 * float(case.l[2]) / int(case.ng[2]))      # wall-model height inside the coarser grid
    h = HotPath(case)
    u, v, w, p = initflow(case)
    rng = np.random.RandomState(1)
    for a in (u, v, w):
        a[1:-1, 1:-1, 1:-1] += 1e-3 * (rng.rand(*[int(x) for x in case.ng]) - 0.5)
    h.upload(u, v, w, p); h.startup()
    dt = 0.5 * h.chkdt()
    assert np.isfinite(dt) and dt > 0.
    for _ in range(3):
        h.step(dt)
    divtot, divmax = h.chkdiv()
    gu, gv, gw, gp, gvis = h.download()
    assert all(np.isfinite(a).all() for a in (gu, gv, gw, gp, gvis))
    scale = max(np.abs(gu).max(), np.abs(gv).max(), np.abs(gw).max(), 1e-30) * max(int(x) / float(l) for x, l in zip(case.ng, case.l))
    assert divmax < 1e-10 * scale, (divmax, scale)
    for d in range(3):
        if case.is_forced[d]:
            assert abs(h.bulk_mean("uvw"[d], "f" if d < 2 else "c") - float(case.velf[d])) < 1e-10
    h.close()
