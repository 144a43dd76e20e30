import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "imp3d_open: the test runs 3-D implicit diffusion with open boundaries in x/y (CALES_IMP3D_OPEN=1 is set for it)")


IMP3D_OPEN_CASES = {"devchan_imp3d", "openy_imp3d"}      # parametrised case names that need the superset below


@pytest.fixture(autouse=True)
def _imp3d_open_boundaries(request, monkeypatch):
    """3-D implicit diffusion with Neumann-Neumann pairs / non-zero velocity BC values in x and y is a SUPERSET of what the reference admits
    (sanity.f90:233-252); cales_check_case refuses it unless CALES_IMP3D_OPEN=1. Only tests that say so get the switch: the explicit marker
    `imp3d_open`, or a parametrised case `name` from IMP3D_OPEN_CASES -- never a substring of the test's name. The refusal itself is tested in
    tests/test_host_side.py."""
    cs = getattr(request.node, "callspec", None)
    if request.node.get_closest_marker("imp3d_open") or (cs is not None and cs.params.get("name") in IMP3D_OPEN_CASES):
        monkeypatch.setenv("CALES_IMP3D_OPEN", "1")
