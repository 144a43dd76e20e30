import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(autouse=True)
def _imp3d_open_boundaries(request, monkeypatch):
    """3-D implicit diffusion with Neumann-Neumann pairs / non-zero velocity BC values in x and y is a SUPERSET of what the reference admits
    (sanity.f90:233-252); cales_check_case refuses it unless CALES_IMP3D_OPEN=1. The tests of that superset switch it on for themselves."""
    nm = request.node.name
    if "open" in nm or "devchan_imp3d" in nm or "openy" in nm:
        monkeypatch.setenv("CALES_IMP3D_OPEN", "1")
