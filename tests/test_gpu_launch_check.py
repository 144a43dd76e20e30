"""Every kernel launch of the library is checked (LAUNCH in cales_amd/csrc/common.hpp): an invalid launch configuration must come back as a
non-zero status with the kernel's name in cales_last_error, fail the context for good, and never leave a silently wrong field behind. The test
hook CALES_TEST_BAD_LAUNCH=<substring of a kernel name> gives the matching launches a block of 4096 threads, which no device accepts, so the
real error path (hipLaunchKernelGGL -> hipGetLastError) is exercised. The reference ignores its istat everywhere (src/solver_gpu.f90:80)."""
import numpy as np
import pytest

from tests.util import load_golden

pytestmark = pytest.mark.gpu


def _create(name, ng):
    from cales_amd.hotpath import HotPath, initflow
    g, case = load_golden(name)
    case.ng[:] = ng
    h = HotPath(case)
    h.upload(*initflow(case))
    return h


def _start(name, ng):
    h = _create(name, ng); h.startup()
    return h


@pytest.mark.parametrize("kernel,name", [("k_momrk", "chan_dsmag"), ("k_lmf_tile", "chan_dsmag"), ("k_fft_x8", "chan_smag"), ("k_gaussel_tile", "tgv_ppp"),
                                         ("k_correc_cell", "chan_smag")])
def test_failed_launch_inside_step_is_reported(kernel, name, monkeypatch):
    from cales_amd.hotpath import CalesError
    h0 = _start(name, (64, 16, 16)); dt = 0.5 * h0.chkdt(); h0.step(dt); good = h0.download(); h0.close()
    monkeypatch.setenv("CALES_TEST_BAD_LAUNCH", kernel)
    h = _create(name, (64, 16, 16))
    with pytest.raises(CalesError, match="kernel launch failed.*" + kernel):
        h.startup()      # (k_lmf_tile runs in startup's cmpt_sgs: the failure then surfaces there)
        h.step(dt)
    # the context is failed for good: every later entry refuses with the same message instead of handing out a half-stepped field
    for call in (h.chkdt, h.chkdiv, lambda: h.step(dt), h.download, lambda: h.bounduvw(True, False)):
        with pytest.raises(CalesError, match="kernel launch failed|intermediate state"):
            call()
    h.close()
    # and the hook is gone with the environment variable: a fresh context gives the good result again
    monkeypatch.delenv("CALES_TEST_BAD_LAUNCH")
    h1 = _start(name, (64, 16, 16)); h1.step(dt); again = h1.download(); h1.close()
    for a, b in zip(good, again):
        assert np.array_equal(a, b)


def test_failed_launch_at_operator_level(monkeypatch):
    from cales_amd.hotpath import CalesError
    monkeypatch.setenv("CALES_TEST_BAD_LAUNCH", "k_fillps")
    h = _start("chan_smag", (24, 20, 12))
    with pytest.raises(CalesError, match="kernel launch failed.*k_fillps"):
        h.fillps(1.0)
    with pytest.raises(CalesError, match="kernel launch failed"):
        h.solver()
    h.close()


def test_every_kernel_name_is_hooked(monkeypatch):
    """the hook matches on the kernel's name: with every launch made invalid ("k_") the first entry that launches anything -- set-up kernels of
    cales_create if there are any, otherwise the upload's repack kernel -- fails and says which kernel"""
    from cales_amd.hotpath import CalesError, HotPath, initflow
    monkeypatch.setenv("CALES_TEST_BAD_LAUNCH", "k_")
    g, case = load_golden("chan_smag"); case.ng[:] = (32, 16, 12)
    with pytest.raises(CalesError, match="kernel launch failed.*k_"):
        h = HotPath(case)
        h.upload(*initflow(case))
