"""N > 1 path on CPU: world_size-2 (and 4) gloo process groups running the exchange class of the GPU path
(cales_amd.decomp.TorchComm) with numpy mirrors of the device layouts; see tests/_gloo_worker.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_exchange_and_distributed_solve(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + world), os.path.join(ROOT, "tests", "_gloo_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GLOO_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
