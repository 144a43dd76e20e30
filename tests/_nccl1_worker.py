"""One-rank process group on the GPU (backend nccl = RCCL): SlabHotPath joins the library's native communicator through the
same code path as an N-rank run (token broadcast, cales_comm_init_rccl) and must reproduce the plain single-GPU step."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

from cales_amd.decomp import SlabHotPath
from cales_amd.hotpath import HotPath, initflow
from tests.util import load_golden

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
_, case = load_golden("chan_dsmag")
case.ng[:] = (32, 16, 16)
u, v, w, p = initflow(case)
ref = HotPath(case); ref.upload(u, v, w, p); ref.startup(); dt = 0.5 * ref.chkdt(); ref.step(dt); a = ref.download(); ref.close()
h = SlabHotPath(case, dist, torch)
assert h.native == (os.environ.get("CALES_COMM", "rccl") == "rccl"), "unexpected exchange layer"
h.upload_initial(); h.startup(); assert abs(0.5 * h.chkdt() / dt - 1) < 1e-14; h.step(dt); b = h.download(); h.close()
for x, y in zip(a, b):
    assert np.array_equal(x, y)
dist.destroy_process_group()
print("NCCL1 OK")
