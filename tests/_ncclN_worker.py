"""N ranks on N GPUs (torch.distributed.run, backend nccl = RCCL over xGMI): every rank steps its y-slab through the library's own
RCCL exchanges (or the torch.distributed callbacks with CALES_COMM=torch) and compares with the single-rank run of the same case,
computed by each rank on its own GPU. Launched by tests/test_gpu_decomp.py::test_n_rank_nccl_process_group on multi-GPU nodes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from cales_amd.decomp import SlabHotPath
from cales_amd.hotpath import HotPath, initflow
from tests.util import load_golden

local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)      # (two ranks on one device are refused by RCCL: "Duplicate GPU detected", tried on the one-GPU box)
torch.cuda.set_device(local)
BACKEND = os.environ.get("CALES_TEST_BACKEND", "nccl")      # gloo: several processes on ONE GPU, messages staged through the host (decomp.StagedGlooComm)
if BACKEND == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
else:
    dist.init_process_group("gloo")
P, r = dist.get_world_size(), dist.get_rank()
# (the static-Smagorinsky duct runs on any number of slabs: the shear planes of the y walls travel to every rank, k_sgs.hip wall_shear_y_planes;
#  the reference stops at two subdomains between two opposite walls, sanity.f90:98-111)
for name, ng in (("chan_dsmag", (64, 16 * P, 24)), ("duct_smag_wm", (32, 16 * P, 16)), ("duct_smag_wm_imp1d", (32, 8 * P, 16)), ("tgv_dsmag_ppp", (32, 8 * P, 16)),
                 ("cavity_nnn", (32, 8 * P, 12)), ("duct_smag", (64, 8 * P, 16))):
    _, case = load_golden(name)
    case.ng[:] = ng
    if P >= 8 and name.startswith("duct_smag_wm"):
        case.hwm = 0.2      # the sampling height must lie inside the slab that owns the wall (sanity.f90:224-231): l(2)/8 = 0.25 is its upper bound
    if case.sgstype == "none" and case.cbcvel[0, 0, 0] != "P":
        case.cbcsgs[:, 0] = "D"
    u, v, w, p = initflow(case)
    ref = HotPath(case); ref.upload(u, v, w, p); ref.startup(); dt = 0.5 * ref.chkdt()
    for _ in range(2):
        ref.step(dt)
    a = ref.download(); ref.close()
    h = SlabHotPath(case, dist, torch)
    assert h.native == (BACKEND == "nccl" and os.environ.get("CALES_COMM", "rccl") == "rccl"), "unexpected exchange layer"
    h.upload_initial(); h.startup()
    assert abs(0.5 * h.chkdt() / dt - 1) < 1e-12
    for _ in range(2):
        h.step(dt)
    b = h.download(); div = h.chkdiv(); j0 = h.lo[1] - 1; n2 = h.n[1]
    h.close()
    for x, y, nm in zip(a[:3] + [a[4]], b[:3] + [b[4]], ("u", "v", "w", "visct")):
        err = np.abs(y[:, 1:-1, :] - x[:, j0 + 1:j0 + n2 + 1, :]).max() / max(np.abs(x).max(), 1e-300)
        assert err < 1e-10, (name, nm, err)
    assert div[1] < 1e-11, (name, div)
    if r == 0:
        print(f"{name} {ng}: ok", flush=True)
dist.barrier()
dist.destroy_process_group()
if r == 0:
    print("NCCLN OK")
