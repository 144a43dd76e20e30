"""Worker of tests/test_decomp_gloo.py: one of WORLD_SIZE CPU ranks (gloo). Runs the SAME exchange class the GPU
path uses (cales_amd.decomp.TorchComm) on CPU tensors, with numpy mirrors of the device pack/unpack layouts, and
checks against single-rank results computed by the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CALES_NO_TORCH"] = "1"

from cales_amd.decomp import TorchComm, mode_block_width, slab_rows, y_neighbours  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.util import F, load_golden  # noqa: E402


def thomas(a, b, c, lam, rhs):
    """reference src/solver.f90:153-179 with eps on the pivots; rhs[..., k]"""
    eps = np.finfo(float).eps
    n = rhs.shape[-1]
    p = rhs.copy(); d = np.zeros(rhs.shape[:-1] + (n,))
    z = 1. / (b[0] + lam + eps); d[..., 0] = c[0] * z; p[..., 0] *= z
    for k in range(1, n):
        z = 1. / (b[k] + lam - a[k] * d[..., k - 1] + eps)
        d[..., k] = c[k] * z
        p[..., k] = (p[..., k] - a[k] * p[..., k - 1]) * z
    for k in range(n - 2, -1, -1):
        p[..., k] -= d[..., k] * p[..., k + 1]
    return p


def main():
    dist.init_process_group("gloo")
    P, r = dist.get_world_size(), dist.get_rank()
    g, case = load_golden("chan_smag")
    case.ng[:] = (12, 8 * P, 6)
    n1, n2g, n3 = (int(x) for x in case.ng)
    n2l = n2g // P; lo, hi = slab_rows(n2g, P, r)
    cw = mode_block_width(n1, P); mh = n1 // 2 + 1
    nbuf = max(2 * P * n3 * n2l * cw, 4 * 3 * (n1 + 2) * (n3 + 2)) + 64
    A = torch.zeros(nbuf, dtype=torch.float64); B = torch.zeros(nbuf, dtype=torch.float64)
    comm = TorchComm(dist, torch, A, B, periodic_y=True)
    rng = np.random.RandomState(5)
    glob = rng.rand(n1 + 2, n2g + 2, n3 + 2)
    glob[:, 0, :] = glob[:, n2g, :]; glob[:, n2g + 1, :] = glob[:, 1, :]          # periodic y ghosts

    # ---- halo exchange (layout of k_pack_y / k_unpack_y: [field][k][i])
    loc = glob[:, lo - 1:hi + 2, :].copy(); loc[:, 0, :] = -1; loc[:, -1, :] = -1
    plane = (n1 + 2) * (n3 + 2)
    A[:plane] = torch.from_numpy(loc[:, 1, :].T.reshape(-1).copy()); A[plane:2 * plane] = torch.from_numpy(loc[:, n2l, :].T.reshape(-1).copy())
    assert comm.halo(0, plane, 0, plane, plane) == 0
    loc[:, 0, :] = B[:plane].numpy().reshape(n3 + 2, n1 + 2).T; loc[:, n2l + 1, :] = B[plane:2 * plane].numpy().reshape(n3 + 2, n1 + 2).T
    assert np.array_equal(loc, glob[:, lo - 1:hi + 2, :]), "halo"
    assert y_neighbours(0, P, False)[0] is None and y_neighbours(P - 1, P, False)[1] is None

    # ---- all-reduce
    A[nbuf - 8:nbuf - 5] = torch.tensor([r + 1., -r, 2. * r])
    comm.allreduce(nbuf - 8, 1, 0); comm.allreduce(nbuf - 7, 1, 1); comm.allreduce(nbuf - 6, 1, 2)
    assert A[nbuf - 8].item() == P * (P + 1) / 2 and A[nbuf - 7].item() == 0. and A[nbuf - 6].item() == 0.

    # ---- distributed Poisson solve with the blocked mode layout [peer][k][jl][mm] (Spec in csrc/common.hpp)
    o = Oracle(case)
    lam, a, b, c, normfft = o.solver_operands(0)
    rhs = o.zeros(); rhs[1:-1, 1:-1, 1:-1] = rng.rand(n1, n2g, n3) - 0.5
    dzf = o.grid()["dzf"][1:-1]                   # compatible r.h.s. (zero volume mean): the singular mode stays O(1)
    rhs[1:-1, 1:-1, 1:-1] -= (rhs[1:-1, 1:-1, 1:-1] * dzf).sum() / (dzf.sum() * n1 * n2g)
    ref = rhs.copy(order="F"); o.solver(ref)
    mine = rhs[1:-1, lo:hi + 1, 1:-1]                                         # (n1, n2l, n3)
    X = np.fft.rfft(mine, axis=0)                                            # modes m = 0..n1/2
    Ac = A[:2 * P * n3 * n2l * cw].numpy().view(np.complex128).reshape(P, n3, n2l, cw)
    Ac[...] = 0
    for m in range(mh):
        Ac[m // cw, :, :, m % cw] = X[m].T                                    # [k][jl]
    comm.alltoall(0, 2 * n3 * n2l * cw)
    Bc = B[:2 * P * n3 * n2l * cw].numpy().view(np.complex128).reshape(P, n3, n2l, cw)
    # the same exchange in k-chunks (cales_alltoall_part_cb: a slice of every peer block, what the pipelined solve sends): same bytes
    whole = B[:2 * P * n3 * n2l * cw].clone(); B[:2 * P * n3 * n2l * cw] = 0.
    nch = 2 if n3 % 2 == 0 else 1
    for ch in range(nch):
        comm.alltoall_part(0, 2 * n3 * n2l * cw, ch * 2 * (n3 // nch) * n2l * cw, 2 * (n3 // nch) * n2l * cw, None)
    assert torch.equal(whole, B[:2 * P * n3 * n2l * cw]), "chunked all-to-all differs from the single exchange"
    T = np.transpose(Bc, (3, 0, 2, 1)).reshape(cw, n2g, n3)                   # (mm, j global, k)
    Xg = np.fft.rfft(rhs[1:-1, 1:-1, 1:-1], axis=0)
    for mm in range(cw):
        if r * cw + mm < mh:
            assert np.abs(T[mm] - Xg[r * cw + mm]).max() < 1e-13, ("forward all-to-all layout", r, mm)
    T = np.fft.fft(T, axis=1)
    dxi, dyi = case.dli[0], case.dli[1]
    lamx = -2. * (1. - np.cos(2 * np.pi * np.arange(mh) / n1)) * dxi ** 2
    lamy = -2. * (1. - np.cos(2 * np.pi * np.arange(n2g) / n2g)) * dyi ** 2
    for mm in range(cw):
        mg = r * cw + mm
        if mg >= mh:
            continue
        lam2 = lamx[mg] + lamy
        T[mm] = thomas(a, b, c, lam2, T[mm].real) + 1j * thomas(a, b, c, lam2, T[mm].imag)
    T = np.fft.ifft(T, axis=1) * n2g                                          # unnormalised inverse, as the device kernels
    Bc[...] = np.transpose(T.reshape(cw, P, n2l, n3), (1, 3, 2, 0))
    comm.alltoall(1, 2 * n3 * n2l * cw)
    Xb = np.zeros((mh, n2l, n3), dtype=np.complex128)
    for m in range(mh):
        Xb[m] = Ac[m // cw, :, :, m % cw].T
    sol = np.fft.irfft(Xb, n=n1, axis=0) * n1 * normfft
    if os.environ.get("GLOO_DEBUG"):
        Tg = np.fft.fft(Xg, axis=1)
        for m in range(mh):
            Tg[m] = thomas(a, b, c, lamx[m] + lamy, Tg[m].real) + 1j * thomas(a, b, c, lamx[m] + lamy, Tg[m].imag)
        Tg = np.fft.ifft(Tg, axis=1) * n2g
        print(r, "Xb err", np.abs(Xb - Tg[:, lo - 1:hi, :]).max(), np.abs(Tg).max())
        solg = np.fft.irfft(Tg, n=n1, axis=0) * n1 * normfft
        print(r, "sol err", np.abs(sol - solg[:, lo - 1:hi, :]).max(), "ref err", np.abs(solg - ref[1:-1, 1:-1, 1:-1] - (solg - ref[1:-1, 1:-1, 1:-1]).mean()).max())
    want = ref[1:-1, lo:hi + 1, 1:-1]
    err = np.abs((sol - sol.mean()) - (want - want.mean())).max() / np.abs(want).max()
    # the zero mode is round-off defined (solver.f90:165): compare with the GLOBAL means removed
    t = torch.tensor([sol.sum(), want.sum()]); dist.all_reduce(t)
    ms, mw = t[0].item() / (n1 * n2g * n3), t[1].item() / (n1 * n2g * n3)
    err = np.abs((sol - ms) - (want - mw)).max() / np.abs(want - mw).max()
    assert err < 1e-11, err
    if r == 0:
        print("GLOO_OK", P, err)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
