"""TEST INFRASTRUCTURE: numpy restatement of the second and third block of the reference's out1d_single_point_chan
(src/output.f90:700-1055, idir = 3): the 38 plane sums of the mean-kinetic-energy / Reynolds-stress budgets and the 6 "leakage"
(divergence) measures per z plane. Whole-array slices instead of the reference's loop nest; haloed Fortran-ordered inputs
(0:n1+1, 0:n2+1, 0:n3+1). Pinned: tests/test_oracle_golden.py::test_plane_statistics compares it with the output of the reference's own
routine (its lines compiled inside a wrapper module by oracle/ref/Makefile; the MODULE as a whole needs 2decomp-fft) on the 14 golden
end-of-step states: <= 3e-14 of each column's largest entry."""
import numpy as np


def _sh(a, di=0, dj=0, dk=0):
    """interior view of a haloed array shifted by (di, dj, dk)"""
    n1, n2, n3 = (s - 2 for s in a.shape)
    return a[1 + di:1 + di + n1, 1 + dj:1 + dj + n2, 1 + dk:1 + dk + n3]


def budget_terms(u, v, w, p, dx, dy, dzc, dzf, lx, ly):
    """(38, n3): output.f90:741-1001 (the sums) times dx dy/(lx ly)"""
    n3 = u.shape[2] - 2
    zc = dzc[1:n3 + 1][None, None, :]; zcm = dzc[0:n3][None, None, :]
    zf = dzf[1:n3 + 1][None, None, :]; zfp = dzf[2:n3 + 2][None, None, :]
    U = lambda di=0, dj=0, dk=0: _sh(u, di, dj, dk)
    V = lambda di=0, dj=0, dk=0: _sh(v, di, dj, dk)
    W = lambda di=0, dj=0, dk=0: _sh(w, di, dj, dk)
    P = lambda di=0, dj=0, dk=0: _sh(p, di, dj, dk)
    dudz4 = 0.25 * ((U(dk=1) - U()) / zc + (U() - U(dk=-1)) / zcm + (U(-1, 0, 1) - U(-1)) / zc + (U(-1) - U(-1, 0, -1)) / zcm)
    dwdx4 = 0.25 * ((W(1) - W()) / dx + (W() - W(-1)) / dx + (W(1, 0, -1) - W(dk=-1)) / dx + (W(dk=-1) - W(-1, 0, -1)) / dx)
    dudy4 = 0.25 * ((U(dj=1) - U()) / dy + (U() - U(dj=-1)) / dy + (U(-1, 1) - U(-1)) / dy + (U(-1) - U(-1, -1)) / dy)
    dwdy4 = 0.25 * ((W(dj=1) - W()) / dy + (W() - W(dj=-1)) / dy + (W(0, 1, -1) - W(dk=-1)) / dy + (W(dk=-1) - W(0, -1, -1)) / dy)
    t = [None] * 38
    t[0] = U()
    t[1] = 0.5 * (U() + U(dk=1))
    t[2] = (U(dk=1) - U()) / zc
    t[3] = (U(dk=1) ** 2 - U() ** 2) / zc
    t[4] = 0.25 * (U(dk=1) + U()) * (W() + W(1))
    t[5] = 0.25 * (U(-1) + U()) * (W() + W(dk=-1))
    t[6] = dudz4
    t[7] = 0.125 * (U(dk=1) + U()) ** 2 * (W() + W(1))
    t[8] = P()
    t[9] = (U() - U(-1)) / dx * P()
    t[10] = (((U() - U(-1)) / dx) ** 2
             + 0.25 * (((U(dj=1) - U()) / dy) ** 2 + ((U() - U(dj=-1)) / dy) ** 2 + ((U(-1, 1) - U(-1)) / dy) ** 2 + ((U(-1) - U(-1, -1)) / dy) ** 2)
             + 0.25 * (((U(dk=1) - U()) / zc) ** 2 + ((U() - U(dk=-1)) / zcm) ** 2 + ((U(-1, 0, 1) - U(-1)) / zc) ** 2 + ((U(-1) - U(-1, 0, -1)) / zcm) ** 2))
    t[11] = (V(dk=1) ** 2 - V() ** 2) / zc
    t[12] = 0.125 * (V(dk=1) + V()) ** 2 * (W() + W(dj=1))
    t[13] = (V() - V(dj=-1)) / dy * P()
    t[14] = (0.25 * (((V(1) - V()) / dx) ** 2 + ((V() - V(-1)) / dx) ** 2 + ((V(1, -1) - V(dj=-1)) / dx) ** 2 + ((V(dj=-1) - V(-1, -1)) / dx) ** 2)
             + ((V() - V(dj=-1)) / dy) ** 2
             + 0.25 * (((V(dk=1) - V()) / zc) ** 2 + ((V() - V(dk=-1)) / zcm) ** 2 + ((V(0, -1, 1) - V(dj=-1)) / zc) ** 2 + ((V(dj=-1) - V(0, -1, -1)) / zcm) ** 2))
    t[15] = 0.5 * ((W(dk=1) ** 2 - W() ** 2) / zfp + (W() ** 2 - W(dk=-1) ** 2) / zf)
    t[16] = W() ** 3
    t[17] = W() * 0.5 * (P(dk=1) + P())
    t[18] = (W() - W(dk=-1)) / zf * P()
    t[19] = (0.25 * (((W(1) - W()) / dx) ** 2 + ((W() - W(-1)) / dx) ** 2 + ((W(1, 0, -1) - W(dk=-1)) / dx) ** 2 + ((W(dk=-1) - W(-1, 0, -1)) / dx) ** 2)
             + 0.25 * (((W(dj=1) - W()) / dy) ** 2 + ((W() - W(dj=-1)) / dy) ** 2 + ((W(0, 1, -1) - W(dk=-1)) / dy) ** 2 + ((W(dk=-1) - W(0, -1, -1)) / dy) ** 2)
             + ((W() - W(dk=-1)) / zf) ** 2)
    t[20] = 0.5 * (W() ** 2 + W(dk=-1) ** 2)
    t[21] = (0.25 * (W() + W(dk=1) + W(1, 0, 1) + W(1)) * U(dk=1) - 0.25 * (W() + W(dk=-1) + W(1, 0, -1) + W(1)) * U()) / zc
    t[22] = W() ** 2
    t[23] = 0.125 * (U(dk=1) + U()) * (W() + W(1)) ** 2
    t[24] = 0.5 * (P(dk=1) + P())
    t[25] = 0.25 * (U() + U(dk=1) + U(-1, 0, 1) + U(-1)) * 0.5 * (P(dk=1) + P())
    t[26] = dudz4 * P() + dwdx4 * P()
    t[27] = (U() - U(-1)) / dx * dwdx4 + dudy4 * dwdy4 + dudz4 * ((W() - W(dk=-1)) / zf)
    t[28] = (U(dk=1) - U()) / zc
    t[29] = ((U() - U(-1)) / dx) ** 2; t[30] = ((U(dj=1) - U()) / dy) ** 2; t[31] = ((U(dk=1) - U()) / zc) ** 2
    t[32] = ((V(1) - V()) / dx) ** 2; t[33] = ((V() - V(dj=-1)) / dy) ** 2; t[34] = ((V(dk=1) - V()) / zc) ** 2
    t[35] = ((W(1) - W()) / dx) ** 2; t[36] = ((W(dj=1) - W()) / dy) ** 2; t[37] = ((W() - W(dk=-1)) / zf) ** 2
    ratio = dx * dy / (lx * ly)
    return np.asfortranarray(np.stack([(q * np.ones_like(U())).sum(axis=(0, 1)) * ratio for q in t]))


def leakage_terms(u, v, w, dx, dy, dzf, lx, ly):
    """(6, n3): output.f90:1005-1041"""
    n3 = u.shape[2] - 2
    zf = dzf[1:n3 + 1][None, None, :]
    div = (_sh(w) - _sh(w, dk=-1)) / zf + (_sh(v) - _sh(v, dj=-1)) / dy + (_sh(u) - _sh(u, -1)) / dx
    ratio = dx * dy / (lx * ly)
    a = np.abs(div)
    return np.asfortranarray(np.stack([a.max(axis=(0, 1)), a.sum(axis=(0, 1)) * ratio, div.sum(axis=(0, 1)) * ratio,
                                       (a * zf).max(axis=(0, 1)), (a * zf).sum(axis=(0, 1)) * ratio, (div * zf).sum(axis=(0, 1)) * ratio]))
