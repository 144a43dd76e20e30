#!/usr/bin/env python3
"""Compute side of ONE rank of a y-slab decomposition on one GPU (development aid): rank 0 of P runs alone, its exchanges are
replaced by local copies of the same volume (halo rows from itself, all-to-all blocks copied across, all-reduce left alone),
so the kernels, pack/unpack passes and layouts of the multi-rank path are timed without a second GPU. The numbers say what
a rank computes per step; the fields are NOT a valid flow (the neighbours' data are missing) and the run stops after a few steps.

  python tools/slabbench.py --ranks 8 --ng 512 512 512 --steps 3"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


class SelfComm:
    def __init__(self, torch, A, B, P):
        self.t, self.A, self.B, self.P = torch, A, B, P
        self.calls = {"halo": [0, 0], "alltoall": [0, 0], "allreduce": [0, 0]}      # calls, doubles sent

    def reset(self):
        for v in self.calls.values():
            v[0] = v[1] = 0

    def halo(self, off_slo, off_shi, off_rlo, off_rhi, count):
        self.calls["halo"][0] += 1; self.calls["halo"][1] += 2 * count
        self.B[off_rlo:off_rlo + count].copy_(self.A[off_shi:off_shi + count])
        self.B[off_rhi:off_rhi + count].copy_(self.A[off_slo:off_slo + count])
        return 0

    def alltoall(self, direction, count):
        src, dst = (self.A, self.B) if direction == 0 else (self.B, self.A)
        n = self.P * count
        self.calls["alltoall"][0] += 1; self.calls["alltoall"][1] += (self.P - 1) * count
        dst[:n].copy_(src[:n])
        return 0

    def allreduce(self, off, count, op):
        self.calls["allreduce"][0] += 1; self.calls["allreduce"][1] += count
        return 0

    # the same on the library's second stream (cales_set_comm_overlap): k-chunks of the transposition, deferred scratch-field halos
    def halo_s(self, off_slo, off_shi, off_rlo, off_rhi, count, stream):
        with self.t.cuda.stream(self.t.cuda.ExternalStream(int(stream))):
            return self.halo(off_slo, off_shi, off_rlo, off_rhi, count)

    def alltoall_part(self, direction, peer_stride, off, count, stream):
        src, dst = (self.A, self.B) if direction == 0 else (self.B, self.A)
        self.calls["alltoall"][0] += 1; self.calls["alltoall"][1] += (self.P - 1) * count
        with self.t.cuda.stream(self.t.cuda.ExternalStream(int(stream))):
            for q in range(self.P):
                dst[q * peer_stride + off:q * peer_stride + off + count].copy_(src[q * peer_stride + off:q * peer_stride + off + count])
        return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--ng", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--sgs", default="dsmag")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bench
    from cales_amd import decomp
    case = bench.channel_case(a.ng, a.sgs)
    world = decomp.LoopbackWorld(a.ranks)
    h = decomp.SlabHotPath(case, torch=torch, nranks=a.ranks, rank=0, loopback=world)
    h.comm = SelfComm(torch, h.A, h.B, a.ranks)
    h.upload_initial(); h.startup()
    dt = 1e-4
    h.step(dt); h.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h.step(dt)
    h.sync(); t = (time.perf_counter() - t0) / a.steps
    h.comm.reset()
    h.profile_reset(); h.profile(True)
    for _ in range(a.steps):
        h.step(dt)
    h.sync(); h.profile(False)
    st = h.profile_stats()
    tot = sum(v[1] for k, v in st.items() if not k.startswith("cmpt_sgs")) / a.steps
    print(f"rank 0 of {a.ranks}, slab {h.n}: {1e3 * t:.3f} ms/step wall (local copies instead of exchanges), kernels {tot:.3f} ms/step")
    print("  " + "  ".join(f"{k}={v[1] / a.steps:.3f}" for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:30]))
    print("  exchanges per step: " + ", ".join(f"{k}: {v[0] / a.steps:.0f} calls, {8e-6 * v[1] / a.steps:.1f} MB out" for k, v in h.comm.calls.items()))
    h.close()


if __name__ == "__main__":
    main()
