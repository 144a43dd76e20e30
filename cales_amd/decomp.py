"""y-slab domain decomposition of the hot path over the GPUs of one node.

The reference decomposes in 2-D pencils (src/initmpi.f90:34-206) and needs four pencil transposes per
Poisson solve (src/solver.f90:50-66) plus halo exchanges in two directions (src/bound.f90:619-723). For
<= 8 GPUs a 1-D decomposition along y is enough: x (contiguous) and z (tridiagonal sweeps, wall planes,
plane averages) stay local, a solve needs ONE all-to-all pair, halos have two neighbours.

libcales_hip.so packs/unpacks on the device and calls back for the three exchanges (include/cales.h,
"multi-GPU"); this module provides the callbacks:

  TorchComm     torch.distributed process group: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" on CPU tensors
  LoopbackComm  P ranks emulated by P threads of ONE process on ONE GPU (device-to-device copies) -- lets the
                1-GPU test box exercise the multi-rank kernels and layouts
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import List, Optional

import numpy as np

from . import capi
from .hotpath import CalesError, HotPath, _p
from .nml import Case

HALO_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
A2A_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64)
ARED_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int)
HALO_S_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)
A2A_PART_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)


def slab_rows(ng2: int, nranks: int, rank: int):
    """Global rows [lo, hi] (1-based, inclusive) owned by `rank` (cales_create: lo(2) = rank*ng2/P + 1)."""
    n2l = ng2 // nranks
    return rank * n2l + 1, (rank + 1) * n2l


def mode_block_width(ng1: int, nranks: int) -> int:
    """Complex x-modes per rank in the transposed layout: ceil((ng1/2+1)/P); the last block is padded. (The library rounds this up to a multiple of eight --
    rows of whole 128-B lines -- where that adds 6 % or less, k_solver.hip solver_setup: `mode_columns_per_rank` of cales_describe_plan is its figure.)"""
    return (ng1 // 2 + 1 + nranks - 1) // nranks


def y_neighbours(rank: int, nranks: int, periodic: bool):
    """(lower, upper) slab neighbours, None at a non-periodic end (MPI_PROC_NULL in src/initmpi.f90:201-204)."""
    lo = (rank - 1) % nranks if (periodic or rank > 0) else None
    hi = (rank + 1) % nranks if (periodic or rank < nranks - 1) else None
    return lo, hi


class TorchComm:
    """Exchanges over a torch.distributed process group; A and B are 1-D tensors of the library's reals (float64, or float32 for the single-precision build) (GPU for nccl, CPU for gloo)."""

    def __init__(self, dist, torch, A, B, periodic_y: bool, group=None):
        self.dist, self.torch, self.A, self.B, self.per = dist, torch, A, B, periodic_y
        self.P, self.r = dist.get_world_size(group), dist.get_rank(group)
        self.group = group
        self.backend = dist.get_backend(group)

    def halo(self, off_slo, off_shi, off_rlo, off_rhi, count) -> int:
        d = self.dist
        lo, hi = y_neighbours(self.r, self.P, self.per)
        ops = []
        # order matters when lower == upper (P = 2, periodic): my "lo" row is the peer's upper ghost
        if lo is not None:
            ops.append(d.P2POp(d.isend, self.A[off_slo:off_slo + count], lo, self.group))
        if hi is not None:
            ops.append(d.P2POp(d.irecv, self.B[off_rhi:off_rhi + count], hi, self.group))
            ops.append(d.P2POp(d.isend, self.A[off_shi:off_shi + count], hi, self.group))
        if lo is not None:
            ops.append(d.P2POp(d.irecv, self.B[off_rlo:off_rlo + count], lo, self.group))
        if ops:
            for req in d.batch_isend_irecv(ops):
                req.wait()
        return 0

    def alltoall(self, direction, count) -> int:
        src, dst = (self.A, self.B) if direction == 0 else (self.B, self.A)
        n = self.P * count
        if self.backend == "gloo":      # gloo has no all_to_all: pairwise exchange
            d = self.dist
            ops = []
            for q in range(self.P):
                if q == self.r:
                    dst[q * count:(q + 1) * count].copy_(src[q * count:(q + 1) * count])
                else:
                    ops.append(d.P2POp(d.isend, src[q * count:(q + 1) * count], q, self.group))
                    ops.append(d.P2POp(d.irecv, dst[q * count:(q + 1) * count], q, self.group))
            for req in d.batch_isend_irecv(ops) if ops else []:
                req.wait()
        else:
            self.dist.all_to_all_single(dst[:n], src[:n], group=self.group)
        return 0

    def allreduce(self, off, count, op) -> int:
        d = self.dist
        rop = (d.ReduceOp.SUM, d.ReduceOp.MAX, d.ReduceOp.MIN)[op]
        d.all_reduce(self.A[off:off + count], op=rop, group=self.group)
        return 0

    def _on(self, stream):
        """context that issues the collectives on the library's second stream (a raw hipStream_t), or a no-op on CPU tensors"""
        import contextlib
        if not stream or self.backend == "gloo":
            return contextlib.nullcontext()
        return self.torch.cuda.stream(self.torch.cuda.ExternalStream(int(stream)))

    def halo_s(self, off_slo, off_shi, off_rlo, off_rhi, count, stream) -> int:
        with self._on(stream):
            return self.halo(off_slo, off_shi, off_rlo, off_rhi, count)

    def alltoall_part(self, direction, peer_stride, off, count, stream) -> int:
        """One k-chunk of the transposition (include/cales.h, cales_alltoall_part_cb): slice [off, off+count) of every peer block."""
        src, dst = (self.A, self.B) if direction == 0 else (self.B, self.A)
        sl = [slice(q * peer_stride + off, q * peer_stride + off + count) for q in range(self.P)]
        d = self.dist
        with self._on(stream):
            if self.backend == "gloo":
                ops = []
                for q in range(self.P):
                    if q == self.r:
                        dst[sl[q]].copy_(src[sl[q]])
                    else:
                        ops.append(d.P2POp(d.isend, src[sl[q]], q, self.group))
                        ops.append(d.P2POp(d.irecv, dst[sl[q]], q, self.group))
                for req in d.batch_isend_irecv(ops) if ops else []:
                    req.wait()
            else:
                d.all_to_all([dst[x] for x in sl], [src[x] for x in sl], group=self.group)
        return 0


class StagedGlooComm(TorchComm):
    """The same exchanges over a gloo process group with the library's buffers on the GPU: every message is staged through host memory
    (device -> pinned host, gloo, host -> device), in order on the library's stream. Not a production path -- RCCL is (comm_rccl.cpp or
    TorchComm on nccl) -- but it runs several real PROCESSES on ONE GPU, which RCCL refuses ("Duplicate GPU detected"): the process-level
    orchestration of `bench.py --gpus N` and of the slab layer (rendezvous, rank order, slab-wise initial fields, reductions over ranks) is
    executed on the one-GPU test box this way."""

    def __init__(self, dist, torch, A, B, periodic_y: bool, group=None):
        super().__init__(dist, torch, A, B, periodic_y, group)
        self.dA, self.dB = A, B
        self.hA = torch.empty(A.numel(), dtype=A.dtype).pin_memory()
        self.hB = torch.empty(B.numel(), dtype=B.dtype).pin_memory()

    def _down(self, dev, host, slices):
        for sl in slices:
            host[sl].copy_(dev[sl], non_blocking=True)
        self.torch.cuda.current_stream().synchronize()

    def _up(self, dev, host, slices):
        for sl in slices:
            dev[sl].copy_(host[sl], non_blocking=True)

    def halo(self, off_slo, off_shi, off_rlo, off_rhi, count) -> int:
        lo, hi = y_neighbours(self.r, self.P, self.per)
        send = ([slice(off_slo, off_slo + count)] if lo is not None else []) + ([slice(off_shi, off_shi + count)] if hi is not None else [])
        recv = ([slice(off_rhi, off_rhi + count)] if hi is not None else []) + ([slice(off_rlo, off_rlo + count)] if lo is not None else [])
        self._down(self.dA, self.hA, send)
        self.A, self.B = self.hA, self.hB
        try:
            super().halo(off_slo, off_shi, off_rlo, off_rhi, count)
        finally:
            self.A, self.B = self.dA, self.dB
        self._up(self.dB, self.hB, recv)
        return 0

    def alltoall(self, direction, count) -> int:
        n = self.P * count
        dsrc, hsrc, ddst, hdst = (self.dA, self.hA, self.dB, self.hB) if direction == 0 else (self.dB, self.hB, self.dA, self.hA)
        self._down(dsrc, hsrc, [slice(0, n)])
        self.A, self.B = self.hA, self.hB
        try:
            super().alltoall(direction, count)
        finally:
            self.A, self.B = self.dA, self.dB
        self._up(ddst, hdst, [slice(0, n)])
        return 0

    def allreduce(self, off, count, op) -> int:
        sl = [slice(off, off + count)]
        self._down(self.dA, self.hA, sl)
        self.A = self.hA
        try:
            super().allreduce(off, count, op)
        finally:
            self.A = self.dA
        self._up(self.dA, self.hA, sl)
        return 0


class LoopbackWorld:
    """Shared state of P emulated ranks (threads) on one device. `events=True`: the exchanges are ordered by recorded HIP events only (no
    host synchronisation of any stream), with a delay in front of every copy -- the form in which a missing consumer-side wait in the
    library (hipStreamWaitEvent on the context's stream for data that arrives on the second stream) shows up as stale data instead of
    being masked by the host waits of the plain form."""

    def __init__(self, nranks: int, events: bool = False, delay_cycles: int = 200000):
        self.P = nranks
        self.barrier = threading.Barrier(nranks)
        self.A: List = [None] * nranks
        self.B: List = [None] * nranks
        self.events = events
        self.delay = delay_cycles
        self.ev_ready: List = [None] * nranks
        self.ev_done: List = [None] * nranks


class LoopbackComm:
    def __init__(self, world: LoopbackWorld, rank: int, torch, A, B, periodic_y: bool, stream):
        self.w, self.r, self.torch, self.A, self.B, self.per, self.stream = world, rank, torch, A, B, periodic_y, stream
        self.P = world.P
        world.A[rank], world.B[rank] = A, B

    def _rendezvous(self):
        self.stream.synchronize()
        self.w.barrier.wait()

    def _exchange(self, st, copies, after=None):
        """Runs `copies()` (reads of the peers' buffers into mine) on stream `st` once every rank's buffers are ready, and returns when no peer
        reads my buffers any more. Plain form: host waits on the stream around two thread barriers. Event form: every rank records an event
        behind what it has queued, the threads meet (host only: the events are recorded, not complete), each stream waits for all events,
        sleeps, copies, records a second event; `after()` (writes into my own buffers that peers were reading) follows the peers' second events."""
        t, w = self.torch, self.w
        if not w.events:
            st.synchronize(); w.barrier.wait()
            with t.cuda.stream(st):
                out = copies()
            st.synchronize(); w.barrier.wait()
            if after is not None:
                with t.cuda.stream(st):
                    after(out)
                st.synchronize(); w.barrier.wait()
            return
        ev = t.cuda.Event(); ev.record(st); w.ev_ready[self.r] = ev
        w.barrier.wait()
        with t.cuda.stream(st):
            for q in range(self.P):
                st.wait_event(w.ev_ready[q])
            if w.delay:
                t.cuda._sleep(w.delay)
            out = copies()
            ev2 = t.cuda.Event(); ev2.record(st); w.ev_done[self.r] = ev2
        w.barrier.wait()
        with t.cuda.stream(st):
            for q in range(self.P):
                st.wait_event(w.ev_done[q])
            if after is not None:
                after(out)
        w.barrier.wait()      # the event slots are free for the next exchange

    def _halo_copies(self, off_slo, off_shi, off_rlo, off_rhi, count):
        lo, hi = y_neighbours(self.r, self.P, self.per)

        def copies():
            if lo is not None:      # lower neighbour's "hi" row -> my lower ghost
                self.B[off_rlo:off_rlo + count].copy_(self.w.A[lo][off_shi:off_shi + count])
            if hi is not None:      # upper neighbour's "lo" row -> my upper ghost
                self.B[off_rhi:off_rhi + count].copy_(self.w.A[hi][off_slo:off_slo + count])
        return copies

    def halo(self, off_slo, off_shi, off_rlo, off_rhi, count) -> int:
        self._exchange(self.stream, self._halo_copies(off_slo, off_shi, off_rlo, off_rhi, count))
        return 0

    def alltoall(self, direction, count) -> int:
        src_all = self.w.A if direction == 0 else self.w.B
        dst = self.B if direction == 0 else self.A

        def copies():
            for q in range(self.P):     # block r of rank q's send buffer -> my block q
                dst[q * count:(q + 1) * count].copy_(src_all[q][self.r * count:(self.r + 1) * count])
        self._exchange(self.stream, copies)
        return 0

    def _ext(self, stream):
        return self.torch.cuda.ExternalStream(int(stream)) if stream else self.stream

    def halo_s(self, off_slo, off_shi, off_rlo, off_rhi, count, stream) -> int:
        self._exchange(self._ext(stream), self._halo_copies(off_slo, off_shi, off_rlo, off_rhi, count))
        return 0

    def alltoall_part(self, direction, peer_stride, off, count, stream) -> int:
        src_all = self.w.A if direction == 0 else self.w.B
        dst = self.B if direction == 0 else self.A

        def copies():
            for q in range(self.P):     # slice of block r of rank q's send buffer -> the same slice of my block q
                dst[q * peer_stride + off:q * peer_stride + off + count].copy_(src_all[q][self.r * peer_stride + off:self.r * peer_stride + off + count])
        self._exchange(self._ext(stream), copies)
        return 0

    def allreduce(self, off, count, op) -> int:
        t = self.torch

        def reduce_():
            parts = t.stack([self.w.A[q][off:off + count] for q in range(self.P)])
            return parts.sum(0) if op == 0 else (parts.max(0).values if op == 1 else parts.min(0).values)

        def store(red):       # in place, once no peer reads my operand any more
            self.A[off:off + count].copy_(red)
        self._exchange(self.stream, reduce_, store)
        return 0


class SlabHotPath(HotPath):
    """HotPath of one rank of a y-slab decomposition. `comm_factory(A, B, periodic_y, stream)` builds the exchanger."""

    def __init__(self, case: Case, dist=None, torch=None, nranks: Optional[int] = None, rank: Optional[int] = None,
                 loopback: Optional[LoopbackWorld] = None, native: Optional[bool] = None):
        if torch is None:
            import torch as _t
            torch = _t
        self.torch = torch
        if loopback is not None:
            assert nranks is not None and rank is not None
        else:
            nranks, rank = dist.get_world_size(), dist.get_rank()
        # a dedicated (non-default) stream: its raw handle is what the library queues on, and the collectives
        # are issued under `with torch.cuda.stream(...)` so RCCL orders itself after the library's kernels
        self.stream = torch.cuda.Stream()
        if case.ng[1] % nranks:
            raise CalesError("ng(2) must be divisible by the number of ranks (y-slab decomposition)")
        super().__init__(case, nranks=nranks, rank=rank, stream=self.stream.cuda_stream)
        self.nranks, self.rank = nranks, rank
        n = C.c_int64(0)
        self._chk(self.L.cales_comm_buffer_doubles(self.h, C.byref(n)))
        self.nbuf = n.value
        # exchanges: by the library itself with RCCL (default for real process groups on GPUs; CALES_COMM=torch disables), or
        # through callbacks into torch.distributed / the loopback world
        if native is None:
            native = loopback is None and os.environ.get("CALES_COMM", "rccl") == "rccl" and dist.get_backend() == "nccl"
        self.native = bool(native) and self._init_native(dist)
        if self.native:
            return
        with torch.cuda.stream(self.stream):
            self.A = torch.zeros(self.nbuf, dtype=torch.float32 if capi.SINGLE else torch.float64, device="cuda")
            self.B = torch.zeros(self.nbuf, dtype=torch.float32 if capi.SINGLE else torch.float64, device="cuda")
        self.stream.synchronize()
        per_y = bool(case.cbcpre[0, 1] == "P" and case.cbcpre[1, 1] == "P")
        if loopback is not None:
            self.comm = LoopbackComm(loopback, rank, torch, self.A, self.B, per_y, self.stream)
        elif dist.get_backend() == "gloo":      # several processes on one GPU (tests): messages staged through the host
            self.comm = StagedGlooComm(dist, torch, self.A, self.B, per_y)
        else:
            self.comm = TorchComm(dist, torch, self.A, self.B, per_y)
        # keep the ctypes thunks alive for the life of the context
        self._cb = (HALO_CB(lambda u, a, b, c_, d, n_: self._guard(self.comm.halo, a, b, c_, d, n_)),
                    A2A_CB(lambda u, d, n_: self._guard(self.comm.alltoall, d, n_)),
                    ARED_CB(lambda u, o, n_, op: self._guard(self.comm.allreduce, o, n_, op)))
        self._chk(self.L.cales_set_comm(self.h, self._cb[0], self._cb[1], self._cb[2], None,
                                        C.c_void_p(self.A.data_ptr()), C.c_void_p(self.B.data_ptr()), C.c_int64(self.nbuf)))
        # exchanges beside the kernels on the library's second stream: registered always, used only with CALES_OVERLAP=1 (common.hpp Flags)
        self._cb2 = (HALO_S_CB(lambda u, a, b, c_, d, n_, st: self._guard2(self.comm.halo_s, a, b, c_, d, n_, st)),
                     A2A_PART_CB(lambda u, d, ps, o, n_, st: self._guard2(self.comm.alltoall_part, d, ps, o, n_, st)))
        if not isinstance(self.comm, StagedGlooComm):      # (the staged exchanges block the host: kept in order on the one stream)
            self._chk(self.L.cales_set_comm_overlap(self.h, self._cb2[0], self._cb2[1]))

    def _init_native(self, dist) -> bool:
        """Rank 0 creates the RCCL rendezvous token, torch.distributed carries it, every rank joins (collective). All ranks
        take the same branch: the decision travels with the broadcast."""
        token = [None]
        if dist.get_rank() == 0:
            buf = (C.c_ubyte * 128)()
            if self.L.cales_comm_unique_id(buf) == 0:
                token[0] = bytes(buf)
        dist.broadcast_object_list(token, src=0)
        if token[0] is None:
            return False
        buf = (C.c_ubyte * 128).from_buffer_copy(token[0])
        self._chk(self.L.cales_comm_init_rccl(self.h, buf))
        return True

    def _guard(self, fn, *a) -> int:
        try:
            with self.torch.cuda.stream(self.stream):
                return int(fn(*a))
        except Exception as e:           # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            self._cb_error = e
            return 1

    def _guard2(self, fn, *a) -> int:
        try:
            return int(fn(*a))
        except Exception as e:           # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            self._cb_error = e
            return 1

    def upload_initial(self):
        """cales_initflow_slab: the rank's rows of the deterministic initial field (src/initflow.f90:17)."""
        u, v, w, p = (self.zeros() for _ in range(4))
        rc = self.L.cales_initflow_slab(C.byref(self.cs), self.case.inivel.encode(), int(self.case.is_wallturb), _p(u), _p(v), _p(w), _p(p))
        if rc:
            raise CalesError(f"cales_initflow_slab failed ({rc})")
        self.upload(u, v, w, p)
        return u, v, w, p

    def upload_global(self, u, v, w, p):
        """Slices this rank's slab (with its y ghost rows) out of global haloed arrays."""
        j0 = self.lo[1] - 1
        sl = slice(j0, j0 + self.n[1] + 2)
        self.upload(*(np.asfortranarray(a[:, sl, :]) for a in (u, v, w, p)))

    def set_global(self, name: str, a: np.ndarray):
        j0 = self.lo[1] - 1
        self.set(name, np.asfortranarray(a[:, j0:j0 + self.n[1] + 2, :]))


def run_loopback(case: Case, nranks: int, body, events: Optional[bool] = None):
    """Runs `body(hotpath, rank)` on `nranks` emulated ranks (threads) sharing one GPU; returns the list of results. `events` (default: the
    environment variable CALES_LOOPBACK_EVENTS): exchanges ordered by HIP events only, see LoopbackWorld."""
    import torch
    if events is None:
        events = os.environ.get("CALES_LOOPBACK_EVENTS", "0") not in ("", "0")
    world = LoopbackWorld(nranks, events=bool(events))
    out: List = [None] * nranks
    err: List = [None] * nranks

    def work(r):
        try:
            torch.cuda.set_device(0)
            h = SlabHotPath(case, torch=torch, nranks=nranks, rank=r, loopback=world)
            out[r] = body(h, r)
            h.sync()
            world.barrier.wait()
            h.close()
        except BaseException as e:      # noqa: BLE001 - report in the main thread
            err[r] = e
            world.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out
