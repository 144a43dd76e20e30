// Native exchanges of the y-slab decomposition: RCCL over xGMI, called straight from the library on the context's stream.
//
// The three exchanges of include/cales.h ("multi-GPU") can be served by host callbacks (cales_set_comm; what the Python layer
// does with torch.distributed) or by this file: halo rows with grouped ncclSend/ncclRecv to the two y neighbours, the
// slab <-> mode-block transposition of the Poisson solve with ncclAllToAll, the small reductions with ncclAllReduce, all
// enqueued on the same HIP stream as the kernels around them, so a substep runs without host synchronisation or interpreter
// calls. The reference needs four pencil transposes per solve (src/solver.f90:50-66) and MPI halo exchanges
// (src/bound.f90:619-723); here a solve has one all-to-all pair.
//
// RCCL is opened with dlopen: the library keeps loading (and the single-GPU path keeps working) where it is absent.
#include "common.hpp"
// ncclDataType_t of `real` (nccl.h: ncclFloat = 7, ncclDouble = 8)
#ifdef CALES_SINGLE
#define NCCL_REAL ncclFloat
#else
#define NCCL_REAL ncclDouble
#endif
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllToAll)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
RcclApi g_api;

// A process that already carries an RCCL (PyTorch wheels bundle their own, built for the HIP runtime they also bundle) must
// keep using THAT copy: a second copy from another ROCm release would run on a runtime it was not built for.
int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *out) {
  const char *nm = info->dlpi_name;
  if (nm && std::strstr(nm, "librccl.so")) { *static_cast<std::string *>(out) = nm; return 1; }
  return 0;
}
bool load_api() {
  if (g_api.lib) return true;
  std::string loaded;
  dl_iterate_phdr(find_loaded_rccl, &loaded);
  if (!loaded.empty()) g_api.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_LOCAL);
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *nm : names) { if (g_api.lib) break; g_api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); }
  if (!g_api.lib) { g_api.err = "RCCL not found (dlopen librccl.so.1)"; return false; }
  bool ok = true;
  auto sym = [&](const char *nm) { void *p = dlsym(g_api.lib, nm); if (!p) { ok = false; g_api.err = std::string("RCCL symbol missing: ") + nm; } return p; };
  g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(sym("ncclCommInitRank"));
  g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(sym("ncclCommDestroy"));
  g_api.GroupStart = reinterpret_cast<decltype(g_api.GroupStart)>(sym("ncclGroupStart"));
  g_api.GroupEnd = reinterpret_cast<decltype(g_api.GroupEnd)>(sym("ncclGroupEnd"));
  g_api.Send = reinterpret_cast<decltype(g_api.Send)>(sym("ncclSend"));
  g_api.Recv = reinterpret_cast<decltype(g_api.Recv)>(sym("ncclRecv"));
  g_api.AllToAll = reinterpret_cast<decltype(g_api.AllToAll)>(sym("ncclAllToAll"));
  g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(sym("ncclAllReduce"));
  g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) { dlclose(g_api.lib); g_api.lib = nullptr; }
  return ok;
}

struct NativeComm { ncclComm_t comm = nullptr; real *A = nullptr, *B = nullptr; };

#define NCHK(c, call)                                                                                     \
  do {                                                                                                    \
    const ncclResult_t r_ = (call);                                                                       \
    if (r_ != ncclSuccess) { (c)->err = std::string("RCCL: ") + g_api.GetErrorString(r_) + " in " #call; return 1; } \
  } while (0)

// y neighbours (MPI_PROC_NULL at a non-periodic end, src/initmpi.f90:201-204)
inline void neighbours(const cales_ctx *c, int &lo, int &hi) {
  lo = (c->per_y || c->rank > 0) ? (c->rank - 1 + c->P) % c->P : -1;
  hi = (c->per_y || c->rank < c->P - 1) ? (c->rank + 1) % c->P : -1;
}

int native_halo_on(void *user, int64_t off_slo, int64_t off_shi, int64_t off_rlo, int64_t off_rhi, int64_t count, hipStream_t st) {
  cales_ctx *c = static_cast<cales_ctx *>(user);
  NativeComm *nc = static_cast<NativeComm *>(c->native_comm);
  int lo, hi; neighbours(c, lo, hi);
  // With two ranks and periodic y both neighbours are the same peer: the k-th send to a peer pairs with its k-th receive
  // from us, so "my lowest row" must be sent first and "its lowest row" (my upper ghost) received first.
  NCHK(c, g_api.GroupStart());
  // an error inside the group must still close it: an open group would queue every later collective of this thread for ever
  ncclResult_t rc = ncclSuccess;
  auto add = [&](ncclResult_t r) { if (rc == ncclSuccess) rc = r; };
  if (lo >= 0) add(g_api.Send(nc->A + off_slo, (size_t)count, NCCL_REAL, lo, nc->comm, st));
  if (hi >= 0) {
    add(g_api.Recv(nc->B + off_rhi, (size_t)count, NCCL_REAL, hi, nc->comm, st));
    add(g_api.Send(nc->A + off_shi, (size_t)count, NCCL_REAL, hi, nc->comm, st));
  }
  if (lo >= 0) add(g_api.Recv(nc->B + off_rlo, (size_t)count, NCCL_REAL, lo, nc->comm, st));
  add(g_api.GroupEnd());
  if (rc != ncclSuccess) { c->err = std::string("RCCL: ") + g_api.GetErrorString(rc) + " in the halo exchange"; return 1; }
  return 0;
}
int native_halo(void *user, int64_t off_slo, int64_t off_shi, int64_t off_rlo, int64_t off_rhi, int64_t count) {
  return native_halo_on(user, off_slo, off_shi, off_rlo, off_rhi, count, static_cast<cales_ctx *>(user)->stream);
}
int native_halo_s(void *user, int64_t off_slo, int64_t off_shi, int64_t off_rlo, int64_t off_rhi, int64_t count, void *stream) {
  return native_halo_on(user, off_slo, off_shi, off_rlo, off_rhi, count, static_cast<hipStream_t>(stream));
}
// one k-chunk of the slab <-> mode-block transposition: P sends and P receives of `count` doubles in one group (the own block too:
// RCCL copies it on the device), on the stream the library hands over
int native_alltoall_part(void *user, int dir, int64_t peer_stride, int64_t off, int64_t count, void *stream) {
  cales_ctx *c = static_cast<cales_ctx *>(user);
  NativeComm *nc = static_cast<NativeComm *>(c->native_comm);
  const real *src = dir == 0 ? nc->A : nc->B; real *dst = dir == 0 ? nc->B : nc->A;
  hipStream_t st = static_cast<hipStream_t>(stream);
  NCHK(c, g_api.GroupStart());
  ncclResult_t rc = ncclSuccess;
  for (int p = 0; p < c->P; ++p) {
    const ncclResult_t r1 = g_api.Send(src + (size_t)p * peer_stride + off, (size_t)count, NCCL_REAL, p, nc->comm, st);
    const ncclResult_t r2 = g_api.Recv(dst + (size_t)p * peer_stride + off, (size_t)count, NCCL_REAL, p, nc->comm, st);
    if (rc == ncclSuccess) rc = r1 != ncclSuccess ? r1 : r2;
  }
  const ncclResult_t r3 = g_api.GroupEnd();
  if (rc == ncclSuccess) rc = r3;
  if (rc != ncclSuccess) { c->err = std::string("RCCL: ") + g_api.GetErrorString(rc) + " in a slice of the all-to-all"; return 1; }
  return 0;
}
int native_alltoall(void *user, int dir, int64_t count) {
  cales_ctx *c = static_cast<cales_ctx *>(user);
  NativeComm *nc = static_cast<NativeComm *>(c->native_comm);
  const real *src = dir == 0 ? nc->A : nc->B; real *dst = dir == 0 ? nc->B : nc->A;
  NCHK(c, g_api.AllToAll(src, dst, (size_t)count, NCCL_REAL, nc->comm, c->stream));
  return 0;
}
int native_allreduce(void *user, int64_t off, int64_t count, int op) {
  cales_ctx *c = static_cast<cales_ctx *>(user);
  NativeComm *nc = static_cast<NativeComm *>(c->native_comm);
  const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
  NCHK(c, g_api.AllReduce(nc->A + off, nc->A + off, (size_t)count, NCCL_REAL, rop, nc->comm, c->stream));
  return 0;
}
}  // namespace

extern "C" {

// Rank 0 creates the rendezvous token (CALES_COMM_ID_BYTES = 128 bytes) and hands it to the other ranks by any means
// (the Python layer broadcasts it with torch.distributed; a Fortran host would use MPI_Bcast).
int cales_comm_unique_id(void *id_out) {
  if (!id_out) return 1;
  if (!load_api()) return 2;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  if (g_api.GetUniqueId(&id) != ncclSuccess) return 3;
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

// Collective over all ranks of the decomposition: joins the RCCL communicator, allocates the two staging buffers and
// registers the native exchanges (instead of cales_set_comm).
int cales_comm_init_rccl(cales_ctx *c, const void *id_in) {
  if (!c || !id_in) return 1;
  if (c->P < 1) { c->err = "cales_comm_init_rccl: bad rank count"; return 1; }
  if (!load_api()) { c->err = g_api.err; return 1; }
  NativeComm *nc = new NativeComm;
  ncclUniqueId id; std::memcpy(&id, id_in, sizeof(id));
  const ncclResult_t r = g_api.CommInitRank(&nc->comm, c->P, id, c->rank);
  if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + g_api.GetErrorString(r); delete nc; return 1; }
  int64_t n = 0; cales_comm_buffer_doubles(c, &n);
  if (hipMalloc(&nc->A, n * sizeof(real)) != hipSuccess || hipMalloc(&nc->B, n * sizeof(real)) != hipSuccess) {
    c->err = "cales_comm_init_rccl: hipMalloc of the staging buffers failed"; g_api.CommDestroy(nc->comm); hipFree(nc->A); delete nc; return 1;
  }
  c->native_comm = nc;
  if (int e = cales_set_comm(c, native_halo, native_alltoall, native_allreduce, c, nc->A, nc->B, n)) return e;
  return cales_set_comm_overlap(c, native_halo_s, native_alltoall_part);
}

// called by cales_destroy
void cales_comm_release_native(cales_ctx *c) {
  NativeComm *nc = static_cast<NativeComm *>(c->native_comm);
  if (!nc) return;
  if (nc->comm && g_api.CommDestroy) g_api.CommDestroy(nc->comm);
  hipFree(nc->A); hipFree(nc->B);
  delete nc; c->native_comm = nullptr;
}

// Single-process check of the native exchanges on a one-rank communicator (tests): halo rows to "both neighbours" (= self,
// periodic), all-to-all of one block, all-reduce. Returns 0 when every received value is the expected one.
int cales_comm_selftest(void) {
  if (!load_api()) return 100;
  ncclUniqueId id; ncclComm_t comm;
  if (g_api.GetUniqueId(&id) != ncclSuccess) return 101;
  if (g_api.CommInitRank(&comm, 1, id, 0) != ncclSuccess) return 102;
  const size_t cnt = 1000; real *A = nullptr, *B = nullptr;
  if (hipMalloc(&A, 4 * cnt * sizeof(real)) != hipSuccess || hipMalloc(&B, 4 * cnt * sizeof(real)) != hipSuccess) return 103;
  std::vector<real> h(4 * cnt); for (size_t q = 0; q < 4 * cnt; ++q) h[q] = (real)q;
  hipMemcpy(A, h.data(), 4 * cnt * sizeof(real), hipMemcpyHostToDevice); hipMemset(B, 0, 4 * cnt * sizeof(real));
  hipStream_t s; hipStreamCreate(&s);
  int rc = 0;
  // the order of native_halo with lo = hi = 0
  if (g_api.GroupStart() != ncclSuccess) rc = 104;
  if (!rc && g_api.Send(A, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 105;                 // my "lo" row
  if (!rc && g_api.Recv(B + cnt, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 106;           // -> upper ghost
  if (!rc && g_api.Send(A + cnt, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 107;           // my "hi" row
  if (!rc && g_api.Recv(B, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 108;                 // -> lower ghost
  if (!rc && g_api.GroupEnd() != ncclSuccess) rc = 109;
  if (!rc && g_api.AllToAll(A + 2 * cnt, B + 2 * cnt, cnt, NCCL_REAL, comm, s) != ncclSuccess) rc = 110;
  if (!rc && g_api.AllReduce(A + 3 * cnt, A + 3 * cnt, cnt, NCCL_REAL, ncclSum, comm, s) != ncclSuccess) rc = 111;
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = 112;
  // the exchanges of the second stream (native_alltoall_part, native_halo_s): a k-chunk = a slice of every peer block as one send/recv group on
  // ANOTHER stream, ordered against the first by events only -- producer (a memset on s) -> event -> group on s2 -> event -> consumer copy on s
  if (!rc) {
    hipStream_t s2; hipEvent_t e1, e2; real *C_ = nullptr;
    if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess || hipMalloc(&C_, cnt * sizeof(real)) != hipSuccess) rc = 130;
    const size_t off = 100, part = 300;
    if (!rc && hipMemsetAsync(B, 0, 4 * cnt * sizeof(real), s) != hipSuccess) rc = 131;
    if (!rc && (hipEventRecord(e1, s) != hipSuccess || hipStreamWaitEvent(s2, e1, 0) != hipSuccess)) rc = 132;
    if (!rc && g_api.GroupStart() != ncclSuccess) rc = 133;
    if (!rc && g_api.Send(A + off, part, NCCL_REAL, 0, comm, s2) != ncclSuccess) rc = 134;
    if (!rc && g_api.Recv(B + off, part, NCCL_REAL, 0, comm, s2) != ncclSuccess) rc = 135;
    if (!rc && g_api.GroupEnd() != ncclSuccess) rc = 136;
    if (!rc && (hipEventRecord(e2, s2) != hipSuccess || hipStreamWaitEvent(s, e2, 0) != hipSuccess)) rc = 137;
    if (!rc && hipMemcpyAsync(C_, B, cnt * sizeof(real), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 138;
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = 139;
    if (!rc) {
      std::vector<real> cc(cnt);
      hipMemcpy(cc.data(), C_, cnt * sizeof(real), hipMemcpyDeviceToHost);
      for (size_t q = 0; q < cnt && !rc; ++q) if (cc[q] != ((q >= off && q < off + part) ? h[q] : (real)0)) rc = 140;
    }
    hipStreamSynchronize(s2); hipStreamDestroy(s2); hipEventDestroy(e1); hipEventDestroy(e2); hipFree(C_);
    // restore what the checks below expect of B
    if (!rc) {
      hipMemset(B, 0, 4 * cnt * sizeof(real));
      if (g_api.GroupStart() != ncclSuccess) rc = 141;
      if (!rc && g_api.Send(A, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 142;
      if (!rc && g_api.Recv(B + cnt, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 143;
      if (!rc && g_api.Send(A + cnt, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 144;
      if (!rc && g_api.Recv(B, cnt, NCCL_REAL, 0, comm, s) != ncclSuccess) rc = 145;
      if (!rc && g_api.GroupEnd() != ncclSuccess) rc = 146;
      if (!rc && g_api.AllToAll(A + 2 * cnt, B + 2 * cnt, cnt, NCCL_REAL, comm, s) != ncclSuccess) rc = 147;
      if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = 148;
    }
  }
  if (!rc) {
    std::vector<real> b(4 * cnt), a(4 * cnt);
    hipMemcpy(b.data(), B, 4 * cnt * sizeof(real), hipMemcpyDeviceToHost); hipMemcpy(a.data(), A, 4 * cnt * sizeof(real), hipMemcpyDeviceToHost);
    for (size_t q = 0; q < cnt && !rc; ++q) {
      if (b[cnt + q] != h[q]) rc = 120;              // upper ghost = the (only) neighbour's lowest row
      else if (b[q] != h[cnt + q]) rc = 121;         // lower ghost = its highest row
      else if (b[2 * cnt + q] != h[2 * cnt + q]) rc = 122;
      else if (a[3 * cnt + q] != h[3 * cnt + q]) rc = 123;
    }
  }
  hipStreamDestroy(s); hipFree(A); hipFree(B); g_api.CommDestroy(comm);
  return rc;
}

}  // extern "C"
