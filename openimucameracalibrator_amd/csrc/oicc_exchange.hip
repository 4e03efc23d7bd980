// liboicc_hip, host side: multi-GPU transport -- the RCCL entry points bound at run time, rank-consistent candidates, the
// owner-computes exchange of the normal equations (include/oicc_hip.h: oicc_set_shard / oicc_set_exchange; see oicc_problem.h).
#include "oicc_problem.h"

namespace oicc {


// ---- native RCCL binding (no link-time dependency: the RCCL of the process is found at run time) ----
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false, p2p = false;
};
RcclApi& rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  void* handles[4] = {RTLD_DEFAULT, dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD), dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD), nullptr};
  for (int k = 0; k < 5 && !api.ok; ++k) {
    void* h = k < 3 ? handles[k] : (k == 3 ? dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL) : dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL));
    if (k > 0 && h == nullptr) continue;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(h, "ncclBroadcast"));
    api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(h, "ncclSend")); api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(h, "ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart")); api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy && api.Broadcast;
    api.p2p = api.ok && api.Send && api.Recv && api.GroupStart && api.GroupEnd;
  }
  return api;
}
void rccl_release(oicc_problem* p) { if (p->rccl_comm) { (void)rccl_api().CommDestroy(static_cast<ncclComm_t>(p->rccl_comm)); p->rccl_comm = nullptr; } }
int rccl_reduce_in_place(void* user, void* device_ptr, int64_t count, void* stream) {
  oicc_problem* p = static_cast<oicc_problem*>(user);
  return rccl_api().AllReduce(device_ptr, device_ptr, size_t(count), ncclDouble, ncclSum, static_cast<ncclComm_t>(p->rccl_comm),
                              static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : -1;
}
// rank 0's copy to every rank, in place (candidate parameters and the step's scalars: all ranks continue from identical bits)
int rccl_broadcast_from_root(oicc_problem* p, void* device_ptr, int64_t count_doubles, hipStream_t stream) {
  return rccl_api().Broadcast(device_ptr, device_ptr, size_t(count_doubles), ncclDouble, 0, static_cast<ncclComm_t>(p->rccl_comm), stream) == ncclSuccess ? 0 : -1;
}
// All ranks continue from identical bits: `xv` (a parameter vector) and, with `with_state`, the step scalars of LmState.
// Native RCCL: rank 0's copy is broadcast.  All-reduce hook (no broadcast there): the mean over the ranks of a pack that the hook
// sums (kernels_solve.hip) -- the ranks' values differ in the last bits only (fp64 atomics of their own solves / sweeps).
int make_rank_consistent(oicc_problem* p, double* xv, bool with_state, hipStream_t st) {
  if (p->rccl_comm != nullptr) {
    if (p->rccl_nranks <= 1) return OICC_OK;
    if (rccl_broadcast_from_root(p, xv, p->pl.total, st) != 0 || (with_state && rccl_broadcast_from_root(p, p->d_state.p, int64_t(sizeof(LmState) / sizeof(double)), st) != 0)) {
      p->err = "broadcast of the candidate failed"; return OICC_ERR_STATE; }
    return OICC_OK;
  }
  if (p->reduce == nullptr) return OICC_OK;
  const int64_t n = p->pl.total;
  if (!p->d_rank_pack.resize(size_t(n + 5))) { p->err = "hipMalloc rank pack"; return OICC_ERR_HIP; }
  launch_rank_pack(xv, n, with_state ? p->d_state.p : nullptr, p->d_rank_pack.p, st);
  if (p->reduce(p->reduce_user, p->d_rank_pack.p, n + 5, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  launch_rank_unpack(xv, n, with_state ? p->d_state.p : nullptr, p->d_rank_pack.p, st);
  return OICC_OK;
}
// ---- owner-computes exchange of the packed normal equations (include/oicc_hip.h: oicc_set_shard) ----
bool owner_exchange_ready(const oicc_problem* p) {
  if (!p->owner.valid || p->shard_n <= 1 || p->reduce == nullptr) return false;
  if (p->rccl_comm != nullptr) return rccl_api().p2p && p->rccl_nranks == p->shard_n;
  return p->exchange != nullptr;
}
int owner_exchange(oicc_problem* p, const NormalEq& ne, hipStream_t st, int64_t* bytes_moved) {
  const oicc_problem::OwnerPlan& op = p->owner;
  const TangentLayout& tl = p->tl;
  const int n = p->shard_n, me = p->shard_rank, L = tl.W + tl.a + 1;
  const bool native = p->rccl_comm != nullptr;
  ncclComm_t comm = static_cast<ncclComm_t>(p->rccl_comm);
  RcclApi& api = rccl_api();
  int64_t moved = 0;
  // (1) halo: partial rows of ranges this rank does not own go to their owners; what the others hold of this rank's range comes in
  //     and is added.  Peers in ascending rank order on every rank, the lower rank of a pair sends first: no cyclic wait with a
  //     blocking transport.
  for (int q = 0; q < n; ++q) {
    if (q == me) continue;
    const int ns = op.send_off[q + 1] - op.send_off[q], nr = op.recv_off[q + 1] - op.recv_off[q];
    if (ns == 0 && nr == 0) continue;
    launch_ne_pack_rows(ne, tl, p->d_xrows.p + op.send_off[q], ns, p->d_xsend.p, st);
    if (native) {
      bool ok = api.GroupStart() == ncclSuccess;
      if (ns) ok = ok && api.Send(p->d_xsend.p, size_t(ns) * L, ncclDouble, q, comm, st) == ncclSuccess;
      if (nr) ok = ok && api.Recv(p->d_xrecv.p, size_t(nr) * L, ncclDouble, q, comm, st) == ncclSuccess;
      ok = (api.GroupEnd() == ncclSuccess) && ok;
      if (!ok) { p->err = "ncclSend / ncclRecv of the halo rows failed"; return OICC_ERR_STATE; }
    } else if (p->exchange(p->exchange_user, OICC_XCHG_SENDRECV, p->d_xsend.p, int64_t(ns) * L, p->d_xrecv.p, int64_t(nr) * L, q, st) != 0) {
      p->err = "exchange callback (send / receive) failed"; return OICC_ERR_STATE; }
    launch_ne_add_rows(ne, tl, p->d_xrows.p + op.recv_off[q], nr, p->d_xrecv.p, st);
    moved += int64_t(ns + nr) * L * int64_t(sizeof(double));
  }
  // (2) gather: every rank's owned range -- its band rows (one contiguous piece), its entries of every arrow row and of the gradient --
  //     broadcast from the owner, in place
  auto bcast = [&](double* ptr, int64_t count, int root) -> bool {
    if (count <= 0) return true;
    if (root != me) moved += count * int64_t(sizeof(double));
    if (native) return api.Broadcast(ptr, ptr, size_t(count), ncclDouble, root, comm, st) == ncclSuccess;
    return p->exchange(p->exchange_user, OICC_XCHG_BROADCAST, ptr, count, ptr, count, root, st) == 0;
  };
  bool ok = true;
  if (native) ok = api.GroupStart() == ncclSuccess;
  for (int k = 0; k < n && ok; ++k) {
    const int64_t r0 = op.cut[k], nr = op.cut[k + 1] - op.cut[k];
    ok = ok && bcast(ne.band() + r0 * tl.W, nr * tl.W, k);
    for (int c = 0; c < tl.a && ok; ++c) ok = bcast(ne.Et() + int64_t(c) * tl.Pb + r0, nr, k);
    ok = ok && bcast(ne.g() + r0, nr, k);
  }
  if (native) ok = (api.GroupEnd() == ncclSuccess) && ok;
  if (!ok) { p->err = "gather of the owned band ranges failed"; return OICC_ERR_STATE; }
  // (3) what every rank contributes to: the arrow corner, the arrow part of the gradient and the cost
  if (tl.a > 0 && p->reduce(p->reduce_user, ne.C(), int64_t(tl.a) * tl.a, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  if (p->reduce(p->reduce_user, ne.g() + tl.Pb, int64_t(tl.a) + 1, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }   // (the cost follows the gradient in the packed buffer)
  moved += 2 * (int64_t(tl.a) * tl.a + tl.a + 1) * int64_t(sizeof(double));
  if (bytes_moved) *bytes_moved = moved;
  return OICC_OK;
}

}  // namespace oicc

extern "C" {

int oicc_rccl_get_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!id) return OICC_ERR_INVALID_ARG;
  RcclApi& api = rccl_api();
  if (!api.ok) return OICC_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (api.GetUniqueId(&u) != ncclSuccess) return OICC_ERR_HIP;
  std::memcpy(id, &u, 128);
  return OICC_OK;
}

int oicc_rccl_init(oicc_problem* p, int32_t nranks, int32_t rank, const uint8_t id[128]) {
  ARG(p, id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "bad RCCL rank / size");
  RcclApi& api = rccl_api();
  if (!api.ok) { p->err = "RCCL (librccl.so.1) not found in this process"; return OICC_ERR_UNSUPPORTED; }
  (void)hipSetDevice(p->device);
  if (p->rccl_comm) { (void)api.CommDestroy(static_cast<ncclComm_t>(p->rccl_comm)); p->rccl_comm = nullptr; }
  ncclUniqueId u; std::memcpy(&u, id, 128);
  ncclComm_t comm = nullptr;
  if (api.CommInitRank(&comm, nranks, u, rank) != ncclSuccess) { p->err = "ncclCommInitRank failed"; return OICC_ERR_HIP; }
  p->rccl_comm = comm; p->rccl_nranks = nranks;
  p->reduce = rccl_reduce_in_place; p->reduce_user = p;
  return OICC_OK;
}

int oicc_set_shard(oicc_problem* p, int32_t nranks, int32_t rank) {
  ARG(p, nranks >= 1 && rank >= 0 && rank < nranks, "shard rank");
  p->shard_n = nranks; p->shard_rank = rank; p->layout_flags = -1;
  return OICC_OK;
}

int oicc_set_exchange(oicc_problem* p, oicc_exchange_fn fn, void* user) { p->exchange = fn; p->exchange_user = user; return OICC_OK; }

int oicc_time_allreduce(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes) {
  int rc = prepare(p, flags); if (rc) return rc;
  if (!p->reduce) { p->err = "no reduction path installed (oicc_rccl_init / oicc_set_allreduce)"; return OICC_ERR_STATE; }
  hipStream_t st = p->stream;
  HIPCK(p, hipMemsetAsync(p->d_ne2.p, 0, p->ne.total * sizeof(double), st));   // (the second buffer: the current system stays intact)
  hipEvent_t e0, e1; HIPCK(p, hipEventCreate(&e0)); HIPCK(p, hipEventCreate(&e1));
  if (p->reduce(p->reduce_user, p->d_ne2.p, p->ne.total, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }   // warm-up (connection set-up)
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) if (p->reduce(p->reduce_user, p->d_ne2.p, p->ne.total, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_call) *ms_per_call = double(ms) / std::max(repeats, 1);
  if (bytes) *bytes = int64_t(p->ne.total) * int64_t(sizeof(double));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return OICC_OK;
}

int oicc_time_exchange(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes_moved) {
  int rc = prepare(p, flags); if (rc) return rc;
  if (!owner_exchange_ready(p)) { p->err = "owner-computes exchange not set up (oicc_set_shard, remote measurements with their owners, a transport)"; return OICC_ERR_STATE; }
  if (repeats < 0) return OICC_OK;                                               // (a local question: is the exchange set up? nothing is sent)
  hipStream_t st = p->stream;
  HIPCK(p, hipMemsetAsync(p->d_ne2.p, 0, p->ne.total * sizeof(double), st));   // (the second buffer: the current system stays intact)
  hipEvent_t e0, e1; HIPCK(p, hipEventCreate(&e0)); HIPCK(p, hipEventCreate(&e1));
  int64_t moved = 0;
  rc = owner_exchange(p, p->ne2, st, &moved); if (rc) return rc;                  // warm-up (connection set-up)
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) { rc = owner_exchange(p, p->ne2, st, &moved); if (rc) return rc; }
  HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_call) *ms_per_call = double(ms) / std::max(repeats, 1);
  if (bytes_moved) *bytes_moved = moved;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return OICC_OK;
}

}  // extern "C"
