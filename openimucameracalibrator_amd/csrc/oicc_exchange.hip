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
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;   // optional (the gather falls back to one broadcast per owner)
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
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
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
// x_identical (round 6): the candidate was retracted from a step every rank holds bit for bit (the gathered step of the distributed
// solve: elementwise retraction of identical inputs) -- only the step's scalars, summed with atomics on every rank, still travel.
int make_rank_consistent(oicc_problem* p, double* xv, bool with_state, hipStream_t st, bool x_identical) {
  if (x_identical && !with_state) return OICC_OK;
  if (p->rccl_comm != nullptr) {
    if (p->rccl_nranks <= 1) return OICC_OK;
    if ((!x_identical && rccl_broadcast_from_root(p, xv, p->pl.total, st) != 0) || (with_state && rccl_broadcast_from_root(p, p->d_state.p, int64_t(sizeof(LmState) / sizeof(double)), st) != 0)) {
      p->err = "broadcast of the candidate failed"; return OICC_ERR_STATE; }
    return OICC_OK;
  }
  if (p->reduce == nullptr) return OICC_OK;
  const int64_t n = x_identical ? 0 : p->pl.total;
  if (!p->d_rank_pack.resize(size_t(n + 5))) { p->err = "hipMalloc rank pack"; return OICC_ERR_HIP; }
  launch_rank_pack(xv, n, with_state ? p->d_state.p : nullptr, p->d_rank_pack.p, st);
  if (p->reduce(p->reduce_user, p->d_rank_pack.p, n + 5, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  launch_rank_unpack(xv, n, with_state ? p->d_state.p : nullptr, p->d_rank_pack.p, st);
  return OICC_OK;
}
// ---- owner-computes exchange of the packed normal equations (include/oicc_hip.h: oicc_set_shard) ----
bool owner_exchange_ready(const oicc_problem* p) {   // this rank alone: can it run the exchange?
  if (!p->owner.valid || p->shard_n <= 1 || p->reduce == nullptr) return false;
  if (p->rccl_comm != nullptr) return rccl_api().p2p && p->rccl_nranks == p->shard_n;
  return p->exchange != nullptr;
}
// ... and all ranks together (round 5, the advisor's finding: a rank that chose the whole-buffer all-reduce while its peers entered
// send / receive would hang them).  Once per layout every rank of a sharded problem (oicc_set_shard with nranks > 1 is a
// collective statement) sums [ready, 1, h, h^2] through the installed reduction, h = a hash of the cuts and of every pair's row
// count as THIS rank derived them: the exchange is used only if every rank is ready and all hashes are equal (n sum h^2 == (sum h)^2,
// exact in doubles for 20-bit hashes); otherwise every rank falls back to the all-reduce of the whole packed buffer.
int owner_exchange_agree(oicc_problem* p, hipStream_t st, bool* use) {
  *use = false;
  if (p->shard_n <= 1 || p->reduce == nullptr) return OICC_OK;
  oicc_problem::OwnerPlan& op = p->owner;
  if (op.agreed_gen == p->layout_gen) { *use = op.agreed; return OICC_OK; }
  const double h = double(op.valid ? ((op.hash ^ (p->opt["distributed_solve"] != 0.0 ? 0x5bd1eu : 0u)) & 0xfffffu) : 0u);   // (the choice of solve is part of what the ranks must agree on)
  double v[4] = {owner_exchange_ready(p) ? 1.0 : 0.0, 1.0, h, h * h};
  if (!p->d_xagree.resize(4)) { p->err = "hipMalloc exchange agreement"; return OICC_ERR_HIP; }
  HIPCK(p, hipMemcpyAsync(p->d_xagree.p, v, sizeof(v), hipMemcpyHostToDevice, st));
  if (p->reduce(p->reduce_user, p->d_xagree.p, 4, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  HIPCK(p, hipMemcpyAsync(v, p->d_xagree.p, sizeof(v), hipMemcpyDeviceToHost, st));
  HIPCK(p, hipStreamSynchronize(st));
  op.agreed = v[0] == v[1] && v[1] == double(p->shard_n) && v[1] * v[3] == v[2] * v[2];
  op.agreed_gen = p->layout_gen;
  if (!op.agreed && p->opt["verbose"] != 0.0) std::printf("[oicc] rank %d: owner-computes exchange not agreed on by all ranks (ready %g of %g, shard size %d): all-reduce of the packed buffer\n", p->shard_rank, v[0], v[1], p->shard_n);
  *use = op.agreed;
  return OICC_OK;
}
// Broadcasts of pieces of the parameter vector between the sets of an owner-computes sweep (oicc_inner.hip): the transport of the exchange
int shard_broadcast_begin(oicc_problem* p) {
  if (p->rccl_comm != nullptr && rccl_api().GroupStart() != ncclSuccess) { p->err = "ncclGroupStart failed"; return OICC_ERR_STATE; }
  return OICC_OK;
}
int shard_broadcast(oicc_problem* p, double* ptr, int64_t count, int root, hipStream_t st) {
  if (count <= 0) return OICC_OK;
  const bool ok = p->rccl_comm != nullptr ? rccl_api().Broadcast(ptr, ptr, size_t(count), ncclDouble, root, static_cast<ncclComm_t>(p->rccl_comm), st) == ncclSuccess
                                          : (p->exchange != nullptr && p->exchange(p->exchange_user, OICC_XCHG_BROADCAST, ptr, count, ptr, count, root, st) == 0);
  if (!ok) { p->err = "broadcast of a piece of the parameter vector failed"; return OICC_ERR_STATE; }
  return OICC_OK;
}
int shard_broadcast_end(oicc_problem* p) {
  if (p->rccl_comm != nullptr && rccl_api().GroupEnd() != ncclSuccess) { p->err = "ncclGroupEnd failed"; return OICC_ERR_STATE; }
  return OICC_OK;
}
// Equal pieces from every rank into every rank's copy of `slots` (slot k = rank k's, `piece` doubles apart; this rank's slot is filled):
// one in-place ncclAllGather, or one broadcast per owner where that entry point is missing / on the hook transport.
static int shard_allgather(oicc_problem* p, double* slots, int64_t piece, hipStream_t st) {
  const int n = p->shard_n, me = p->shard_rank;
  RcclApi& api = rccl_api();
  const bool native = p->rccl_comm != nullptr;
  ncclComm_t comm = static_cast<ncclComm_t>(p->rccl_comm);
  bool ok = true;
  if (native && api.AllGather != nullptr) ok = api.AllGather(slots + int64_t(me) * piece, slots, size_t(piece), ncclDouble, comm, st) == ncclSuccess;
  else {
    if (native) ok = api.GroupStart() == ncclSuccess;
    for (int k = 0; k < n && ok; ++k) {
      double* ptr = slots + int64_t(k) * piece;
      ok = native ? api.Broadcast(ptr, ptr, size_t(piece), ncclDouble, k, comm, st) == ncclSuccess
                  : (p->exchange != nullptr && p->exchange(p->exchange_user, OICC_XCHG_BROADCAST, ptr, piece, ptr, piece, k, st) == 0);
    }
    if (native) ok = (api.GroupEnd() == ncclSuccess) && ok;
  }
  if (!ok) { p->err = "all-gather between the ranks failed"; return OICC_ERR_STATE; }
  return OICC_OK;
}

// ---- distributed linear solve (round 6): see kernels_bcr.hip, "Distributed block cyclic reduction" ----
// Usable when the ranks agreed on the exchange (same cuts everywhere), the geometry is the cyclic reduction's, and every rank owns
// at least one 64-column block: all of it derived from agreed data, so every rank answers alike.
bool dist_solve_usable(oicc_problem* p) {
  oicc_problem::DistSolve& ds = p->dist;
  const oicc_problem::OwnerPlan& op = p->owner;
  if (ds.gen == p->layout_gen) return ds.usable;
  ds.gen = p->layout_gen; ds.usable = false;
  const TangentLayout& tl = p->tl;
  const int n = p->shard_n;
  if (n < 2 || n > 64 || !op.valid || !op.agreed || op.agreed_gen != p->layout_gen || p->opt["distributed_solve"] == 0.0) return false;
  const int algo = int(p->opt["solver_algorithm"]);
  if (!bcr_applicable(tl) || tl.a + 1 > int(p->opt["bcr_max_border"]) || (algo != 0 && algo != 4)) return false;
  const int nblk = (tl.Pb + 63) / 64;
  ds.b0.assign(size_t(n) + 1, 0);
  for (int k = 0; k <= n; ++k) ds.b0[size_t(k)] = k == n ? nblk : op.cut[size_t(k)] / 64;
  int max_loc = 0;
  for (int k = 0; k < n; ++k) { const int c = ds.b0[size_t(k) + 1] - ds.b0[size_t(k)]; if (c < 1 || (k > 0 && op.cut[size_t(k)] % 64 != 0)) return false; max_loc = std::max(max_loc, c); }
  BcrDist& d = ds.d;
  d.nranks = n; d.rank = p->shard_rank; d.b0 = ds.b0[size_t(d.rank)]; d.n_loc = ds.b0[size_t(d.rank) + 1] - d.b0; d.max_loc = max_loc;
  d.ws_doubles = bcr_dist_workspace_doubles(tl, d.n_loc, n);
  d.msg_piece = (bcr_dist_msg_doubles(tl) + 7) / 8 * 8; d.x_piece = (int64_t(max_loc) * 64 + tl.a + 7) / 8 * 8;
  if (!ds.ws.resize(size_t(d.ws_doubles)) || !ds.msg.resize(size_t(d.msg_piece) * size_t(n)) || !ds.xg.resize(size_t(d.x_piece) * size_t(n)) || !ds.d_b0.upload(ds.b0, p->stream)) return false;
  if (hipStreamSynchronize(p->stream) != hipSuccess) return false;   // (b0 may be reassigned)
  d.ws = ds.ws.p; d.msg = ds.msg.p; d.xg = ds.xg.p; d.d_b0 = ds.d_b0.p;
  ds.usable = true;
  return true;
}
int dist_solve(oicc_problem* p, const NormalEq& ne, const SolveBuffers& sb_in, double radius, int reuse_diagonal, double min_diag, double max_diag, hipStream_t st) {
  oicc_problem::DistSolve& ds = p->dist;
  SolveBuffers sb = sb_in; sb.radius = radius;
  if (launch_bcr_dist_forward(ne, p->tl, sb, reuse_diagonal, min_diag, max_diag, ds.d, st) != 0) { p->err = "distributed solve: geometry / workspace"; return OICC_ERR_STATE; }
  int rc = shard_allgather(p, ds.d.msg, ds.d.msg_piece, st); if (rc) return rc;
  if (launch_bcr_dist_middle(p->tl, sb, ds.d, st) != 0) { p->err = "distributed solve: top system"; return OICC_ERR_STATE; }
  rc = shard_allgather(p, ds.d.xg, ds.d.x_piece, st); if (rc) return rc;
  launch_bcr_dist_finish(p->tl, sb, ds.d, st);
  HIPCK(p, hipGetLastError());
  ++ds.solves; ds.last_step_gathered = true;
  return OICC_OK;
}
int lm_solve_any(oicc_problem* p, const NormalEq& ne, const SolveBuffers& sb, double radius, int reuse_diagonal, double min_diag, double max_diag, hipStream_t st) {
  p->dist.last_step_gathered = false;
  if (p->shard_n > 1 && p->reduce != nullptr && dist_solve_usable(p)) return dist_solve(p, ne, sb, radius, reuse_diagonal, min_diag, max_diag, st);
  if (launch_lm_solve(ne, p->tl, sb, radius, reuse_diagonal, min_diag, max_diag, st) != 0) {
    p->err = "band/arrow geometry exceeds the single-workgroup solver (half bandwidth or arrow too large for 160 KB LDS)"; return OICC_ERR_UNSUPPORTED; }
  return OICC_OK;
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
  //     and is added.  ONE pack launch for all peers; native transport: every send and receive in ONE group (both neighbours at once:
  //     round 4 ran one group per peer in rank order, a chain of N - 1 hand-overs).  Blocking hook transport: peers in ascending rank
  //     order on every rank, the lower rank of a pair sends first -- no cyclic wait.
  const int n_send = op.send_off[n], recv0 = op.recv_off[0];
  launch_ne_pack_rows(ne, tl, p->d_xrows.p, n_send, p->d_xsend.p, st);
  if (native) {
    bool ok = api.GroupStart() == ncclSuccess;
    for (int q = 0; q < n && ok; ++q) {
      if (q == me) continue;
      const int ns = op.send_off[q + 1] - op.send_off[q], nr = op.recv_off[q + 1] - op.recv_off[q];
      if (ns) ok = ok && api.Send(p->d_xsend.p + int64_t(op.send_off[q]) * L, size_t(ns) * L, ncclDouble, q, comm, st) == ncclSuccess;
      if (nr) ok = ok && api.Recv(p->d_xrecv.p + int64_t(op.recv_off[q] - recv0) * L, size_t(nr) * L, ncclDouble, q, comm, st) == ncclSuccess;
    }
    ok = (api.GroupEnd() == ncclSuccess) && ok;
    if (!ok) { p->err = "ncclSend / ncclRecv of the halo rows failed"; return OICC_ERR_STATE; }
  }
  for (int q = 0; q < n; ++q) {
    if (q == me) continue;
    const int ns = op.send_off[q + 1] - op.send_off[q], nr = op.recv_off[q + 1] - op.recv_off[q];
    if (ns == 0 && nr == 0) continue;
    double* rbuf = p->d_xrecv.p + int64_t(op.recv_off[q] - recv0) * L;
    if (!native && p->exchange(p->exchange_user, OICC_XCHG_SENDRECV, p->d_xsend.p + int64_t(op.send_off[q]) * L, int64_t(ns) * L, rbuf, int64_t(nr) * L, q, st) != 0) {
      p->err = "exchange callback (send / receive) failed"; return OICC_ERR_STATE; }
    launch_ne_add_rows(ne, tl, p->d_xrows.p + op.recv_off[q], nr, rbuf, st);   // (per peer: two peers may hold the same row of a narrow range)
    moved += int64_t(ns + nr) * L * int64_t(sizeof(double));
  }
  // (2) gather: every rank's owned range as ONE contiguous message of packed rows [band W | arrow a | g] (round 5; round 4 sent the
  //     a + 2 strided pieces of a range as separate broadcasts: 88 collectives per pass at N = 8, a = 9, thousands under POINTS).
  //     Slot k of the gather buffer belongs to rank k; native: one in-place ncclAllGather of equal (padded) slots, or one broadcast
  //     per owner where that entry point is missing; hook transport: one broadcast per owner.
  double* slots = p->d_xgather.p;
  if (dist_solve_usable(p)) {
    // distributed solve (round 6): the band rows stay where they are -- the owner eliminates them; every rank still needs the
    // diagonal (Jacobi scaling, Levenberg-Marquardt diagonal) and the gradient of ALL rows: two doubles per row
    const int64_t piece2 = int64_t(std::max(op.max_owned, 1)) * 2;
    launch_ne_pack_diag_g(ne, tl, op.cut[me], op.cut[me + 1] - op.cut[me], slots + int64_t(me) * piece2, st);
    const int rcg = shard_allgather(p, slots, piece2, st); if (rcg) return rcg;
    moved += int64_t(n - 1) * piece2 * int64_t(sizeof(double));
    launch_ne_unpack_diag_g(ne, tl, p->d_xcut.p, n, me, piece2, slots, st);
    if (tl.a > 0 && p->reduce(p->reduce_user, ne.C(), int64_t(tl.a) * tl.a, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
    if (p->reduce(p->reduce_user, ne.g() + tl.Pb, int64_t(tl.a) + 1, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
    moved += 2 * (int64_t(tl.a) * tl.a + tl.a + 1) * int64_t(sizeof(double));
    if (bytes_moved) *bytes_moved = moved;
    return OICC_OK;
  }
  const int64_t piece = int64_t(std::max(op.max_owned, 1)) * L;
  launch_ne_pack_range(ne, tl, op.cut[me], op.cut[me + 1] - op.cut[me], slots + int64_t(me) * piece, st);
  bool ok = true;
  if (native && api.AllGather != nullptr) {
    ok = api.AllGather(slots + int64_t(me) * piece, slots, size_t(piece), ncclDouble, comm, st) == ncclSuccess;
    moved += int64_t(n - 1) * piece * int64_t(sizeof(double));
  } else {
    if (native) ok = api.GroupStart() == ncclSuccess;
    for (int k = 0; k < n && ok; ++k) {
      const int64_t cnt = int64_t(op.cut[k + 1] - op.cut[k]) * L;
      if (cnt <= 0) continue;
      if (k != me) moved += cnt * int64_t(sizeof(double));
      double* ptr = slots + int64_t(k) * piece;
      ok = native ? api.Broadcast(ptr, ptr, size_t(cnt), ncclDouble, k, comm, st) == ncclSuccess
                  : p->exchange(p->exchange_user, OICC_XCHG_BROADCAST, ptr, cnt, ptr, cnt, k, st) == 0;
    }
    if (native) ok = (api.GroupEnd() == ncclSuccess) && ok;
  }
  if (!ok) { p->err = "gather of the owned band ranges failed"; return OICC_ERR_STATE; }
  launch_ne_unpack_ranges(ne, tl, p->d_xcut.p, n, me, piece, slots, st);
  // (3) what every rank contributes to: the arrow corner, the arrow part of the gradient and the cost
  if (tl.a > 0 && p->reduce(p->reduce_user, ne.C(), int64_t(tl.a) * tl.a, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  if (p->reduce(p->reduce_user, ne.g() + tl.Pb, int64_t(tl.a) + 1, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }   // (the cost follows the gradient in the packed buffer)
  moved += 2 * (int64_t(tl.a) * tl.a + tl.a + 1) * int64_t(sizeof(double));
  if (bytes_moved) *bytes_moved = moved;
  return OICC_OK;
}

}  // namespace oicc

namespace { struct EventPair { hipEvent_t a = nullptr, b = nullptr; ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } }; }   // (destroyed on every exit of the timing entry points)
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
// Debug / measurement entry (outside include/oicc_hip.h; tests, bench.py): the DISTRIBUTED cyclic reduction of `nranks` ranks run by ONE
// process on the unsharded problem -- every rank's forward part in turn (into its slot of the gather buffer: the "all-gather" is a
// no-op in one address space), then every rank's top system + local back substitution, then the step -- and checked against the
// packed normal equations: out[0] = ||M delta - rhs|| / ||rhs||, out[1] = Cholesky failure flag, then per rank r
// out[2 + 2 r] = ms of its forward part, out[3 + 2 r] = ms of its top system + back substitution (HIP events, the rank alone on the
// device: what ONE rank of an N-GPU run spends in the solve besides the two gathers).  Block ranges: equal split of the blocks.
int oicc_debug_dist_solve_emulated(oicc_problem* p, int32_t flags, int32_t nranks, double radius, int32_t repeats, double* out) {
  int rc = prepare(p, flags); if (rc) return rc;
  hipStream_t st = p->stream;
  auto saved = p->reduce; p->reduce = nullptr;
  rc = eval_pass(p, p->d_x.p, true); p->reduce = saved; if (rc) return rc;
  const TangentLayout& tl = p->tl;
  const int nblk = (tl.Pb + 63) / 64;
  ARG(p, out != nullptr && nranks >= 2 && nranks <= 64 && nranks <= nblk && bcr_applicable(tl) && tl.a + 1 <= int(p->opt["bcr_max_border"]), "distributed solve: ranks / geometry");
  SolveBuffers sb = solve_buffers(p); sb.radius = radius;
  launch_lm_scale(p->ne, tl, sb.scale, p->opt["jacobi_scaling"] != 0, st);
  HIPCK(p, hipMemsetAsync(p->d_state.p, 0, sizeof(LmState), st));
  std::vector<int32_t> b0(size_t(nranks) + 1); int max_loc = 0;
  for (int k = 0; k <= nranks; ++k) b0[size_t(k)] = int32_t(int64_t(nblk) * k / nranks);
  for (int k = 0; k < nranks; ++k) max_loc = std::max(max_loc, int(b0[size_t(k) + 1] - b0[size_t(k)]));
  DevBuf<int32_t> d_b0; DevBuf<double> msg, xg; std::vector<DevBuf<double>> ws(static_cast<size_t>(nranks));
  const int64_t msg_piece = (bcr_dist_msg_doubles(tl) + 7) / 8 * 8, x_piece = (int64_t(max_loc) * 64 + tl.a + 7) / 8 * 8;
  if (!d_b0.upload(b0, st) || !msg.resize(size_t(msg_piece) * size_t(nranks)) || !xg.resize(size_t(x_piece) * size_t(nranks))) { p->err = "hipMalloc distributed solve"; return OICC_ERR_HIP; }
  std::vector<BcrDist> ds(static_cast<size_t>(nranks));
  for (int k = 0; k < nranks; ++k) {
    BcrDist& d = ds[size_t(k)];
    d.nranks = nranks; d.rank = k; d.b0 = b0[size_t(k)]; d.n_loc = b0[size_t(k) + 1] - d.b0; d.max_loc = max_loc; d.d_b0 = d_b0.p;
    d.ws_doubles = bcr_dist_workspace_doubles(tl, d.n_loc, nranks);
    if (!ws[size_t(k)].resize(size_t(d.ws_doubles))) { p->err = "hipMalloc distributed solve"; return OICC_ERR_HIP; }
    d.ws = ws[size_t(k)].p; d.msg = msg.p; d.msg_piece = msg_piece; d.xg = xg.p; d.x_piece = x_piece;
  }
  EventPair ev; HIPCK(p, hipEventCreate(&ev.a)); HIPCK(p, hipEventCreate(&ev.b));
  const double min_diag = p->opt["min_lm_diagonal"], max_diag = p->opt["max_lm_diagonal"];
  const int reps = std::max<int>(repeats, 1);
  for (int phase = 0; phase < 2; ++phase)
    for (int k = 0; k < nranks; ++k) {
      for (int it = 0; it < reps + 1; ++it) {     // (the first run is the warm-up; every run leaves the same results)
        if (it == 1) HIPCK(p, hipEventRecord(ev.a, st));
        const int r2 = phase == 0 ? launch_bcr_dist_forward(p->ne, tl, sb, 0, min_diag, max_diag, ds[size_t(k)], st) : launch_bcr_dist_middle(tl, sb, ds[size_t(k)], st);
        if (r2 != 0) { p->err = "distributed solve: geometry / workspace"; return OICC_ERR_STATE; }
      }
      HIPCK(p, hipEventRecord(ev.b, st)); HIPCK(p, hipEventSynchronize(ev.b));
      float ms = 0; (void)hipEventElapsedTime(&ms, ev.a, ev.b);
      out[2 + 2 * k + phase] = double(ms) / reps;
    }
  launch_bcr_dist_finish(tl, sb, ds[0], st);
  DevBuf<double> acc; if (!acc.resize(2 + size_t(tl.a))) return OICC_ERR_HIP;
  launch_lm_solve_residual(p->ne, tl, sb, acc.p, st);
  double h[2] = {0, 0}; LmState hs;
  HIPCK(p, hipMemcpyAsync(h, acc.p, sizeof(h), hipMemcpyDeviceToHost, st));
  HIPCK(p, hipMemcpyAsync(&hs, p->d_state.p, sizeof(hs), hipMemcpyDeviceToHost, st));
  HIPCK(p, hipStreamSynchronize(st));
  out[0] = h[1] > 0.0 ? std::sqrt(h[0] / h[1]) : std::sqrt(h[0]); out[1] = double(hs.chol_failed);
  return OICC_OK;
}
// debug read-out (outside include/oicc_hip.h; tests, bench.py): out4 = [distributed solves run so far, this rank's first block, its block count, ranks]
int oicc_debug_dist_solve_info(const oicc_problem* p, int64_t out4[4]) {
  if (!p || !out4) return OICC_ERR_INVALID_ARG;
  out4[0] = p->dist.solves; out4[1] = p->dist.d.b0; out4[2] = p->dist.d.n_loc; out4[3] = p->dist.usable ? p->dist.d.nranks : 0;
  return OICC_OK;
}


int oicc_time_allreduce(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes) {
  int rc = prepare(p, flags); if (rc) return rc;
  if (!p->reduce) { p->err = "no reduction path installed (oicc_rccl_init / oicc_set_allreduce)"; return OICC_ERR_STATE; }
  hipStream_t st = p->stream;
  HIPCK(p, hipMemsetAsync(p->d_ne2.p, 0, p->ne.total * sizeof(double), st));   // (the second buffer: the current system stays intact)
  EventPair ev; HIPCK(p, hipEventCreate(&ev.a)); HIPCK(p, hipEventCreate(&ev.b)); hipEvent_t e0 = ev.a, e1 = ev.b;
  if (p->reduce(p->reduce_user, p->d_ne2.p, p->ne.total, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }   // warm-up (connection set-up)
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) if (p->reduce(p->reduce_user, p->d_ne2.p, p->ne.total, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_call) *ms_per_call = double(ms) / std::max(repeats, 1);
  if (bytes) *bytes = int64_t(p->ne.total) * int64_t(sizeof(double));
  return OICC_OK;
}

int oicc_time_exchange(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes_moved) {
  int rc = prepare(p, flags); if (rc) return rc;
  if (!owner_exchange_ready(p)) { p->err = "owner-computes exchange not set up (oicc_set_shard, remote measurements with their owners, a transport)"; return OICC_ERR_STATE; }
  if (repeats < 0) return OICC_OK;                                               // (a local question: is the exchange set up? nothing is sent)
  hipStream_t st = p->stream;
  HIPCK(p, hipMemsetAsync(p->d_ne2.p, 0, p->ne.total * sizeof(double), st));   // (the second buffer: the current system stays intact)
  EventPair ev; HIPCK(p, hipEventCreate(&ev.a)); HIPCK(p, hipEventCreate(&ev.b)); hipEvent_t e0 = ev.a, e1 = ev.b;
  int64_t moved = 0;
  rc = owner_exchange(p, p->ne2, st, &moved); if (rc) return rc;                  // warm-up (connection set-up)
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) { rc = owner_exchange(p, p->ne2, st, &moved); if (rc) return rc; }
  HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_call) *ms_per_call = double(ms) / std::max(repeats, 1);
  if (bytes_moved) *bytes_moved = moved;
  return OICC_OK;
}

}  // extern "C"
