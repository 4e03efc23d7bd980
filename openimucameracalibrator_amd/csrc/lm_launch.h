// Launchers of the problem-independent part of one Levenberg-Marquardt step (Jacobi scaling, gradient norm,
// damped system, band + arrow linear solve); shared by the spline problem (oicc_problem.hip) and view bundle
// adjustment (oicc_ba.hip).  Host-side declarations only.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include <utility>
#include <cstring>
#include "oicc_device.h"

namespace oicc {
// kernels_solve.hip
int64_t solve_workspace_doubles(const TangentLayout& tl);
void launch_lm_scale(const NormalEq& ne, const TangentLayout& tl, double* scale, int jacobi, hipStream_t st);
void launch_lm_gradmax(const NormalEq& ne, int P, LmState* s, hipStream_t st);
void launch_lm_build(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal, double min_diag,
                     double max_diag, hipStream_t st);
int launch_band_arrow_cholesky(const TangentLayout& tl, const SolveBuffers& sb, hipStream_t st);
void launch_lm_solve_residual(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, double* acc, hipStream_t st);
int64_t bcr_workspace_doubles(const TangentLayout& tl);
int launch_bcr_solve(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal, double min_diag,
                     double max_diag, hipStream_t st);
bool bcr_applicable(const TangentLayout& tl);
// distributed block cyclic reduction (kernels_bcr.hip): rank-local forward part / top system + local back substitution / the gathered step
int64_t bcr_dist_workspace_doubles(const TangentLayout& tl, int n_loc, int nranks);
int64_t bcr_dist_msg_doubles(const TangentLayout& tl);
int launch_bcr_dist_forward(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal, double min_diag, double max_diag, const BcrDist& d, hipStream_t st);
int launch_bcr_dist_middle(const TangentLayout& tl, const SolveBuffers& sb, const BcrDist& d, hipStream_t st);
void launch_bcr_dist_finish(const TangentLayout& tl, const SolveBuffers& sb, const BcrDist& d, hipStream_t st);
// damped system + factorisation + solve (solution in sb.step_s): block cyclic reduction when the
// geometry allows (hb <= 64, arrow <= 63 columns), else the time-partitioned band sweep
static inline int launch_lm_solve(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb_in, double radius, int reuse_diagonal,
                                  double min_diag, double max_diag, hipStream_t st) {
  SolveBuffers sb = sb_in; sb.radius = radius;
  if (sb.algo != 1 && launch_bcr_solve(ne, tl, sb, reuse_diagonal, min_diag, max_diag, st) == 0) return 0;
  if (sb.algo >= 2 && tl.Pb > 0) return -1;
  launch_lm_build(ne, tl, sb, reuse_diagonal, min_diag, max_diag, st);
  return launch_band_arrow_cholesky(tl, sb, st);
}

// device buffer that grows on demand (host-side RAII)
template <class T>
struct DevBuf {
  T* p = nullptr; size_t n = 0;
  bool owned = true;   // false: a piece of a DevArena block
  ~DevBuf() { if (p && owned) (void)hipFree(p); }
  void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
  bool resize(size_t count) {
    if (count <= n && p) return true;
    release();
    if (count == 0) return true;
    if (hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)) != hipSuccess) { p = nullptr; return false; }
    n = count; return true;
  }
  bool upload(const std::vector<T>& h, hipStream_t st) {
    if (!resize(std::max<size_t>(h.size(), 1))) return false;
    if (h.empty()) return true;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st) == hipSuccess;
  }
  void attach(T* ptr, size_t count) { release(); p = ptr; n = count; owned = false; }
};

// Many host arrays -> ONE device block with ONE copy (a hipMalloc + hipMemcpy per array costs 10-20 us each: 0.3 ms for the ~30
// measurement arrays of the GoPro9 configuration).  add() registers an array, commit() packs them into a host staging block
// (256-byte aligned pieces), grows the device block if needed, copies once and points every DevBuf at its piece.
struct DevArena {
  struct Item { const void* src; size_t bytes, off; void (*attach)(void* buf, void* dev, size_t count); void* buf; size_t count; size_t rbytes = 0; void (*detach)(void* buf) = nullptr; };   // rbytes: size of a piece without host data
  std::vector<Item> items; std::vector<unsigned char> stage; unsigned char* dev = nullptr; size_t cap = 0;
  std::vector<std::pair<void*, void (*)(void*)>> attached;   // the buffers the last commit() pointed into this block, and how to detach one (p = nullptr, n = 0)
  ~DevArena() { if (dev) (void)hipFree(dev); }
  template <class T>
  void add(DevBuf<T>& b, const std::vector<T>& h) {
    items.push_back(Item{h.data(), h.size() * sizeof(T), 0, [](void* buf, void* d, size_t c) { static_cast<DevBuf<T>*>(buf)->attach(static_cast<T*>(d), c); }, &b, std::max<size_t>(h.size(), 1)});
    items.back().detach = [](void* buf) { DevBuf<T>* q = static_cast<DevBuf<T>*>(buf); if (!q->owned) { q->p = nullptr; q->n = 0; q->owned = true; } };
  }
  template <class T>
  void reserve(DevBuf<T>& b, size_t count) {   // a piece without host data (workspace)
    items.push_back(Item{nullptr, 0, 0, [](void* buf, void* d, size_t c) { static_cast<DevBuf<T>*>(buf)->attach(static_cast<T*>(d), c); }, &b, std::max<size_t>(count, 1)});
    items.back().rbytes = std::max<size_t>(count, 1) * sizeof(T);
    items.back().detach = [](void* buf) { DevBuf<T>* q = static_cast<DevBuf<T>*>(buf); if (!q->owned) { q->p = nullptr; q->n = 0; q->owned = true; } };
  }
  bool commit(hipStream_t st) {   // (the staging block outlives the asynchronous copy: it belongs to the arena)
    size_t total = 0;
    size_t copy_bytes = 0;   // the pieces with host data come first: one copy covers them
    for (Item& it : items) if (!it.rbytes) { it.off = total; total += (std::max<size_t>(it.bytes, 8) + 255) & ~size_t(255); }
    copy_bytes = total;
    for (Item& it : items) if (it.rbytes) { it.off = total; total += (it.rbytes + 255) & ~size_t(255); }
    if (total > cap) {
      if (dev) (void)hipFree(dev);
      dev = nullptr; cap = 0;
      if (hipMalloc(reinterpret_cast<void**>(&dev), total) != hipSuccess) { dev = nullptr; return false; }
      cap = total;
    }
    // small arrays travel together through the staging block; a large one goes straight from where it lies (no second host copy)
    constexpr size_t kDirect = size_t(1) << 20;
    stage.resize(copy_bytes);
    size_t staged_end = 0;
    for (const Item& it : items) if (it.bytes && it.bytes < kDirect) { std::memcpy(stage.data() + it.off, it.src, it.bytes); staged_end = std::max(staged_end, it.off + it.bytes); }
    for (const Item& it : items) if (it.bytes >= kDirect && hipMemcpyAsync(dev + it.off, it.src, it.bytes, hipMemcpyHostToDevice, st) != hipSuccess) return false;
    // (the staged pieces may be interleaved with direct ones: copy the runs of staged pieces)
    {
      size_t run0 = 0; bool open = false; size_t run1 = 0;
      for (const Item& it : items) {
        if (it.rbytes) continue;
        const bool staged = it.bytes < kDirect;
        if (staged) { if (!open) { run0 = it.off; open = true; } run1 = it.off + ((std::max<size_t>(it.bytes, 8) + 255) & ~size_t(255)); }
        else if (open) { if (hipMemcpyAsync(dev + run0, stage.data() + run0, run1 - run0, hipMemcpyHostToDevice, st) != hipSuccess) return false; open = false; }
      }
      if (open && hipMemcpyAsync(dev + run0, stage.data() + run0, run1 - run0, hipMemcpyHostToDevice, st) != hipSuccess) return false;
    }
    (void)staged_end;
    // a buffer that pointed into this block after the LAST commit and is not part of this one would keep a pointer into space that
    // is laid out anew (the advisor's finding: d_tl_pts is only registered under POINTS): detach it first
    for (const auto& a : attached) { bool again = false; for (const Item& it : items) again = again || it.buf == a.first; if (!again) a.second(a.first); }
    attached.clear();
    for (const Item& it : items) { it.attach(it.buf, dev + it.off, it.count); attached.emplace_back(it.buf, it.detach); }
    items.clear();
    return true;
  }
};
}  // namespace oicc
