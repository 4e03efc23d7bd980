// The trust-region decision of one Levenberg-Marquardt iteration on the device (LmCtl, oicc_device.h): a pure function of the
// control state, the step's scalars (LmState) and the candidate's cost -- evaluated by the one-thread lm_decide_kernel
// (kernels_solve.hip: after the last enqueued iteration) and, folded, by EVERY workgroup of the next iteration's build kernel
// (kernels_bcr.hip / kernels_solve.hip: all of them derive the same next state, one of them stores it).
// Mirrors the host loop of oicc_optimize = TrustRegionMinimizer::Minimize [EXT Ceres 2.1.0] with the reference's options:
// invalid step -> shrink; parameter / function tolerance; IsStepSuccessful -> swap the buffers, StepAccepted(rho); else
// StepRejected; then the tests made before the next iteration (iterations, radius, gradient tolerance of an accepted point).
#pragma once
#include <hip/hip_runtime.h>
#include "oicc_device.h"

namespace oicc {

// c: the state the iteration ran with -> the state the next one runs with (seq + 1).  rec / push: the iteration record and whether it
// enters the trace (c.trace_n is already advanced).  No memory is written.
__device__ __forceinline__ void lm_decide_compute(LmCtl& c, const LmState& hs, double cand_cost, LmIterRec* rec_out, bool* push_out) {
  const double cost = c.cost;
  double radius = c.radius;
  int done = LM_RUNNING;
  bool push = true;
  LmIterRec rec;
  const int iter = c.iter + 1;
  c.iter = iter;
  const double model_cost_change = hs.model_cost_change;
  const bool ok = hs.chol_failed == 0 && isfinite(model_cost_change) && isfinite(hs.step_norm_sq) && model_cost_change > 0.0;
  if (c.hold) {          // benchmark: the full decision arithmetic, no state change
    const double rel = (cost - cand_cost) / model_cost_change;
    rec = LmIterRec{iter, ok && rel > c.min_rel_dec ? 1 : 0, cand_cost, cost - cand_cost, hs.gradient_max_norm, sqrt(hs.step_norm_sq), rel, radius};
    if (!ok) done = LM_DONE_INVALID_STEPS;   // (the benchmark's system must stay solvable)
  } else if (!ok) {         // invalid step: LINEAR_SOLVER_FAILURE or a non-positive model decrease
    c.invalid += 1;
    if (c.invalid >= c.max_invalid) { done = LM_DONE_INVALID_STEPS; rec = LmIterRec{iter, 0, cost, 0.0, c.gmax, 0.0, 0.0, radius}; push = false; }
    else {
      radius /= c.decrease_factor; c.decrease_factor *= 2.0; c.reuse_diagonal = 1; c.num_unsuccessful += 1;
      rec = LmIterRec{iter, 0, cost, 0.0, c.gmax, 0.0, 0.0, radius};
    }
  } else {
    c.invalid = 0;
    const double x_norm = sqrt(hs.x_norm_sq), step_norm = sqrt(hs.step_norm_sq);
    const double cost_change = cost - cand_cost, rel_dec = cost_change / model_cost_change;
    if (step_norm <= c.ptol * (x_norm + c.ptol)) { done = LM_DONE_PARAMETER_TOL; rec = LmIterRec{iter, 0, cost, cost_change, c.gmax, step_norm, rel_dec, radius}; }
    else if (fabs(cost_change) <= c.ftol * cost) { done = LM_DONE_FUNCTION_TOL; rec = LmIterRec{iter, 0, cost, cost_change, c.gmax, step_norm, rel_dec, radius}; }
    else if (rel_dec > c.min_rel_dec) {   // IsStepSuccessful: the candidate and its normal equations become current
      double* t = c.xp[0]; c.xp[0] = c.xp[1]; c.xp[1] = t;
      t = c.nep[0]; c.nep[0] = c.nep[1]; c.nep[1] = t;
      t = c.segp[0]; c.segp[0] = c.segp[1]; c.segp[1] = t;
      c.cost = cand_cost; c.gmax = hs.gradient_max_norm; c.num_successful += 1;
      const double q = 2.0 * rel_dec - 1.0;
      radius = fmin(c.max_radius, radius / fmax(1.0 / 3.0, 1.0 - q * q * q));
      c.decrease_factor = 2.0; c.reuse_diagonal = 0;
      rec = LmIterRec{iter, 1, cand_cost, cost_change, hs.gradient_max_norm, step_norm, rel_dec, radius};
    } else {
      radius /= c.decrease_factor; c.decrease_factor *= 2.0; c.reuse_diagonal = 1; c.num_unsuccessful += 1;
      rec = LmIterRec{iter, 0, cost, cost_change, c.gmax, step_norm, rel_dec, radius};
    }
  }
  c.radius = radius;
  if (push && c.trace_n < c.trace_cap) c.trace_n += 1; else push = false;
  if (done == LM_RUNNING && !c.hold) {   // what the host loop tests before it starts the next iteration, in its order
    if (iter >= c.max_iters) done = LM_DONE_MAX_ITERATIONS;
    else if (radius <= c.min_radius) done = LM_DONE_MIN_RADIUS;
    else if (rec.step_is_successful && c.gmax <= c.gtol) done = LM_DONE_GRADIENT_TOL;
  }
  c.done = done;
  c.seq += 1;
  *rec_out = rec; *push_out = push;
}
// ONE thread of the grid: the side effects of a decision -- the next state, the iteration record, the time stamp, the host's word
__device__ __forceinline__ void lm_decide_publish(LmCtl* out, const LmCtl& c, const LmIterRec& rec, bool push) {
  *out = c;
  if (push && c.trace != nullptr) c.trace[c.trace_n - 1] = rec;
  if (c.stamps != nullptr && c.seq - 1 < c.trace_cap) c.stamps[3 * (c.seq - 1) + 2] = wall_clock64();
  if (c.host != nullptr)   // the host polls this behind the iterations it has enqueued; it reads nothing else the device wrote, so no release fence
    __hip_atomic_store(&c.host->word, ((long long)c.done << 32) | (c.seq & 0xffffffffll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The state an iteration's build kernel runs with.  prev == nullptr: *cur as the host (or an earlier kernel) left it.  Else every
// workgroup derives it from the previous iteration's state and results; `writer` (one thread of the grid) also stores it.
// Called by all threads of the workgroup; returns false when the loop is done (the caller returns).
__device__ __forceinline__ bool lm_ctl_next_state(const SolveBuffers& sb, bool writer, LmCtl* s_c /* __shared__ */) {
  if (threadIdx.x == 0) {
    if (sb.ctl_prev == nullptr) *s_c = *sb.ctl;
    else {
      LmCtl c = *sb.ctl_prev;
      LmIterRec rec; bool push = false; const bool decide = c.done == 0;
      if (decide) { const LmState hs = *sb.st_prev; lm_decide_compute(c, hs, c.nep[1][sb.off_cost], &rec, &push); }
      *s_c = c;
      if (writer) { if (decide) lm_decide_publish(const_cast<LmCtl*>(sb.ctl), c, rec, push); else *const_cast<LmCtl*>(sb.ctl) = c; }
    }
  }
  __syncthreads();
  return s_c->done == 0;
}

}  // namespace oicc
