// Step-size interpolation of Ceres' Armijo line search (host side of oicc_optimize's bounds line search).
//
// Ceres 2.1.0 (ceres-solver, internal/ceres/line_search.cc: LineSearch::InterpolatingPolynomialMinimizingStepSize,
// polynomial.cc: FindInterpolatingPolynomial / MinimizePolynomial) fits the polynomial through the samples of the search
// (value and slope at 0, at the current and, when there is one, at the previous trial step size) and takes its minimiser
// over [max_step_contraction * x, min_step_contraction * x].  The polynomial is kept here in Newton form on the repeated
// nodes (Hermite divided differences), the minimiser is the best of the two interval ends and the stationary points inside.
// The reference enables the search implicitly: a Ceres problem with parameter bounds (the bias knots, impl.h:206-240) gets
// max_num_line_search_step_size_iterations = 20 Armijo steps along the projected path before every candidate evaluation.
#pragma once
#include <cmath>

namespace oicc {

struct LsSample { double x = 0, value = 0, gradient = 0; bool has_gradient = false; };

struct LsPolynomial {
  static constexpr int kMax = 6;
  double z[kMax], c[kMax]; int m = 0;
  // samples with pairwise different x; a sample with a slope enters as a double node
  bool fit(const LsSample* s, int n) {
    double col[kMax]; bool twin[kMax];
    m = 0;
    for (int i = 0; i < n; ++i) {
      if (m + (s[i].has_gradient ? 2 : 1) > kMax) return false;
      z[m] = s[i].x; col[m] = s[i].value; twin[m] = false; ++m;
      if (s[i].has_gradient) { z[m] = s[i].x; col[m] = s[i].gradient; twin[m] = true; ++m; }
    }
    if (m == 0) return false;
    // order-0 column: every node carries its sample's value; twins remember the slope for order 1
    double val[kMax], slope[kMax];
    for (int i = 0; i < m; ++i) { if (twin[i]) { val[i] = val[i - 1]; slope[i] = col[i]; } else { val[i] = col[i]; slope[i] = 0.0; } }
    double dd[kMax];
    for (int i = 0; i < m; ++i) dd[i] = val[i];
    c[0] = dd[0];
    for (int k = 1; k < m; ++k) {
      for (int i = m - 1; i >= k; --i) {
        const double den = z[i] - z[i - k];
        if (k == 1 && twin[i]) dd[i] = slope[i];
        else { if (den == 0.0) return false; dd[i] = (dd[i] - dd[i - 1]) / den; }
      }
      c[k] = dd[k];
    }
    for (int k = 0; k < m; ++k) if (!std::isfinite(c[k])) return false;
    return true;
  }
  void eval(double x, double* p, double* dp) const {
    double v = c[m - 1], d = 0.0;
    for (int k = m - 2; k >= 0; --k) { d = d * (x - z[k]) + v; v = v * (x - z[k]) + c[k]; }
    *p = v; *dp = d;
  }
  double minimum(double lo, double hi) const {
    double pl, ph, d;
    eval(lo, &pl, &d); eval(hi, &ph, &d);
    double best = lo, bv = pl;
    if (ph < bv) { best = hi; bv = ph; }
    if (m < 3) return best;
    const int G = 2048;
    double xp = lo, fp, v; eval(lo, &v, &fp);
    for (int i = 1; i <= G; ++i) {
      const double xn = lo + (hi - lo) * double(i) / G; double fn; eval(xn, &v, &fn);
      if ((fp <= 0.0 && fn >= 0.0) || (fp >= 0.0 && fn <= 0.0)) {
        double a = xp, b = xn, fa = fp;
        for (int k = 0; k < 80; ++k) { const double mid = 0.5 * (a + b); double fm; eval(mid, &v, &fm); if ((fa <= 0.0) == (fm <= 0.0)) { a = mid; fa = fm; } else b = mid; }
        const double r = 0.5 * (a + b); double pv; eval(r, &pv, &d);
        if (pv < bv) { bv = pv; best = r; }
      }
      xp = xn; fp = fn;
    }
    return best;
  }
};

// Next trial step size: minimiser of the interpolating polynomial on [1e-3 x, 0.6 x] (Ceres' max / min step contraction);
// a trial without a finite cost is halved.
inline double ls_next_step_size(const LsSample& initial, const LsSample* previous, const LsSample& current) {
  const double lo = 1e-3 * current.x, hi = 0.6 * current.x;
  if (!std::isfinite(current.value)) return std::fmin(std::fmax(0.5 * current.x, lo), hi);
  LsSample s[3]; int n = 0;
  s[n++] = initial; s[n++] = current;
  if (!std::isfinite(s[1].gradient)) s[1].has_gradient = false;
  if (previous != nullptr && std::isfinite(previous->value)) { s[n] = *previous; if (!std::isfinite(s[n].gradient)) s[n].has_gradient = false; ++n; }
  LsPolynomial poly;
  if (!poly.fit(s, n)) return hi;
  return poly.minimum(lo, hi);
}

}  // namespace oicc
