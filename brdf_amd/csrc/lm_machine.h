// lm_machine.h -- resumable Levenberg-Marquardt drivers for the MI355X BRDF fitter.
//
// The reference runs dlevmar_dif / dlevmar_bc_dif as ordinary CPU loops that call back into the
// model once per evaluation (levmar/lm_core.c:438-842, levmar/lmbc_core.c:369-1129).  On the GPU the
// n-sized work of every evaluation is a data-parallel *pass* (one kernel launch in the streamed
// regime, one block/wave-wide sweep in the batched regime) that ends in a handful of reduced sums.
// The scalar algorithm between two passes is therefore written here as a state machine:
//
//     machine.start(...)            -> machine.req describes the first pass
//     machine.step(sums, maxabs)    -> consumes the reduced sums of the pass just executed and
//                                      leaves the next pass in machine.req (RQ_DONE when finished)
//
// The machine never sees a sample; it holds ~100 doubles and is cheap enough to be re-executed
// redundantly by every workgroup (streamed regime) or by every lane of the fit's wave/block
// (batched regime), which removes every intra-launch hand-off from the design.
//
// Arithmetic of the scalar steps (damping update, Crout LU with implicit scaling, projection,
// Schnabel line search, projected-gradient fallback, Broyden bookkeeping, info[] layout, stop codes)
// follows the reference so that trajectories agree; the cited lines say which part is mirrored.
// Compiles as plain C++ too (tests/cpp drives it on the host against the oracle).
#pragma once

#include <float.h>
#include <math.h>

#if defined(__HIPCC__)
#define LM_HD __host__ __device__ __forceinline__
#else
#define LM_HD inline
#endif

namespace brdf {

// ---- constants of the reference (lm.c:35-36, lmbc.c:35-38, levmar.h:95-100, lmbc_core.c:430-433)
constexpr double kEpsilon = 1E-12;
constexpr double kOneThird = 0.3333333334;
constexpr int kLsItMax = 150;
constexpr double kLsPow = 2.1;
constexpr double kInitMu = 1E-03;
constexpr double kStopThresh = 1E-17;
constexpr double kDiffDelta = 1E-06;
constexpr int kInfoSz = 10;
constexpr int kLmError = -1;

enum ReqKind : int {
  RQ_DONE = 0,
  RQ_EVAL = 1,       // sums[0] = sum (x-f(p))^2 ; maxabs = max |x-f(p)|
  RQ_SCALED = 2,     // sums[0] = sum ((x-f(p))/scal)^2           (overflow re-check, lmbc_core.c:146-170)
  RQ_JAC = 3,        // FD Jacobian at p, nothing stored: sums = [JtJ lower (row-major), Jte, sum e^2]
  RQ_DIF_INIT = 4,   // hx[sel_hx] <- f(p); sums[0] = sum e^2
  RQ_DIF_JAC = 5,    // FD Jacobian at p from hx[sel_hx]; J[sel_j] <- J; sums = [JtJ lower, Jte]
  RQ_DIF_TRIAL = 6   // hx[!sel_hx] <- f(q); J[!sel_j] <- Broyden(J[sel_j]); sums = [sum e_new^2,
                     //   JnTJn lower, JnT e_new, JnT e_old]
};

template <int M>
struct Request {
  int kind;
  int central;  // RQ_JAC / RQ_DIF_JAC: 0 forward, 1 central differences
  int sel_hx;   // which of the two hx buffers holds f(current p)     (dif only)
  int sel_j;    // which of the two Jacobian buffers is current       (dif only)
  double p[M];  // evaluation point / Jacobian base point
  double d[M];  // finite-difference steps                            (misc_core.c:155-158)
  double q[M];  // trial point p+Dp                                   (RQ_DIF_TRIAL)
  double dp[M]; // Dp                                                 (RQ_DIF_TRIAL)
  double dp_l2; // ||Dp||^2
  double scal;  // RQ_SCALED divisor
};

template <int M>
struct SumLayout {
  static constexpr int NL = M * (M + 1) / 2;
  static constexpr int JAC = NL + M + 1;          // RQ_JAC
  static constexpr int DIF_JAC = NL + M;          // RQ_DIF_JAC
  static constexpr int DIF_TRIAL = 1 + NL + 2 * M;  // RQ_DIF_TRIAL
  static constexpr int MAX = DIF_TRIAL > JAC ? DIF_TRIAL : JAC;
};

LM_HD double lm_abs(double v) { return (v >= 0.0) ? v : -v; }
LM_HD bool lm_finite(double v) { return (v - v) == 0.0; }  // false for NaN and +-Inf

// Crout LU with implicit row scaling + partial pivoting and the DBL_EPSILON zero-pivot rule, then
// forward/back substitution: Axb_core.c:1197-1270.  A, B untouched; returns 0 if a row of A is zero.
template <int M>
LM_HD int lu_solve(const double *A, const double *B, double *x) {
  double a[M * M], scale[M];
  int perm[M];
  for (int i = 0; i < M * M; ++i) a[i] = A[i];
  for (int i = 0; i < M; ++i) x[i] = B[i];
  for (int i = 0; i < M; ++i) {
    double big = 0.0;
    for (int j = 0; j < M; ++j) {
      const double t = lm_abs(a[i * M + j]);
      if (t > big) big = t;
    }
    if (big == 0.0) return 0;
    scale[i] = 1.0 / big;
  }
  for (int j = 0; j < M; ++j) {
    int pivot = -1;
    double big = 0.0;
    for (int i = 0; i < j; ++i) {
      double s = a[i * M + j];
      for (int k = 0; k < i; ++k) s -= a[i * M + k] * a[k * M + j];
      a[i * M + j] = s;
    }
    for (int i = j; i < M; ++i) {
      double s = a[i * M + j];
      for (int k = 0; k < j; ++k) s -= a[i * M + k] * a[k * M + j];
      a[i * M + j] = s;
      const double t = scale[i] * lm_abs(s);
      if (t >= big) {
        big = t;
        pivot = i;
      }
    }
    if (pivot < 0) pivot = j;  // only reachable with NaN input; keeps indexing in range
    if (j != pivot) {
      for (int k = 0; k < M; ++k) {
        const double t = a[pivot * M + k];
        a[pivot * M + k] = a[j * M + k];
        a[j * M + k] = t;
      }
      scale[pivot] = scale[j];
    }
    perm[j] = pivot;
    if (a[j * M + j] == 0.0) a[j * M + j] = DBL_EPSILON;
    if (j != M - 1) {
      const double t = 1.0 / a[j * M + j];
      for (int i = j + 1; i < M; ++i) a[i * M + j] *= t;
    }
  }
  int first = 0;
  for (int i = 0; i < M; ++i) {
    const int j = perm[i];
    double s = x[j];
    x[j] = x[i];
    if (first != 0) {
      for (int jj = first - 1; jj < i; ++jj) s -= a[i * M + jj] * x[jj];
    } else if (s != 0.0) {
      first = i + 1;
    }
    x[i] = s;
  }
  for (int i = M - 1; i >= 0; --i) {
    double s = x[i];
    for (int j = i + 1; j < M; ++j) s -= a[i * M + j] * x[j];
    x[i] = s / a[i * M + i];
  }
  return 1;
}

// covar = sumsq/(n-M) * inverse(JtJ) by the same LU, column by column: misc_core.c:426-591.
template <int M>
LM_HD int lu_covar(const double *JtJ, double *C, double sumsq, int n) {
  for (int l = 0; l < M; ++l) {
    double rhs[M], col[M];
    for (int i = 0; i < M; ++i) rhs[i] = (i == l) ? 1.0 : 0.0;
    if (!lu_solve<M>(JtJ, rhs, col)) return 0;
    for (int i = 0; i < M; ++i) C[i * M + l] = col[i];
  }
  const double fact = sumsq / (double)(n - M);
  for (int i = 0; i < M * M; ++i) C[i] *= fact;
  return M;
}

template <int M>
LM_HD void fd_steps(const double *p, double delta, double *d) {  // misc_core.c:155-158
  for (int j = 0; j < M; ++j) {
    double s = 1E-04 * p[j];
    s = lm_abs(s);
    if (s < delta) s = delta;
    d[j] = s;
  }
}

template <int M>
LM_HD void unpack_lower(const double *s, double *jtj) {
  int c = 0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j <= i; ++j, ++c) {
      jtj[i * M + j] = s[c];
      jtj[j * M + i] = s[c];
    }
}

struct FitOptions {
  double tau, eps1, eps2, eps2sq, eps3, delta;
  int forward;
};

// opts == NULL selects the defaults of levmar.h:98-100 (lm_core.c:507-526)
LM_HD FitOptions make_options(const double *opts) {
  FitOptions o;
  if (opts) {
    o.tau = opts[0];
    o.eps1 = opts[1];
    o.eps2 = opts[2];
    o.eps2sq = opts[2] * opts[2];
    o.eps3 = opts[3];
    o.delta = opts[4];
  } else {
    o.tau = kInitMu;
    o.eps1 = kStopThresh;
    o.eps2 = kStopThresh;
    o.eps2sq = kStopThresh * kStopThresh;
    o.eps3 = kStopThresh;
    o.delta = kDiffDelta;
  }
  o.forward = 1;
  if (o.delta < 0.0) {
    o.delta = -o.delta;
    o.forward = 0;
  }
  return o;
}

// =================================================================================================
// dlevmar_dif: unconstrained LM, forward/central FD Jacobian refreshed lazily, Broyden rank-one
// updates in between (lm_core.c:438-842).  The Broyden update and the J^T J / J^T e of the updated
// Jacobian are produced speculatively by the trial pass (RQ_DIF_TRIAL) so that one pass per LM
// iteration suffices; the machine decides afterwards which of the speculative results are live.
// =================================================================================================
template <int M>
struct DifMachine {
  enum Phase : int { D_INIT_EVAL = 1, D_ITER_TOP, D_AFTER_JAC, D_GRADIENT, D_SOLVE, D_AFTER_TRIAL, D_REJECT, D_FINISH, D_DONE };
  FitOptions o;
  int itmax, n, want_covar;
  int phase, k, stop, nu, nfev, njap, nlss, updjac, updp, newjac, refresh;
  int sel_hx, sel_j;
  double p[M], mu, p_e2, init_e2, jte_inf, p_l2, dp_l2, pdp_e2;
  double jtj[M * M], jte[M], diag[M], dp[M], pdp[M];
  double spec_jtj[M * M], spec_jte[M];  // normal equations of the Broyden-updated J, adopted lazily
  double info[kInfoSz], covar[M * M];
  int ret;
  Request<M> req;

  LM_HD void start(const double *p0, int n_, int itmax_, const double *opts, int want_covar_) {
    o = make_options(opts);
    itmax = itmax_;
    n = n_;
    want_covar = want_covar_;
    k = 0;
    stop = 0;
    nfev = njap = nlss = 0;
    updjac = 0;
    updp = 1;
    newjac = 0;
    refresh = (M >= 10) ? M : 10;  // "K", lm_core.c:495
    sel_hx = sel_j = 0;
    mu = jte_inf = p_l2 = 0.0;
    p_e2 = init_e2 = pdp_e2 = 0.0;
    dp_l2 = DBL_MAX;
    ret = kLmError;
    for (int i = 0; i < M; ++i) {
      p[i] = p0[i];
      jte[i] = diag[i] = dp[i] = pdp[i] = 0.0;
      spec_jte[i] = 0.0;
    }
    for (int i = 0; i < M * M; ++i) jtj[i] = spec_jtj[i] = covar[i] = 0.0;
    for (int i = 0; i < kInfoSz; ++i) info[i] = 0.0;
    clear_req();
    if (n < M) {  // lm_core.c:502-505
      phase = D_DONE;
      req.kind = RQ_DONE;
      return;
    }
    req.kind = RQ_DIF_INIT;
    for (int i = 0; i < M; ++i) req.p[i] = p[i];
    phase = D_INIT_EVAL;
  }

  LM_HD void clear_req() {
    req.kind = RQ_DONE;
    req.central = 0;
    req.sel_hx = sel_hx;
    req.sel_j = sel_j;
    req.dp_l2 = 0.0;
    req.scal = 1.0;
    for (int i = 0; i < M; ++i) req.p[i] = req.d[i] = req.q[i] = req.dp[i] = 0.0;
  }

  LM_HD void gradient_stats() {  // lm_core.c:657-662
    p_l2 = jte_inf = 0.0;
    for (int i = 0; i < M; ++i) {
      const double t = lm_abs(jte[i]);
      if (jte_inf < t) jte_inf = t;
      diag[i] = jtj[i * M + i];
      p_l2 += p[i] * p[i];
    }
  }

  LM_HD void step(const double *s, double /*maxabs*/) {
    for (;;) {
      switch (phase) {
      case D_INIT_EVAL:  // lm_core.c:551-564
        nfev = 1;
        p_e2 = s[0];
        init_e2 = p_e2;
        if (!lm_finite(p_e2)) stop = 7;
        nu = 20;
        phase = D_ITER_TOP;
        break;

      case D_ITER_TOP:
        if (!(k < itmax && !stop)) {
          phase = D_FINISH;
          break;
        }
        if (p_e2 <= o.eps3) {
          stop = 6;
          phase = D_FINISH;
          break;
        }
        if ((updp && nu > 16) || updjac == refresh) {  // fresh FD Jacobian, lm_core.c:578-588
          clear_req();
          req.kind = RQ_DIF_JAC;
          req.central = !o.forward;
          for (int i = 0; i < M; ++i) req.p[i] = p[i];
          fd_steps<M>(p, o.delta, req.d);
          ++njap;
          nfev += o.forward ? M : 2 * M;
          nu = 2;
          updjac = 0;
          updp = 0;
          newjac = 1;
          phase = D_AFTER_JAC;
          return;
        }
        phase = D_GRADIENT;
        break;

      case D_AFTER_JAC:
        unpack_lower<M>(s, jtj);
        for (int i = 0; i < M; ++i) jte[i] = s[SumLayout<M>::NL + i];
        newjac = 0;
        gradient_stats();
        phase = D_SOLVE;
        break;

      case D_GRADIENT:
        if (newjac) {  // lm_core.c:590-664 with the sums the trial pass produced for the updated J
          newjac = 0;
          for (int i = 0; i < M * M; ++i) jtj[i] = spec_jtj[i];
          for (int i = 0; i < M; ++i) jte[i] = spec_jte[i];
          gradient_stats();
        }
        phase = D_SOLVE;
        break;

      case D_SOLVE: {
        if (jte_inf <= o.eps1) {  // lm_core.c:676-680
          dp_l2 = 0.0;
          stop = 1;
          phase = D_FINISH;
          break;
        }
        if (k == 0) {  // lm_core.c:683-687
          double t = -DBL_MAX;
          for (int i = 0; i < M; ++i)
            if (diag[i] > t) t = diag[i];
          mu = o.tau * t;
        }
        for (int i = 0; i < M; ++i) jtj[i * M + i] += mu;
        const int solved = lu_solve<M>(jtj, jte, dp);
        ++nlss;
        if (!solved) {
          phase = D_REJECT;
          break;
        }
        dp_l2 = 0.0;
        for (int i = 0; i < M; ++i) {
          const double t = dp[i];
          pdp[i] = p[i] + t;
          dp_l2 += t * t;
        }
        if (dp_l2 <= o.eps2sq * p_l2) {
          stop = 2;
          phase = D_FINISH;
          break;
        }
        if (dp_l2 >= (p_l2 + o.eps2) / (kEpsilon * kEpsilon)) {
          stop = 4;
          phase = D_FINISH;
          break;
        }
        clear_req();
        req.kind = RQ_DIF_TRIAL;
        for (int i = 0; i < M; ++i) {
          req.p[i] = p[i];
          req.q[i] = pdp[i];
          req.dp[i] = dp[i];
        }
        req.dp_l2 = dp_l2;
        ++nfev;
        phase = D_AFTER_TRIAL;
        return;
      }

      case D_AFTER_TRIAL: {  // lm_core.c:742-790
        pdp_e2 = s[0];
        if (!lm_finite(pdp_e2)) {
          stop = 7;
          phase = D_FINISH;
          break;
        }
        const double dF = p_e2 - pdp_e2;
        const bool updated = (updp || dF > 0);
        if (updated) {  // adopt the speculatively updated Jacobian
          sel_j ^= 1;
          ++updjac;
          newjac = 1;
        }
        double dL = 0.0;
        for (int i = 0; i < M; ++i) dL += dp[i] * (mu * dp[i] + jte[i]);
        const bool accepted = (dL > 0.0 && dF > 0.0);
        if (updated) {  // keep the updated Jacobian's products, paired with the residual that stays live;
                        // they replace jtj/jte at the top of the next iteration, as in the reference
          unpack_lower<M>(s + 1, spec_jtj);
          const double *g = s + 1 + SumLayout<M>::NL + (accepted ? 0 : M);
          for (int i = 0; i < M; ++i) spec_jte[i] = g[i];
        }
        if (accepted) {
          double t = (2.0 * dF / dL - 1.0);
          t = 1.0 - t * t * t;
          mu = mu * ((t >= kOneThird) ? t : kOneThird);
          nu = 2;
          for (int i = 0; i < M; ++i) p[i] = pdp[i];
          sel_hx ^= 1;  // e, hx <- trial values
          p_e2 = pdp_e2;
          updp = 1;
          ++k;
          phase = D_ITER_TOP;
          break;
        }
        phase = D_REJECT;
        break;
      }

      case D_REJECT: {  // lm_core.c:797-806
        mu *= nu;
        const int nu2 = (int)((unsigned)nu << 1);
        if (nu2 <= nu) {
          stop = 5;
          phase = D_FINISH;
          break;
        }
        nu = nu2;
        for (int i = 0; i < M; ++i) jtj[i * M + i] = diag[i];
        ++k;
        phase = D_ITER_TOP;
        break;
      }

      case D_FINISH: {  // lm_core.c:809-841
        if (k >= itmax) stop = 3;
        for (int i = 0; i < M; ++i) jtj[i * M + i] = diag[i];
        info[0] = init_e2;
        info[1] = p_e2;
        info[2] = jte_inf;
        info[3] = dp_l2;
        double t = -DBL_MAX;
        for (int i = 0; i < M; ++i)
          if (t < jtj[i * M + i]) t = jtj[i * M + i];
        info[4] = mu / t;
        info[5] = (double)k;
        info[6] = (double)stop;
        info[7] = (double)nfev;
        info[8] = (double)njap;
        info[9] = (double)nlss;
        if (want_covar) lu_covar<M>(jtj, covar, p_e2, n);
        ret = (stop != 4 && stop != 7) ? k : kLmError;
        clear_req();
        phase = D_DONE;
        return;
      }

      default:
        req.kind = RQ_DONE;
        return;
      }
    }
  }
};

// =================================================================================================
// dlevmar_bc_dif = dlevmar_bc_der driven by a finite-difference Jacobian (lmbc_core.c:369-1129):
// projected LM step; if it does not reduce the error enough, Schnabel's backtracking line search
// along it; if that is not a descent direction or fails, a projected-gradient search.
// =================================================================================================
template <int M>
struct BcMachine {
  enum Phase : int {
    B_INIT_EVAL = 1, B_ITER_TOP, B_AFTER_JAC, B_SOLVE, B_AFTER_LM_EVAL, B_AFTER_LM_NORM, B_LM_JUDGE,
    B_LS_ISSUE, B_LS_EVAL, B_PG_BEGIN, B_PG_ISSUE, B_PG_EVAL, B_PG_NORM, B_PG_JUDGE, B_COMMIT,
    B_END_ITER, B_FINISH, B_DONE
  };
  FitOptions o;
  int itmax, n, want_covar;
  int has_lb, has_ub, has_dscl;
  double lb[M], ub[M], dscl[M];
  int phase, k, stop, nu, nfev, njev, nlss, gprev, infeasible_mask, bad_input;
  double p[M], mu, p_e2, init_e2, jte_inf, p_l2, dp_l2, pdp_e2, keep_max;
  double jtj[M * M], jte[M], diag[M], dp[M], pdp[M], p_start[M];
  double t, t0, gdp;
  // line-search locals (lmbc_core.c:218-225)
  double ls_f0, ls_lambda, ls_plmbda, ls_pfpls, ls_tlmbda, ls_rmnlmb, ls_slp;
  int ls_first, ls_left;
  double info[kInfoSz], covar[M * M];
  int ret;
  Request<M> req;

  LM_HD static double median3(double a, double b, double c) {  // lmbc_core.c:59-61
    return (a >= b) ? ((c >= a) ? a : ((c <= b) ? b : c)) : ((c >= b) ? b : ((c <= a) ? a : c));
  }
  LM_HD void project(double *v) const {  // lmbc_core.c:68-88
    if (!has_lb && !has_ub) return;
    for (int i = M; i-- > 0;) {
      if (has_lb && has_ub)
        v[i] = median3(lb[i], v[i], ub[i]);
      else if (has_ub) {
        if (v[i] > ub[i]) v[i] = ub[i];
      } else {
        if (v[i] < lb[i]) v[i] = lb[i];
      }
    }
  }
  LM_HD void clear_req() {
    req.kind = RQ_DONE;
    req.central = 0;
    req.sel_hx = req.sel_j = 0;
    req.dp_l2 = 0.0;
    req.scal = 1.0;
    for (int i = 0; i < M; ++i) req.p[i] = req.d[i] = req.q[i] = req.dp[i] = 0.0;
  }
  // ask for ||x - f(v)||^2 where v lives in the (possibly scaled) search space
  LM_HD void request_eval(const double *v, int kind = RQ_EVAL) {
    clear_req();
    req.kind = kind;
    for (int i = 0; i < M; ++i) req.p[i] = has_dscl ? v[i] * dscl[i] : v[i];
    ++nfev;
  }

  LM_HD void start(const double *p0, int n_, const double *lb_, const double *ub_, const double *dscl_,
                   int itmax_, const double *opts, int want_covar_) {
    o = make_options(opts);
    if (opts) {  // bc_dif reads delta as |opts[4]| and the sign as the FD flavour: lmbc_core.c:1105,1115
      o.forward = (opts[4] >= 0.0);
      o.delta = lm_abs(opts[4]);
    }
    itmax = itmax_;
    n = n_;
    want_covar = want_covar_;
    has_lb = lb_ != nullptr;
    has_ub = ub_ != nullptr;
    has_dscl = dscl_ != nullptr;
    k = 0;
    stop = 0;
    nu = 2;
    nfev = njev = nlss = 0;
    gprev = 0;
    infeasible_mask = 0;
    bad_input = 0;
    mu = jte_inf = p_l2 = t = t0 = gdp = 0.0;
    p_e2 = init_e2 = pdp_e2 = keep_max = 0.0;
    dp_l2 = DBL_MAX;
    ls_f0 = ls_lambda = ls_plmbda = ls_pfpls = ls_tlmbda = ls_rmnlmb = ls_slp = 0.0;
    ls_first = 1;
    ls_left = 0;
    ret = kLmError;
    for (int i = 0; i < M; ++i) {
      p[i] = p0[i];
      lb[i] = has_lb ? lb_[i] : -DBL_MAX;
      ub[i] = has_ub ? ub_[i] : DBL_MAX;
      dscl[i] = has_dscl ? dscl_[i] : 1.0;
      jte[i] = diag[i] = dp[i] = pdp[i] = 0.0;
    }
    for (int i = 0; i < M * M; ++i) jtj[i] = covar[i] = 0.0;
    for (int i = 0; i < kInfoSz; ++i) info[i] = 0.0;
    clear_req();
    phase = B_DONE;
    if (n < M) {  // lmbc_core.c:440-443
      bad_input = 1;
      return;
    }
    if (has_lb && has_ub)  // lmbc_core.c:451-454 (box_check, misc_core.c:661-671)
      for (int i = 0; i < M; ++i)
        if (lb[i] > ub[i]) {
          bad_input = 2;
          return;
        }
    if (has_dscl)  // lmbc_core.c:456-461
      for (int i = M; i-- > 0;)
        if (dscl[i] <= 0.0) {
          bad_input = 3;
          return;
        }
    for (int i = 0; i < M; ++i) p_start[i] = p[i];
    project(p);  // lmbc_core.c:514-520; the stderr warning is printed by the host shim from the mask
    for (int i = 0; i < M; ++i)
      if (p_start[i] != p[i]) infeasible_mask |= (1 << i);
    clear_req();
    req.kind = RQ_EVAL;  // the first evaluation is at the unscaled projected start, lmbc_core.c:523
    for (int i = 0; i < M; ++i) req.p[i] = p[i];
    phase = B_INIT_EVAL;
  }

  LM_HD void accept_trial() {  // p <- pdp, ||e||^2 <- trial value
    for (int i = 0; i < M; ++i) p[i] = pdp[i];
    p_e2 = pdp_e2;
  }

  LM_HD void step(const double *s, double maxabs) {
    constexpr double alpha = 1e-4, beta = 0.9, gamma = 0.99995, rho = 1e-8, tming = 1e-18, tini = 1.0;
    for (;;) {
      switch (phase) {
      case B_INIT_EVAL:  // lmbc_core.c:523-540
        nfev = 1;
        p_e2 = s[0];
        init_e2 = p_e2;
        if (!lm_finite(p_e2)) stop = 7;
        if (has_dscl)
          for (int i = M; i-- > 0;) {
            p[i] /= dscl[i];
            if (has_ub && ub[i] != DBL_MAX) ub[i] = ub[i] / dscl[i];
            if (has_lb && lb[i] != -DBL_MAX) lb[i] = lb[i] / dscl[i];
          }
        phase = B_ITER_TOP;
        break;

      case B_ITER_TOP: {
        if (!(k < itmax && !stop)) {
          phase = B_FINISH;
          break;
        }
        if (p_e2 <= o.eps3) {
          stop = 6;
          phase = B_FINISH;
          break;
        }
        clear_req();  // FD Jacobian at the unscaled point, lmbc_core.c:555-561 + :1043-1054
        req.kind = RQ_JAC;
        req.central = !o.forward;
        for (int i = 0; i < M; ++i) req.p[i] = has_dscl ? p[i] * dscl[i] : p[i];
        fd_steps<M>(req.p, o.delta, req.d);
        ++njev;
        phase = B_AFTER_JAC;
        return;
      }

      case B_AFTER_JAC: {
        unpack_lower<M>(s, jtj);
        for (int i = 0; i < M; ++i) jte[i] = s[SumLayout<M>::NL + i];
        if (has_dscl) {  // J <- J*D (lmbc_core.c:562-569) folded into the reduced products
          for (int i = 0; i < M; ++i) {
            jte[i] *= dscl[i];
            for (int j = 0; j < M; ++j) jtj[i * M + j] *= dscl[i] * dscl[j];
          }
        }
        int nactive = 0, satisfied = 0;  // lmbc_core.c:639-646
        p_l2 = jte_inf = 0.0;
        for (int i = 0; i < M; ++i) {
          if (has_ub && p[i] == ub[i]) {
            ++nactive;
            if (jte[i] > 0.0) ++satisfied;
          } else if (has_lb && p[i] == lb[i]) {
            ++nactive;
            if (jte[i] < 0.0) ++satisfied;
          } else {
            const double a = lm_abs(jte[i]);
            if (jte_inf < a) jte_inf = a;
          }
          diag[i] = jtj[i * M + i];
          p_l2 += p[i] * p[i];
        }
        if (satisfied == nactive && (jte_inf <= o.eps1)) {
          dp_l2 = 0.0;
          stop = 1;
          phase = B_FINISH;
          break;
        }
        if (k == 0) {  // lmbc_core.c:666-674
          if (!has_lb && !has_ub) {
            double m0 = -DBL_MAX;
            for (int i = 0; i < M; ++i)
              if (diag[i] > m0) m0 = diag[i];
            mu = o.tau * m0;
          } else
            mu = 0.5 * o.tau * p_e2;  // Kanzow's starting damping
        }
        phase = B_SOLVE;
        break;
      }

      case B_SOLVE: {  // lmbc_core.c:677-734
        for (int i = 0; i < M; ++i) jtj[i * M + i] += mu;
        const int solved = lu_solve<M>(jtj, jte, dp);
        ++nlss;
        if (!solved) {  // :788-804
          mu *= nu;
          const int nu2 = (int)((unsigned)nu << 1);
          if (nu2 <= nu) {
            stop = 5;
            phase = B_END_ITER;
            break;
          }
          nu = nu2;
          for (int i = 0; i < M; ++i) jtj[i * M + i] = diag[i];
          break;  // solve again
        }
        for (int i = 0; i < M; ++i) pdp[i] = p[i] + dp[i];
        project(pdp);
        dp_l2 = 0.0;
        for (int i = 0; i < M; ++i) {
          const double d = pdp[i] - p[i];
          dp[i] = d;
          dp_l2 += d * d;
        }
        if (dp_l2 <= o.eps2sq * p_l2) {
          stop = 2;
          phase = B_END_ITER;
          break;
        }
        if (dp_l2 >= (p_l2 + o.eps2) / (kEpsilon * kEpsilon)) {
          stop = 4;
          phase = B_END_ITER;
          break;
        }
        request_eval(pdp);
        phase = B_AFTER_LM_EVAL;
        return;
      }

      case B_AFTER_LM_EVAL:  // overflow guard, lmbc_core.c:748-751
        pdp_e2 = s[0];
        if (!lm_finite(pdp_e2)) {
          if (!lm_finite(maxabs)) {
            stop = 7;
            phase = B_END_ITER;
            break;
          }
          keep_max = maxabs;
          request_eval(pdp, RQ_SCALED);
          --nfev;  // not a user-visible function evaluation
          req.scal = maxabs;
          phase = B_AFTER_LM_NORM;
          return;
        }
        phase = B_LM_JUDGE;
        break;

      case B_AFTER_LM_NORM:
        if (!lm_finite(keep_max * sqrt(s[0]))) {
          stop = 7;
          phase = B_END_ITER;
          break;
        }
        phase = B_LM_JUDGE;
        break;

      case B_LM_JUDGE: {
        if (pdp_e2 <= gamma * p_e2) {  // LM step taken, lmbc_core.c:753-785
          double dL = 0.0;
          for (int i = 0; i < M; ++i) dL += dp[i] * (mu * dp[i] + jte[i]);
          if (dL > 0.0) {
            const double dF = p_e2 - pdp_e2;
            double q = (2.0 * dF / dL - 1.0);
            q = 1.0 - q * q * q;
            mu = mu * ((q >= kOneThird) ? q : kOneThird);
          } else {
            const double q = 0.1 * pdp_e2;
            mu = (mu >= q) ? q : mu;
          }
          nu = 2;
          accept_trial();
          gprev = 0;
          phase = B_END_ITER;
          break;
        }
        gdp = 0.0;  // lmbc_core.c:811-816
        for (int i = 0; i < M; ++i) {
          jte[i] = -jte[i];
          gdp += jte[i] * dp[i];
        }
        if (!(gdp <= -rho * pow(dp_l2, kLsPow / 2.0))) {
          phase = B_PG_BEGIN;
          break;
        }
        // ---- line-search prologue, lmbc_core.c:227-249 (x = p, f = p_e2, g = jte, step = dp)
        const double steptl = 1e3 * sqrt(DBL_EPSILON);
        double pn = sqrt(p_l2);
        const double stepmx = 1e3 * ((pn >= 1.0) ? pn : 1.0);
        ls_f0 = p_e2 * 0.5;
        double acc = 0.0;
        for (int i = M; i-- > 0;) acc += dp[i] * dp[i];
        double sln = sqrt(acc);
        if (sln > stepmx) {
          const double scl = stepmx / sln;
          for (int i = M; i-- > 0;) dp[i] *= scl;
          sln = stepmx;
        }
        double rln = 0.0;
        ls_slp = 0.0;
        for (int i = M; i-- > 0;) {
          ls_slp += jte[i] * dp[i];
          const double den = (lm_abs(p[i]) >= 1.0) ? lm_abs(p[i]) : 1.0;
          const double rel = lm_abs(dp[i]) / den;
          if (rln < rel) rln = rel;
        }
        ls_rmnlmb = steptl / rln;
        ls_lambda = 1.0;
        ls_first = 1;
        ls_plmbda = ls_pfpls = ls_tlmbda = 0.0;
        ls_left = kLsItMax;
        phase = B_LS_ISSUE;
        break;
      }

      case B_LS_ISSUE: {  // lmbc_core.c:253-266
        if (ls_left-- <= 0) {  // iteration limit: failure -> projected gradient
          phase = B_PG_BEGIN;
          break;
        }
        for (int i = M; i-- > 0;) pdp[i] = p[i] + ls_lambda * dp[i];
        project(pdp);
        clear_req();
        req.kind = RQ_EVAL;
        if (!has_dscl) {
          for (int i = 0; i < M; ++i) req.p[i] = pdp[i];
        } else {  // the reference multiplies and divides xpls in place, lmbc_core.c:263-265
          for (int i = M; i-- > 0;) {
            pdp[i] *= dscl[i];
            req.p[i] = pdp[i];
            pdp[i] /= dscl[i];
          }
        }
        ++nfev;
        phase = B_LS_EVAL;
        return;
      }

      case B_LS_EVAL: {  // lmbc_core.c:269-332
        const double fpls = 0.5 * s[0];
        pdp_e2 = s[0];
        if (fpls <= ls_f0 + ls_slp * alpha * ls_lambda) {  // satisfactory point
          if (!lm_finite(pdp_e2)) {  // lmbc_core.c:828
            phase = B_PG_BEGIN;
            break;
          }
          gprev = 0;
          phase = B_COMMIT;
          break;
        }
        if (ls_lambda < ls_rmnlmb) {
          phase = B_PG_BEGIN;
          break;
        }
        if (!lm_finite(fpls)) {
          ls_lambda *= 0.1;
          ls_first = 1;
        } else {
          if (ls_first) {
            ls_tlmbda = -ls_lambda * ls_slp / ((fpls - ls_f0 - ls_slp) * 2.0);
            ls_first = 0;
          } else {
            const double t1 = fpls - ls_f0 - ls_lambda * ls_slp;
            const double t2 = ls_pfpls - ls_f0 - ls_plmbda * ls_slp;
            const double t3 = 1.0 / (ls_lambda - ls_plmbda);
            const double a3 = 3.0 * t3 * (t1 / (ls_lambda * ls_lambda) - t2 / (ls_plmbda * ls_plmbda));
            const double b = t3 * (t2 * ls_lambda / (ls_plmbda * ls_plmbda) - t1 * ls_plmbda / (ls_lambda * ls_lambda));
            const double disc = b * b - a3 * ls_slp;
            if (disc > b * b)
              ls_tlmbda = (-b + ((a3 < 0) ? -sqrt(disc) : sqrt(disc))) / a3;
            else
              ls_tlmbda = (-b + ((a3 < 0) ? sqrt(disc) : -sqrt(disc))) / a3;
            if (ls_tlmbda > ls_lambda * 0.5) ls_tlmbda = ls_lambda * 0.5;
          }
          ls_plmbda = ls_lambda;
          ls_pfpls = fpls;
          if (ls_tlmbda < ls_lambda * 0.1)
            ls_lambda *= 0.1;
          else
            ls_lambda = ls_tlmbda;
        }
        phase = B_LS_ISSUE;
        break;
      }

      case B_PG_BEGIN: {  // lmbc_core.c:877-885 (jte already holds -J^T e)
        double g2 = 0.0;
        for (int i = 0; i < M; ++i) g2 += jte[i] * jte[i];
        g2 = sqrt(g2);
        g2 = 100.0 / (1.0 + g2);
        t0 = (g2 <= tini) ? g2 : tini;
        t = gprev ? t : t0;
        phase = B_PG_ISSUE;
        break;
      }

      case B_PG_ISSUE: {  // loop head of lmbc_core.c:885
        if (!(t > tming)) {  // search failed, :937-939
          gprev = 0;
          phase = B_END_ITER;
          break;
        }
        for (int i = 0; i < M; ++i) pdp[i] = p[i] - t * jte[i];
        project(pdp);
        dp_l2 = 0.0;
        for (int i = 0; i < M; ++i) {
          const double d = pdp[i] - p[i];
          dp[i] = d;
          dp_l2 += d * d;
        }
        request_eval(pdp);
        phase = B_PG_EVAL;
        return;
      }

      case B_PG_EVAL:
        pdp_e2 = s[0];
        if (!lm_finite(pdp_e2)) {  // lmbc_core.c:915-918
          if (!lm_finite(maxabs)) {
            stop = 7;
            phase = B_FINISH;
            break;
          }
          keep_max = maxabs;
          request_eval(pdp, RQ_SCALED);
          --nfev;
          req.scal = maxabs;
          phase = B_PG_NORM;
          return;
        }
        phase = B_PG_JUDGE;
        break;

      case B_PG_NORM:
        if (!lm_finite(keep_max * sqrt(s[0]))) {
          stop = 7;
          phase = B_FINISH;  // "goto breaknested": k is not advanced
          break;
        }
        phase = B_PG_JUDGE;
        break;

      case B_PG_JUDGE: {  // lmbc_core.c:923-935
        gdp = 0.0;
        for (int i = 0; i < M; ++i) gdp += jte[i] * dp[i];
        if (gprev && pdp_e2 <= p_e2 + 2.0 * 0.99999 * gdp) {  // remembered t was too small
          t = t0;
          gprev = 0;
          t *= beta;  // the reference's `continue` still runs the loop increment
          phase = B_PG_ISSUE;
          break;
        }
        if (pdp_e2 <= p_e2 + 2.0 * alpha * gdp) {
          gprev = 1;
          phase = B_COMMIT;
          break;
        }
        t *= beta;
        phase = B_PG_ISSUE;
        break;
      }

      case B_COMMIT: {  // lmbc_core.c:950-967
        dp_l2 = 0.0;
        for (int i = 0; i < M; ++i) {
          const double d = pdp[i] - p[i];
          dp_l2 += d * d;
        }
        if (dp_l2 <= o.eps2sq * p_l2) {
          stop = 2;
          phase = B_END_ITER;
          break;
        }
        accept_trial();
        phase = B_END_ITER;
        break;
      }

      case B_END_ITER:
        ++k;
        phase = B_ITER_TOP;
        break;

      case B_FINISH: {  // lmbc_core.c:973-1021, :1119-1124
        if (k >= itmax) stop = 3;
        for (int i = 0; i < M; ++i) jtj[i * M + i] = diag[i];
        info[0] = init_e2;
        info[1] = p_e2;
        info[2] = jte_inf;
        info[3] = dp_l2;
        double m0 = -DBL_MAX;
        for (int i = 0; i < M; ++i)
          if (m0 < jtj[i * M + i]) m0 = jtj[i * M + i];
        info[4] = mu / m0;
        info[5] = (double)k;
        info[6] = (double)stop;
        info[7] = (double)nfev + (double)njev * (o.forward ? (M + 1) : (2 * M));
        info[8] = (double)njev;
        info[9] = (double)nlss;
        if (want_covar) {
          lu_covar<M>(jtj, covar, p_e2, n);
          if (has_dscl)
            for (int i = M; i-- > 0;)
              for (int j = M; j-- > 0;) covar[i * M + j] *= (dscl[i] * dscl[j]);
        }
        if (has_dscl)
          for (int i = 0; i < M; ++i) p[i] *= dscl[i];
        ret = (stop != 4 && stop != 7) ? k : kLmError;
        clear_req();
        phase = B_DONE;
        return;
      }

      default:
        req.kind = RQ_DONE;
        return;
      }
    }
  }
};

}  // namespace brdf
