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
#ifndef LM_STAMP
#define LM_STAMP(i) do {} while (0)  // diagnostic builds (-DBRDF_STAMPS) time the sections of a step
#endif

// a guarded phase block of BcMachine::run (see there); ONE_LANE: the comparison stays on the scalar unit
#ifndef LM_PHASE_ENTER
#define LM_PHASE_ENTER(X) true  // diagnostic builds of lane_fit.hip time the phases (a wave's serial walk over its lanes' phases)
#endif
#define LM_PHASE(X) if ((ONE_LANE ? lm_uniform(ph) : ph) == X && LM_PHASE_ENTER(X)) do
#define LM_PHASE_END while (0);
// in front of a gated block: a GATED step outside a heavy round stops here (RQ_YIELD)
#define LM_GATE(X)                                    \
  if (GATED && !heavy && ph == X) {                   \
    req.kind = RQ_YIELD;                              \
    { h.phase = ph; return; }                         \
  }

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
  RQ_DIF_TRIAL = 6,  // speculative:  hx[!sel_hx] <- f(q); J[!sel_j] <- Broyden(J[sel_j]); sums = [sum e_new^2,
                     //   JnTJn lower, JnT e_new, JnT e_old]
                     // two-step:     wrk <- f(q); sums = [sum e_new^2]
  RQ_DIF_UPDATE = 7, // two-step only: J <- Broyden(J, wrk, hx); sums = [JTJ lower, JT e] with e = x-wrk if
                     //   aux (step accepted) else x-hx; finally hx <- wrk if aux
  RQ_EVAL_MULTI = 8, // sums[j] = sum (x-f(pk[j]))^2 for j < nk: several candidates of a projected-gradient search
                     //   in ONE sweep (the samples are read once; a pass's fixed cost is paid once)
  RQ_YIELD = 9       // no pass: a GATED step (BcMachine::run) stopped in front of an expensive phase; call step() again
                     //   with heavy = true (lane_fit.hip runs the expensive phases of its 64 machines in common rounds)
};

constexpr int kMaxCand = 8;  // candidates per RQ_EVAL_MULTI

// Real: the arithmetic type of a machine -- double (dlevmar_*, every BRDF kernel) or float (the slevmar_* twins,
// levmar.h:208-310, instantiated from the same source exactly as the reference instantiates its *_core.c files with
// LM_REAL = float, lm.c:43-63).  Every floating literal below is written Real(x), the reference's LM_CNST(x).
template <class Real>
struct LmLimits;
template <>
struct LmLimits<double> {
  static LM_HD double eps() { return DBL_EPSILON; }
  static LM_HD double max() { return DBL_MAX; }
};
template <>
struct LmLimits<float> {
  static LM_HD float eps() { return FLT_EPSILON; }
  static LM_HD float max() { return FLT_MAX; }
};

template <int M, class Real = double>
struct Request {
  int kind;
  int central;  // RQ_JAC / RQ_DIF_JAC: 0 forward, 1 central differences
  int sel_hx;   // which of the two hx buffers holds f(current p)     (dif only)
  int sel_j;    // which of the two Jacobian buffers is current       (dif only)
  int aux;      // RQ_DIF_UPDATE: 1 if the trial step was accepted
  Real p[M];  // evaluation point / Jacobian base point
  Real d[M];  // finite-difference steps                            (misc_core.c:155-158)
  Real q[M];  // trial point p+Dp                                   (RQ_DIF_TRIAL)
  Real dp[M]; // Dp                                                 (RQ_DIF_TRIAL)
  Real dp_l2; // ||Dp||^2
  Real scal;  // RQ_SCALED divisor
  int nk;       // RQ_EVAL_MULTI: number of candidate points
  int pad_;
  Real pk[kMaxCand][M];
};

template <int M>
struct SumLayout {
  static constexpr int NL = M * (M + 1) / 2;
  static constexpr int JAC = NL + M + 1;          // RQ_JAC
  static constexpr int DIF_JAC = NL + M;          // RQ_DIF_JAC
  static constexpr int DIF_TRIAL = 1 + NL + 2 * M;  // RQ_DIF_TRIAL
  static constexpr int MAX = DIF_TRIAL > JAC ? DIF_TRIAL : JAC;
};

// On the device the LM step runs on ONE lane of a wave.  Values loaded from the LDS-resident machine are not
// provably wave-uniform to the compiler, and a `switch` on such a value is lowered to exec-masked, structurised
// control flow that walks every case with a compare + saveexec (~2000 cycles per step for the phase dispatch
// alone, measured).  lm_uniform() tells the compiler the truth -- the value is the same in every active lane --
// so the dispatch becomes scalar branches.  ONLY valid where a single lane steps (step<true>): the
// four-fits-per-wave kernel steps four machines in four lanes at once and must not use it.
LM_HD int lm_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// lane `lane`'s value of v (lane: a compile-time constant after unrolling)
__device__ __forceinline__ double lm_read_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lm_read_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
#endif

template <class Real>
LM_HD Real lm_abs(Real v) { return (v >= Real(0.0)) ? v : -v; }
template <class Real>
LM_HD bool lm_finite(Real v) { return (v - v) == Real(0.0); }  // false for NaN and +-Inf

// Crout LU with implicit row scaling + partial pivoting and the DBL_EPSILON zero-pivot rule, then
// forward/back substitution: Axb_core.c:1197-1270.  A, B untouched; returns 0 if a row of A is zero.
// (M <= 8: fully unrolled; the wide instantiations of the host-callback path, M = 9..16, run on the host and keep their loops.)
// Every array index below is a compile-time constant after unrolling (row exchanges and the
// permuted right-hand-side picks are written as selects over all candidate rows), so on the GPU the
// factorisation lives entirely in registers -- no scratch memory, no indirect register access.
template <int M, class Real>
LM_HD int lu_solve(const Real *A, const Real *B, Real *x) {
  constexpr int kUnrollLu = M <= 8 ? 64 : 1;  // (unroll count of every loop below: all of it, or nothing)
  Real a[M * M], scale[M];
  int perm[M];
#pragma unroll kUnrollLu
  for (int i = 0; i < M * M; ++i) a[i] = A[i];
#pragma unroll kUnrollLu
  for (int i = 0; i < M; ++i) x[i] = B[i];
#pragma unroll kUnrollLu
  for (int i = 0; i < M; ++i) {
    Real big = Real(0.0);
#pragma unroll kUnrollLu
    for (int j = 0; j < M; ++j) {
      const Real t = lm_abs(a[i * M + j]);
      if (t > big) big = t;
    }
    if (big == Real(0.0)) return 0;
    scale[i] = Real(1.0) / big;
  }
#pragma unroll kUnrollLu
  for (int j = 0; j < M; ++j) {
    int pivot = j;
    Real big = Real(0.0);
#pragma unroll kUnrollLu
    for (int i = 0; i < j; ++i) {
      Real s = a[i * M + j];
#pragma unroll kUnrollLu
      for (int k = 0; k < i; ++k) s -= a[i * M + k] * a[k * M + j];
      a[i * M + j] = s;
    }
#pragma unroll kUnrollLu
    for (int i = j; i < M; ++i) {
      Real s = a[i * M + j];
#pragma unroll kUnrollLu
      for (int k = 0; k < j; ++k) s -= a[i * M + k] * a[k * M + j];
      a[i * M + j] = s;
      const Real t = scale[i] * lm_abs(s);
      if (t >= big) {
        big = t;
        pivot = i;
      }
    }
    // exchange rows j and pivot (pivot >= j); scale[pivot] <- scale[j]
    // (measured: a wave-level "does any lane exchange this row" guard in front of these selects changes nothing -- 2,465 against
    // 2,463 cycles per solve in the resident dlevmar_dif kernel: the solve is bound by its chain of eight divisions)
#pragma unroll kUnrollLu
    for (int r = j + 1; r < M; ++r) {
      if (pivot == r) {
#pragma unroll kUnrollLu
        for (int k = 0; k < M; ++k) {
          const Real t = a[r * M + k];
          a[r * M + k] = a[j * M + k];
          a[j * M + k] = t;
        }
        scale[r] = scale[j];
      }
    }
    perm[j] = pivot;
    if (a[j * M + j] == Real(0.0)) a[j * M + j] = LmLimits<Real>::eps();
    if (j != M - 1) {
      const Real t = Real(1.0) / a[j * M + j];
#pragma unroll kUnrollLu
      for (int i = j + 1; i < M; ++i) a[i * M + j] *= t;
    }
  }
  int first = 0;
#pragma unroll kUnrollLu
  for (int i = 0; i < M; ++i) {
    // s = x[perm[i]]; x[perm[i]] = x[i];   with perm[i] >= i
    Real s = x[i];
#pragma unroll kUnrollLu
    for (int r = i + 1; r < M; ++r) {
      if (perm[i] == r) {
        s = x[r];
        x[r] = x[i];
      }
    }
    if (first != 0) {
#pragma unroll kUnrollLu
      for (int jj = 0; jj < i; ++jj)
        if (jj >= first - 1) s -= a[i * M + jj] * x[jj];
    } else if (s != Real(0.0)) {
      first = i + 1;
    }
    x[i] = s;
  }
#pragma unroll kUnrollLu
  for (int i = M - 1; i >= 0; --i) {
    Real s = x[i];
#pragma unroll kUnrollLu
    for (int j = i + 1; j < M; ++j) s -= a[i * M + j] * x[j];
    x[i] = s / a[i * M + i];
  }
  return 1;
}

// covar = sumsq/(n-M) * inverse(JtJ) by the same LU, column by column: misc_core.c:426-591.
template <int M, class Real>
LM_HD int lu_covar(const Real *JtJ, Real *C, Real sumsq, int n) {
  for (int l = 0; l < M; ++l) {
    Real rhs[M], col[M];
    for (int i = 0; i < M; ++i) rhs[i] = (i == l) ? Real(1.0) : Real(0.0);
    if (!lu_solve<M>(JtJ, rhs, col)) return 0;
    for (int i = 0; i < M; ++i) C[i * M + l] = col[i];
  }
  const Real fact = sumsq / (Real)(n - M);
  for (int i = 0; i < M * M; ++i) C[i] *= fact;
  return M;
}

template <int M, class Real>
LM_HD void fd_steps(const Real *p, Real delta, Real *d) {  // misc_core.c:155-158
  for (int j = 0; j < M; ++j) {
    Real s = Real(1E-04) * p[j];
    s = lm_abs(s);
    if (s < delta) s = delta;
    d[j] = s;
  }
}

template <int M, class Real>
LM_HD void unpack_lower(const Real *s, Real *jtj) {
  int c = 0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j <= i; ++j, ++c) {
      jtj[i * M + j] = s[c];
      jtj[j * M + i] = s[c];
    }
}

template <class Real = double>
struct FitOptionsT {
  Real tau, eps1, eps2, eps2sq, eps3, delta;
  int forward;
};

// opts == NULL selects the defaults of levmar.h:98-100 (lm_core.c:507-526)
template <class Real>
LM_HD FitOptionsT<Real> make_options(const Real *opts) {
  FitOptionsT<Real> o;
  if (opts) {
    o.tau = opts[0];
    o.eps1 = opts[1];
    o.eps2 = opts[2];
    o.eps2sq = opts[2] * opts[2];
    o.eps3 = opts[3];
    o.delta = opts[4];
  } else {
    o.tau = Real(kInitMu);
    o.eps1 = Real(kStopThresh);
    o.eps2 = Real(kStopThresh);
    o.eps2sq = Real(kStopThresh) * Real(kStopThresh);
    o.eps3 = Real(kStopThresh);
    o.delta = Real(kDiffDelta);
  }
  o.forward = 1;
  if (o.delta < Real(0.0)) {
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
template <int M, class Real = double>
struct DifMachine {
  enum Phase : int { D_INIT_EVAL = 1, D_ITER_TOP, D_AFTER_JAC, D_GRADIENT, D_SOLVE, D_AFTER_TRIAL, D_AFTER_UPDATE, D_DECIDE, D_REJECT, D_AFTER_MULTI, D_FINISH, D_DONE };
  // Cold: configuration and results, touched at start/finish only.  Hot: everything an LM step reads or
  // writes.  (Measured on gfx950: running the step on a register copy of Hot makes hipcc spill to scratch
  // and is slower than stepping in place in LDS, so step() works in place.)
  struct Cold {
    FitOptionsT<Real> o;
    int itmax, n, want_covar, refresh;
    int speculative;  // 1: the trial pass also produces the products of the Broyden-updated Jacobian (one pass per LM
                      //    iteration).  The launch chain double-buffers J in HBM and commits by flipping sel_j; the
                      //    resident and batched kernels keep f(p+Dp) per sample and apply the adopted update at the top
                      //    of the next pass.  Every BRDF kernel uses this protocol.
                      // 0: trial pass, decision, then an update pass (RQ_DIF_UPDATE), the plain restatement of
                      //    lm_core.c:742-790: the host-callback path (generic_fit.hip), where hx and J are whole
                      //    vectors in HBM, and the host harness (tests/cpp/host_machine.cpp; bit-exact, like 1)
    int multi;        // > 1: a chain of rejections is evaluated up to `multi` trial points at a time.  While no step has been
                      //   taken since the last fresh Jacobian (updp == 0) a rejected trial changes nothing but the damping
                      //   (lm_core.c:797-806: mu *= nu, nu *= 2; no Broyden update, :757), so the NEXT trial points -- the
                      //   solutions of (J^T J + mu_j I) dp = J^T e for mu_j = mu nu, mu nu 2nu, ... -- are known before the current
                      //   one has been judged.  Near the end of a fit (every step rejected until ||Dp|| falls under eps2) that is
                      //   most of the passes.  RQ_EVAL_MULTI evaluates them in one sweep; they are judged in the reference's order,
                      //   only the judged ones are counted, and a candidate that reduces the error is evaluated again by the plain
                      //   trial pass (which forms the Broyden sums): p, info[] and the trajectory are those of one trial at a time.
    Real info[kInfoSz], covar[M * M];
    int ret;
  };
  // Everything an LM step reads or writes, except the request it leaves: the counters and flags, and the reals.  run() takes the
  // two halves separately, so that a kernel that steps ONE machine with a whole wave can keep the counters in scalar registers for
  // the whole fit (resident_fit_impl.h: every `if (h.k < c.itmax ...)` on an LDS-resident machine is a dependent ds_read ->
  // s_waitcnt -> compare -> branch) while the reals stay where the machine lives.
  struct CoreInts {
    int phase, k, stop, nu, nfev, njap, nlss, updjac, updp, newjac;
    int sel_hx, sel_j, accepted;
    int chain, single, mcnt;  // multi: rejections without an update in a row; next solve issues a plain trial; candidates out
  };
  struct CoreReals {
    Real p[M], mu, p_e2, jte_inf, p_l2, dp_l2, pdp_e2;
    Real jtj[M * M], jte[M], dp[M];
  };
  struct Core : CoreInts, CoreReals {};
  struct Cool {  // the less busy half of the state (see Hot)
    Real init_e2, diag[M], pdp[M];
    Real spec_jtj[M * M], spec_jte[M];  // normal equations of the Broyden-updated J, adopted lazily
    Real ml2[kMaxCand];                  // multi: ||Dp||^2 of the candidates out (info[3] is the last one judged)
  };
  // Core + Cool + the request.  run() takes them separately, so that a kernel can step on a REGISTER copy of Core
  // while the rest stays in LDS (resident_fit.hip: the sweeping waves read the request there; with Cool in registers as
  // well, hipcc spilled ~100 VGPRs of the step to scratch, which cost more than the LDS round trips it saved)
  struct Hot : Core {
    Cool cool;
    Request<M, Real> req;
  };
  // where ONE machine is stepped by a whole wave (every lane the same values): moves the counters and flags of a
  // register copy into scalar registers, so that the step's integer logic and branches run on the scalar unit
  static LM_HD void uniform_ints(CoreInts &h) {
    h.phase = lm_uniform(h.phase);
    h.k = lm_uniform(h.k);
    h.stop = lm_uniform(h.stop);
    h.nu = lm_uniform(h.nu);
    h.nfev = lm_uniform(h.nfev);
    h.njap = lm_uniform(h.njap);
    h.nlss = lm_uniform(h.nlss);
    h.updjac = lm_uniform(h.updjac);
    h.updp = lm_uniform(h.updp);
    h.newjac = lm_uniform(h.newjac);
    h.sel_hx = lm_uniform(h.sel_hx);
    h.sel_j = lm_uniform(h.sel_j);
    h.accepted = lm_uniform(h.accepted);
    h.chain = lm_uniform(h.chain);
    h.single = lm_uniform(h.single);
    h.mcnt = lm_uniform(h.mcnt);
  }
  Cold c;
  Hot h;

  LM_HD void start(const Real *p0, int n_, int itmax_, const Real *opts, int want_covar_, int speculative_ = 1, int multi_ = 1) {
    Request<M, Real> &req = h.req;
    Cool &cool = h.cool;
    c.o = make_options(opts);
    c.speculative = speculative_;
    c.multi = (multi_ < 1 || !speculative_) ? 1 : ((multi_ > kMaxCand) ? kMaxCand : multi_);
    h.chain = h.single = h.mcnt = 0;
    for (int j = 0; j < kMaxCand; ++j) cool.ml2[j] = Real(0.0);
    h.accepted = 0;
    c.itmax = itmax_;
    c.n = n_;
    c.want_covar = want_covar_;
    h.k = 0;
    h.stop = 0;
    h.nfev = h.njap = h.nlss = 0;
    h.updjac = 0;
    h.updp = 1;
    h.newjac = 0;
    c.refresh = (M >= 10) ? M : 10;  // "K", lm_core.c:495
    h.sel_hx = h.sel_j = 0;
    h.mu = h.jte_inf = h.p_l2 = Real(0.0);
    h.p_e2 = cool.init_e2 = h.pdp_e2 = Real(0.0);
    h.dp_l2 = LmLimits<Real>::max();
    c.ret = kLmError;
    for (int i = 0; i < M; ++i) {
      h.p[i] = p0[i];
      h.jte[i] = cool.diag[i] = h.dp[i] = cool.pdp[i] = Real(0.0);
      cool.spec_jte[i] = Real(0.0);
    }
    for (int i = 0; i < M * M; ++i) h.jtj[i] = cool.spec_jtj[i] = c.covar[i] = Real(0.0);
    for (int i = 0; i < kInfoSz; ++i) c.info[i] = Real(0.0);
    clear_req(h, req);
    if (c.n < M) {  // lm_core.c:502-505
      h.phase = D_DONE;
      req.kind = RQ_DONE;
      return;
    }
    req.kind = RQ_DIF_INIT;
    for (int i = 0; i < M; ++i) req.p[i] = h.p[i];
    h.phase = D_INIT_EVAL;
  }

  static LM_HD void clear_req(const CoreInts &h, Request<M, Real> &req) {
    req.kind = RQ_DONE;
    req.central = 0;
    req.sel_hx = h.sel_hx;
    req.sel_j = h.sel_j;
    req.aux = 0;
    req.dp_l2 = Real(0.0);
    req.scal = Real(1.0);
    req.nk = 0;
    for (int i = 0; i < M; ++i) req.p[i] = req.d[i] = req.q[i] = req.dp[i] = Real(0.0);
  }

  static LM_HD void gradient_stats(CoreReals &h, Cool &cool) {  // lm_core.c:657-662
    h.p_l2 = h.jte_inf = Real(0.0);
    for (int i = 0; i < M; ++i) {
      const Real t = lm_abs(h.jte[i]);
      if (h.jte_inf < t) h.jte_inf = t;
      cool.diag[i] = h.jtj[i * M + i];
      h.p_l2 += h.p[i] * h.p[i];
    }
  }

  // ONE_LANE: the caller guarantees that exactly one lane of the wave executes this step (see lm_uniform)
  // MULTI: compiles the multi-candidate rejection chains in (Cold::multi > 1 turns them on)
  template <bool ONE_LANE = false, bool MULTI = false>
  LM_HD void step(const Real *s, Real maxabs) { run<ONE_LANE, MULTI>(c, h, h.cool, h.req, s, maxabs); }

  template <bool ONE_LANE, bool MULTI = false>
  static LM_HD void run(Cold &c, Core &h, Cool &cool, Request<M, Real> &req, const Real *s, Real maxabs) {
    run<ONE_LANE, MULTI>(c, static_cast<CoreInts &>(h), static_cast<CoreReals &>(h), cool, req, s, maxabs);
  }
  template <bool ONE_LANE, bool MULTI = false>
  static LM_HD void run(Cold &c, CoreInts &hi, CoreReals &h, Cool &cool, Request<M, Real> &req, const Real *s, Real /*maxabs*/) {
    int ph = ONE_LANE ? lm_uniform(hi.phase) : hi.phase;  // scalar register: phase transitions become scalar jumps
    LM_STAMP(0);
    for (;;) {
      if (ONE_LANE) ph = lm_uniform(ph);  // re-assert uniformity: assignments under (formally) divergent branches lose it
      // (an ordered chain of guarded blocks, not a switch: see BcMachine::run)
      if (ph <= 0 || ph >= D_DONE) {
        req.kind = RQ_DONE;
        { hi.phase = ph; return; }
      }
      LM_PHASE(D_INIT_EVAL) {  // lm_core.c:551-564
        hi.nfev = 1;
        h.p_e2 = s[0];
        cool.init_e2 = h.p_e2;
        if (!lm_finite(h.p_e2)) hi.stop = 7;
        hi.nu = 20;
        ph = D_ITER_TOP;
        break;
      } LM_PHASE_END

      LM_PHASE(D_AFTER_JAC) {
        unpack_lower<M>(s, h.jtj);
        for (int i = 0; i < M; ++i) h.jte[i] = s[SumLayout<M>::NL + i];
        hi.newjac = 0;
        gradient_stats(h, cool);
        ph = D_SOLVE;
        break;
      } LM_PHASE_END

      LM_PHASE(D_AFTER_TRIAL) {  // lm_core.c:742-790
        h.pdp_e2 = s[0];
        if (!lm_finite(h.pdp_e2)) {
          hi.stop = 7;
          ph = D_FINISH;
          break;
        }
        const Real dF = h.p_e2 - h.pdp_e2;
        const bool updated = (hi.updp || dF > 0);
        if (MULTI) hi.chain = updated ? 0 : hi.chain + 1;  // (not updated => dF <= 0 => rejected below)
        Real dL = Real(0.0);
        for (int i = 0; i < M; ++i) dL += h.dp[i] * (h.mu * h.dp[i] + h.jte[i]);
        hi.accepted = (dL > Real(0.0) && dF > Real(0.0)) ? 1 : 0;
        if (hi.accepted) {  // damping update uses dF, dL of this step: do it now, they are not kept
          Real t = (Real(2.0) * dF / dL - Real(1.0));
          t = Real(1.0) - t * t * t;
          h.mu = h.mu * ((t >= Real(kOneThird)) ? t : Real(kOneThird));
        }
        if (updated) {
          ++hi.updjac;
          hi.newjac = 1;
          if (c.speculative) {  // adopt the speculatively updated Jacobian; keep its products, paired with the
                                // residual that stays live -- they replace jtj/jte at the top of the next
                                // iteration, as in the reference
            hi.sel_j ^= 1;
            unpack_lower<M>(s + 1, cool.spec_jtj);
            const Real *g = s + 1 + SumLayout<M>::NL + (hi.accepted ? 0 : M);
            for (int i = 0; i < M; ++i) cool.spec_jte[i] = g[i];
          } else {
            clear_req(hi, req);
            req.kind = RQ_DIF_UPDATE;
            req.aux = hi.accepted;
            for (int i = 0; i < M; ++i) {
              req.p[i] = h.p[i];
              req.q[i] = cool.pdp[i];
              req.dp[i] = h.dp[i];
            }
            req.dp_l2 = h.dp_l2;
            ph = D_AFTER_UPDATE;
            { hi.phase = ph; return; }
          }
        }
        LM_STAMP(1);
        ph = D_DECIDE;
        break;
      } LM_PHASE_END

      LM_PHASE(D_AFTER_UPDATE) {
        unpack_lower<M>(s, cool.spec_jtj);
        for (int i = 0; i < M; ++i) cool.spec_jte[i] = s[SumLayout<M>::NL + i];
        ph = D_DECIDE;
        break;
      } LM_PHASE_END

      if constexpr (MULTI)
      LM_PHASE(D_AFTER_MULTI) {  // judge the candidates of one sweep in the reference's order: each is one iteration of
                                 // lm_core.c:566-807 that ends in the rejection branch without a Broyden update
        const int cnt = hi.mcnt;
        int next = D_ITER_TOP;
        for (int j = 0; j < kMaxCand; ++j) {
          if (j >= cnt || !(hi.k < c.itmax)) break;  // (the iteration count ends the loop at its top, lm_core.c:566)
          const Real e2 = s[j];
          if (lm_finite(e2) && h.p_e2 - e2 > Real(0.0)) {  // dF > 0: Broyden update, maybe an accepted step -- the plain
            hi.single = 1;                                   // trial pass evaluates this candidate again and forms its sums
            break;
          }
          ++hi.nlss;  // the solve that produced this candidate (lm_core.c:691) ...
          ++hi.nfev;  // ... and its evaluation (lm_core.c:738)
          h.pdp_e2 = e2;
          h.dp_l2 = cool.ml2[j];
          if (!lm_finite(e2)) {  // lm_core.c:749
            hi.stop = 7;
            next = D_FINISH;
            break;
          }
          ++hi.chain;
          h.mu *= hi.nu;  // lm_core.c:797-806
          const int nu2 = (int)((unsigned)hi.nu << 1);
          if (nu2 <= hi.nu) {
            hi.stop = 5;
            next = D_FINISH;
            break;
          }
          hi.nu = nu2;
          ++hi.k;
        }
        ph = next;
        break;
      } LM_PHASE_END

      LM_PHASE(D_DECIDE) {
        if (hi.accepted) {
          hi.nu = 2;
          for (int i = 0; i < M; ++i) h.p[i] = cool.pdp[i];
          if (c.speculative) hi.sel_hx ^= 1;  // e, hx <- trial values
          h.p_e2 = h.pdp_e2;
          hi.updp = 1;
          ++hi.k;
          ph = D_ITER_TOP;
          break;
        }
        ph = D_REJECT;
        break;
      } LM_PHASE_END

      LM_PHASE(D_REJECT) {  // lm_core.c:797-806
        h.mu *= hi.nu;
        const int nu2 = (int)((unsigned)hi.nu << 1);
        if (nu2 <= hi.nu) {
          hi.stop = 5;
          ph = D_FINISH;
          break;
        }
        hi.nu = nu2;
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] = cool.diag[i];
        ++hi.k;
        ph = D_ITER_TOP;
        break;
      } LM_PHASE_END

      LM_PHASE(D_ITER_TOP) {
        if (!(hi.k < c.itmax && !hi.stop)) {
          ph = D_FINISH;
          break;
        }
        if (h.p_e2 <= c.o.eps3) {
          hi.stop = 6;
          ph = D_FINISH;
          break;
        }
        if ((hi.updp && hi.nu > 16) || hi.updjac == c.refresh) {  // fresh FD Jacobian, lm_core.c:578-588
          clear_req(hi, req);
          req.kind = RQ_DIF_JAC;
          req.central = !c.o.forward;
          for (int i = 0; i < M; ++i) req.p[i] = h.p[i];
          fd_steps<M>(h.p, c.o.delta, req.d);
          ++hi.njap;
          hi.nfev += c.o.forward ? M : 2 * M;
          hi.nu = 2;
          hi.updjac = 0;
          hi.updp = 0;
          hi.newjac = 1;
          if (MULTI) hi.chain = 0;
          ph = D_AFTER_JAC;
          { hi.phase = ph; return; }
        }
        ph = D_GRADIENT;
        break;
      } LM_PHASE_END

      LM_PHASE(D_GRADIENT) {
        LM_STAMP(5);
        if (hi.newjac) {  // lm_core.c:590-664 with the sums the trial pass produced for the updated J
          hi.newjac = 0;
          for (int i = 0; i < M * M; ++i) h.jtj[i] = cool.spec_jtj[i];
          for (int i = 0; i < M; ++i) h.jte[i] = cool.spec_jte[i];
          gradient_stats(h, cool);
        }
        ph = D_SOLVE;
        break;
      } LM_PHASE_END

      LM_PHASE(D_SOLVE) {
        LM_STAMP(2);
        if (h.jte_inf <= c.o.eps1) {  // lm_core.c:676-680
          h.dp_l2 = Real(0.0);
          hi.stop = 1;
          ph = D_FINISH;
          break;
        }
        if (hi.k == 0) {  // lm_core.c:683-687
          Real t = -LmLimits<Real>::max();
          for (int i = 0; i < M; ++i)
            if (cool.diag[i] > t) t = cool.diag[i];
          h.mu = c.o.tau * t;
        }
        if (MULTI && c.multi > 1 && hi.updp == 0 && hi.chain >= 1 && !hi.single) {
          // no step taken since the fresh Jacobian and the last trial was rejected without an update: if this trial is
          // rejected too, the next one differs by its damping only.  Form the trial points of the next rejections as the
          // loop would (same solves, same tests), stop in front of the first one the loop would not evaluate.
          Real A[M * M], b[M], pc[M];
          for (int i = 0; i < M * M; ++i) A[i] = h.jtj[i];
          for (int i = 0; i < M; ++i) {
            b[i] = h.jte[i];
            pc[i] = h.p[i];
          }
          Real mu = h.mu;
          int nu = hi.nu, kk = hi.k, cnt = 0;
          const Real pl2 = h.p_l2;
#if defined(__HIP_DEVICE_COMPILE__)
          if (ONE_LANE) {
            // On the device the step runs on all 64 lanes of a wave with identical values (MULTI is only compiled into kernels
            // that step that way): the solves are independent given their damping, so lane j % 8 solves candidate j -- one LU's
            // latency (~2,200 cycles) instead of eight -- and readlane collects the trial points.  Same arithmetic per
            // candidate as the loop below, hence the same bits.
            const int myj = (int)(__lane_id() & (kMaxCand - 1));
            Real my_mu = mu;
            int limit = 0;
            for (int j = 0; j < kMaxCand; ++j) {  // the dampings of the chain (wave-uniform)
              if (j >= c.multi || !(kk < c.itmax)) break;
              if (j == myj) my_mu = mu;
              limit = j + 1;
              mu *= nu;
              const int nu2 = (int)((unsigned)nu << 1);
              if (nu2 <= nu) break;
              nu = nu2;
              ++kk;
            }
            Real dpj[M], q[M], l2 = Real(0.0);
            for (int i = 0; i < M; ++i) A[i * M + i] = cool.diag[i] + my_mu;
            bool ok = lu_solve<M>(A, b, dpj) != 0;
            for (int i = 0; i < M; ++i) {
              q[i] = pc[i] + dpj[i];
              l2 += dpj[i] * dpj[i];
            }
            ok = ok && !(l2 <= c.o.eps2sq * pl2 || l2 >= (pl2 + c.o.eps2) / (Real(kEpsilon) * Real(kEpsilon)));
            const unsigned good = (unsigned)__builtin_amdgcn_ballot_w64(ok) & 0xFFu;  // bit j: candidate j would be evaluated
            for (int j = 0; j < kMaxCand; ++j) {
              if (j >= limit || !(good >> j & 1u)) break;
              for (int i = 0; i < M; ++i) req.pk[j][i] = lm_read_lane(q[i], j);
              cool.ml2[j] = lm_read_lane(l2, j);
              ++cnt;
            }
          } else
#endif
          for (int j = 0; j < kMaxCand; ++j) {
            if (j >= c.multi || !(kk < c.itmax)) break;
            Real dpj[M];
            for (int i = 0; i < M; ++i) A[i * M + i] = cool.diag[i] + mu;
            if (!lu_solve<M>(A, b, dpj)) break;
            Real l2 = Real(0.0);
            for (int i = 0; i < M; ++i) {
              req.pk[j][i] = pc[i] + dpj[i];
              l2 += dpj[i] * dpj[i];
            }
            if (l2 <= c.o.eps2sq * pl2 || l2 >= (pl2 + c.o.eps2) / (Real(kEpsilon) * Real(kEpsilon))) break;
            cool.ml2[j] = l2;
            ++cnt;
            mu *= nu;
            const int nu2 = (int)((unsigned)nu << 1);
            if (nu2 <= nu) break;
            nu = nu2;
            ++kk;
          }
          if (cnt >= 2) {
            hi.mcnt = cnt;
            req.kind = RQ_EVAL_MULTI;
            req.nk = cnt;
            req.scal = Real(1.0);
            ph = D_AFTER_MULTI;
            { hi.phase = ph; return; }
          }
        }
        if (MULTI) hi.single = 0;
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] += h.mu;
        const int solved = lu_solve<M>(h.jtj, h.jte, h.dp);
        LM_STAMP(3);
        ++hi.nlss;
        if (!solved) {
          ph = D_REJECT;
          break;
        }
        h.dp_l2 = Real(0.0);
        for (int i = 0; i < M; ++i) {
          const Real t = h.dp[i];
          cool.pdp[i] = h.p[i] + t;
          h.dp_l2 += t * t;
        }
        if (h.dp_l2 <= c.o.eps2sq * h.p_l2) {
          hi.stop = 2;
          ph = D_FINISH;
          break;
        }
        if (h.dp_l2 >= (h.p_l2 + c.o.eps2) / (Real(kEpsilon) * Real(kEpsilon))) {
          hi.stop = 4;
          ph = D_FINISH;
          break;
        }
        clear_req(hi, req);
        req.kind = RQ_DIF_TRIAL;
        for (int i = 0; i < M; ++i) {
          req.p[i] = h.p[i];
          req.q[i] = cool.pdp[i];
          req.dp[i] = h.dp[i];
        }
        req.dp_l2 = h.dp_l2;
        ++hi.nfev;
        ph = D_AFTER_TRIAL;
        LM_STAMP(4);
        { hi.phase = ph; return; }
      } LM_PHASE_END

      LM_PHASE(D_FINISH) {  // lm_core.c:809-841
        if (hi.k >= c.itmax) hi.stop = 3;
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] = cool.diag[i];
        c.info[0] = cool.init_e2;
        c.info[1] = h.p_e2;
        c.info[2] = h.jte_inf;
        c.info[3] = h.dp_l2;
        Real t = -LmLimits<Real>::max();
        for (int i = 0; i < M; ++i)
          if (t < h.jtj[i * M + i]) t = h.jtj[i * M + i];
        c.info[4] = h.mu / t;
        c.info[5] = (Real)hi.k;
        c.info[6] = (Real)hi.stop;
        c.info[7] = (Real)hi.nfev;
        c.info[8] = (Real)hi.njap;
        c.info[9] = (Real)hi.nlss;
        if (c.want_covar) lu_covar<M>(h.jtj, c.covar, h.p_e2, c.n);
        c.ret = (hi.stop != 4 && hi.stop != 7) ? hi.k : kLmError;
        clear_req(hi, req);
        ph = D_DONE;
        { hi.phase = ph; return; }
      } LM_PHASE_END
    }
  }
};

// =================================================================================================
// dlevmar_bc_dif = dlevmar_bc_der driven by a finite-difference Jacobian (lmbc_core.c:369-1129):
// projected LM step; if it does not reduce the error enough, Schnabel's backtracking line search
// along it; if that is not a descent direction or fails, a projected-gradient search.
// =================================================================================================
template <int M, class Real = double>
struct BcMachine {
  enum Phase : int {
    B_INIT_EVAL = 1, B_ITER_TOP, B_AFTER_JAC, B_SOLVE, B_AFTER_LM_EVAL, B_AFTER_LM_NORM, B_LM_JUDGE, B_LS_PROLOGUE,
    B_LS_ISSUE, B_LS_EVAL, B_PG_BEGIN, B_PG_ISSUE, B_PG_EVAL, B_PG_NORM, B_PG_JUDGE, B_PG_MULTI, B_COMMIT,
    B_END_ITER, B_FINISH, B_DONE
  };
  struct Cold {  // configuration + results (see DifMachine for the Hot/Cold rationale)
    FitOptionsT<Real> o;
    int itmax, n, want_covar;
    int has_lb, has_ub, has_dscl;
    Real lb[M], ub[M], dscl[M];
    int infeasible_mask, bad_input;
    int bad_config;    // configure(): 0, or the argument check that failed (becomes bad_input of every fit begun)
    int analytic_jac;  // 1: dlevmar_bc_der (caller's Jacobian): no nfev correction at the end (lmbc_core.c:1119-1124)
    int multi;         // candidates evaluated per pass in the projected-gradient search (1 = one at a time).  The
                       // search of lmbc_core.c:885-935 tries t, 0.9t, 0.81t, ... along one fixed direction: the
                       // next points are known before the current one has been judged, so K of them share a sweep.
                       // Candidates are judged in the reference's order and only the judged ones count in nfev:
                       // the trajectory and info[] are exactly those of the one-at-a-time search.
    int spec_jac;      // 1: a candidate that becomes p when it is taken (the LM trial point, a line-search point, the start)
                       // is evaluated by a JACOBIAN pass at that point (RQ_JAC: its sum of squares is the evaluation's,
                       // bit for bit; its J^T J / J^T e are kept).  When the candidate is taken, the Jacobian pass the next
                       // iteration opens with (lmbc_core.c:555-561 at the same point, same steps) is already done: one pass
                       // per accepted iteration instead of two.  Where a pass costs its latency, not its arithmetic (one fit
                       // spread over the chip), that is the gain; the batched kernels, bound by arithmetic, leave it off.
                       // Counters (nfev, njev) advance exactly as without it: info[] and the trajectory do not change.
                       // (step<..., SPECJ = true> only; never with dscl, whose scaled points do not round-trip.)
    Real p_start[M];
    Real info[kInfoSz], covar[M * M];
    int ret;
  };
  struct Core {  // everything an LM step reads or writes, except the request it leaves
    int phase, k, stop, nu, nfev, njev, nlss, gprev;
    int spec_state;  // spec_jac: 0 nothing kept, 1 cool.sj holds the Jacobian sums at pdp, 2 ... at p, 3 B_AFTER_JAC reads them
    Real p[M], mu, p_e2, jte_inf, p_l2, dp_l2, pdp_e2;
    Real jtj[M * M], jte[M], dp[M], pdp[M];
    Real t, gdp;
    int ls_first, ls_left;
    int pg_n, pg_single;  // candidates in flight; force the next projected-gradient request to a single candidate
  };
  struct Cool {  // the less busy half of the state (see Hot)
    Real init_e2, keep_max, diag[M], t0;
    Real sj[SumLayout<M>::JAC];  // spec_jac: the sums of the Jacobian pass that evaluated the candidate
    // line-search locals (lmbc_core.c:218-225)
    Real ls_f0, ls_lambda, ls_plmbda, ls_pfpls, ls_tlmbda, ls_rmnlmb, ls_slp;
  };
  // Core + Cool + the request.  run() takes them separately, so that a kernel can step on a REGISTER copy of Core
  // while the rest stays in LDS (resident_fit.hip: the sweeping waves read the request there; with Cool in registers as
  // well, hipcc spilled ~100 VGPRs of the step to scratch, which cost more than the LDS round trips it saved)
  struct Hot : Core {
    Cool cool;
    Request<M, Real> req;
  };
  // where ONE machine is stepped by a whole wave (every lane the same values): moves the counters and flags of a
  // register copy into scalar registers, so that the step's integer logic and branches run on the scalar unit
  static LM_HD void uniform_ints(Core &h) {
    h.phase = lm_uniform(h.phase);
    h.k = lm_uniform(h.k);
    h.stop = lm_uniform(h.stop);
    h.nu = lm_uniform(h.nu);
    h.nfev = lm_uniform(h.nfev);
    h.njev = lm_uniform(h.njev);
    h.nlss = lm_uniform(h.nlss);
    h.gprev = lm_uniform(h.gprev);
    h.spec_state = lm_uniform(h.spec_state);
    h.ls_first = lm_uniform(h.ls_first);
    h.ls_left = lm_uniform(h.ls_left);
    h.pg_n = lm_uniform(h.pg_n);
    h.pg_single = lm_uniform(h.pg_single);
  }
  Cold c;
  Hot h;

  LM_HD static Real median3(Real a, Real b, Real c) {  // lmbc_core.c:59-61
    return (a >= b) ? ((c >= a) ? a : ((c <= b) ? b : c)) : ((c >= b) ? b : ((c <= a) ? a : c));
  }
  // v must not alias the machine (callers pass a local array): the box is read once, up front -- on the
  // device every dependent re-read of the LDS-resident machine costs a full LDS round trip
  static LM_HD void project(const Cold &c, Real *v) {  // lmbc_core.c:68-88
    const int has_lb = c.has_lb, has_ub = c.has_ub;
    if (!has_lb && !has_ub) return;
    Real lo[M], hi[M];
    for (int i = 0; i < M; ++i) {
      lo[i] = c.lb[i];
      hi[i] = c.ub[i];
    }
    for (int i = M; i-- > 0;) {
      if (has_lb && has_ub)
        v[i] = median3(lo[i], v[i], hi[i]);
      else if (has_ub) {
        if (v[i] > hi[i]) v[i] = hi[i];
      } else {
        if (v[i] < lo[i]) v[i] = lo[i];
      }
    }
  }
  static LM_HD void clear_req(const Core &h, Request<M, Real> &req) {
    req.kind = RQ_DONE;
    req.central = 0;
    req.sel_hx = req.sel_j = req.aux = 0;
    req.dp_l2 = Real(0.0);
    req.scal = Real(1.0);
    for (int i = 0; i < M; ++i) req.p[i] = req.d[i] = req.q[i] = req.dp[i] = Real(0.0);
  }
  // ask for ||x - f(v)||^2 where v lives in the (possibly scaled) search space.  Only the fields an
  // evaluation pass reads are written (kind, p, scal); v is a local array.
  static LM_HD void request_eval(const Cold &c, Core &h, Request<M, Real> &req, const Real *v, int kind = RQ_EVAL) {
    req.kind = kind;
    req.scal = Real(1.0);
    if (c.has_dscl) {
      for (int i = 0; i < M; ++i) req.p[i] = v[i] * c.dscl[i];
    } else {
      for (int i = 0; i < M; ++i) req.p[i] = v[i];
    }
    ++h.nfev;
  }
  // the same evaluation, asked for as a Jacobian pass at v (Cold::spec_jac); v is unscaled (never used with dscl)
  static LM_HD void request_eval_jac(const Cold &c, Core &h, Request<M, Real> &req, const Real *v) {
    req.kind = RQ_JAC;
    req.scal = Real(1.0);
    req.central = !c.o.forward;
    for (int i = 0; i < M; ++i) req.p[i] = v[i];
    fd_steps<M>(req.p, c.o.delta, req.d);
    ++h.nfev;
    h.spec_state = 1;
  }

  // start() = configure() + begin().  configure() fills the part of Cold every fit of a batch shares (options, box, limits,
  // flags; the reference's argument checks, which read nothing else); begin() is the per-fit part.  Where 64 machines step
  // side by side in the lanes of one wave (lane_fit.hip) the kernel configures ONCE, in wave-uniform control flow, so that
  // the shared half stays in scalar registers, and begins a fit per lane.
  LM_HD void configure(int n_, const Real *lb_, const Real *ub_, const Real *dscl_, int itmax_, const Real *opts,
                       int want_covar_, int multi_ = 1, int spec_jac_ = 0) {
    c.multi = (multi_ < 1) ? 1 : ((multi_ > kMaxCand) ? kMaxCand : multi_);
    c.o = make_options(opts);
    if (opts) {  // bc_dif reads delta as |opts[4]| and the sign as the FD flavour: lmbc_core.c:1105,1115
      c.o.forward = (opts[4] >= Real(0.0));
      c.o.delta = lm_abs(opts[4]);
    }
    c.itmax = itmax_;
    c.n = n_;
    c.want_covar = want_covar_;
    c.has_lb = lb_ != nullptr;
    c.has_ub = ub_ != nullptr;
    c.has_dscl = dscl_ != nullptr;
    c.analytic_jac = 0;
    c.spec_jac = (spec_jac_ && !c.has_dscl) ? 1 : 0;
    for (int i = 0; i < M; ++i) {
      c.lb[i] = c.has_lb ? lb_[i] : -LmLimits<Real>::max();
      c.ub[i] = c.has_ub ? ub_[i] : LmLimits<Real>::max();
      c.dscl[i] = c.has_dscl ? dscl_[i] : Real(1.0);
    }
    c.bad_config = 0;
    if (c.n < M) {  // lmbc_core.c:440-443
      c.bad_config = 1;
      return;
    }
    if (c.has_lb && c.has_ub)  // lmbc_core.c:451-454 (box_check, misc_core.c:661-671)
      for (int i = 0; i < M; ++i)
        if (c.lb[i] > c.ub[i]) {
          c.bad_config = 2;
          return;
        }
    if (c.has_dscl)  // lmbc_core.c:456-461
      for (int i = M; i-- > 0;)
        if (c.dscl[i] <= Real(0.0)) {
          c.bad_config = 3;
          return;
        }
  }
  LM_HD void begin(const Real *p0) {
    Request<M, Real> &req = h.req;
    Cool &cool = h.cool;
    h.pg_n = h.pg_single = 0;
    h.spec_state = 0;
    h.k = 0;
    h.stop = 0;
    h.nu = 2;
    h.nfev = h.njev = h.nlss = 0;
    h.gprev = 0;
    c.infeasible_mask = 0;
    c.bad_input = c.bad_config;
    h.mu = h.jte_inf = h.p_l2 = h.t = cool.t0 = h.gdp = Real(0.0);
    h.p_e2 = cool.init_e2 = h.pdp_e2 = cool.keep_max = Real(0.0);
    h.dp_l2 = LmLimits<Real>::max();
    cool.ls_f0 = cool.ls_lambda = cool.ls_plmbda = cool.ls_pfpls = cool.ls_tlmbda = cool.ls_rmnlmb = cool.ls_slp = Real(0.0);
    h.ls_first = 1;
    h.ls_left = 0;
    c.ret = kLmError;
    for (int i = 0; i < M; ++i) {
      h.p[i] = p0[i];
      h.jte[i] = cool.diag[i] = h.dp[i] = h.pdp[i] = Real(0.0);
    }
    for (int i = 0; i < M * M; ++i) h.jtj[i] = c.covar[i] = Real(0.0);
    for (int i = 0; i < SumLayout<M>::JAC; ++i) cool.sj[i] = Real(0.0);
    for (int i = 0; i < kInfoSz; ++i) c.info[i] = Real(0.0);
    clear_req(h, req);
    h.phase = B_DONE;
    if (c.bad_input) return;
    for (int i = 0; i < M; ++i) c.p_start[i] = h.p[i];
    project(c, h.p);  // lmbc_core.c:514-520; the stderr warning is printed by the host shim from the mask
    for (int i = 0; i < M; ++i)
      if (c.p_start[i] != h.p[i]) c.infeasible_mask |= (1 << i);
    clear_req(h, req);
    req.kind = RQ_EVAL;  // the first evaluation is at the unscaled projected start, lmbc_core.c:523
    for (int i = 0; i < M; ++i) req.p[i] = h.p[i];
    if (c.spec_jac) {  // ... as the Jacobian pass the first iteration opens with (same point: no dscl here)
      request_eval_jac(c, h, req, h.p);
      h.nfev = 0;
    }
    h.phase = B_INIT_EVAL;
  }
  LM_HD void start(const Real *p0, int n_, const Real *lb_, const Real *ub_, const Real *dscl_,
                   int itmax_, const Real *opts, int want_covar_, int multi_ = 1, int spec_jac_ = 0) {
    configure(n_, lb_, ub_, dscl_, itmax_, opts, want_covar_, multi_, spec_jac_);
    begin(p0);
  }
  // (the analytic-Jacobian flag is set by the callers after start(): the kind of the FIRST request does not depend on it)

  template <bool SPECJ = false>
  static LM_HD void accept_trial(Core &h) {  // p <- pdp, ||e||^2 <- trial value
    for (int i = 0; i < M; ++i) h.p[i] = h.pdp[i];
    h.p_e2 = h.pdp_e2;
    if (SPECJ) h.spec_state = (h.spec_state == 1) ? 2 : 0;  // the Jacobian sums kept for pdp are now those of p
  }
  // the sum of squares a candidate's pass returned: slot 0 of an evaluation pass, the last slot of a Jacobian pass (whose
  // other sums are kept for the iteration that may open at this point)
  template <bool SPECJ>
  static LM_HD Real candidate_e2(Core &h, Cool &cool, const Real *s) {
    if (SPECJ && h.spec_state == 1) {
      for (int i = 0; i < SumLayout<M>::JAC; ++i) cool.sj[i] = s[i];
      return s[SumLayout<M>::JAC - 1];
    }
    return s[0];
  }

  // MULTI = false compiles the multi-candidate projected-gradient machinery out (callers that start with multi = 1).
  // GATED: the expensive phases -- everything behind a Jacobian pass (B_AFTER_JAC, B_SOLVE: the 3x3 LU), the line-search
  // prologue (pow, square roots, divisions) and B_FINISH -- only run when `heavy` is set; otherwise the step stops in front
  // of them with RQ_YIELD.  Pure scheduling: a machine's trajectory does not depend on when its phases run.
  // SPECJ: compiles the speculative Jacobian passes in (Cold::spec_jac turns them on)
  template <bool ONE_LANE = false, bool MULTI = true, bool GATED = false, bool SPECJ = false>
  LM_HD void step(const Real *s, Real maxabs, bool heavy = true) { run<ONE_LANE, MULTI, GATED, SPECJ>(c, h, h.cool, h.req, s, maxabs, heavy); }

  template <bool ONE_LANE, bool MULTI, bool GATED, bool SPECJ = false>
  static LM_HD void run(Cold &c, Core &h, Cool &cool, Request<M, Real> &req, const Real *s, Real maxabs, bool heavy) {
    constexpr Real alpha = Real(1e-4), beta = Real(0.9), gamma = Real(0.99995), rho = Real(1e-8), tming = Real(1e-18), tini = Real(1.0);
    int ph = ONE_LANE ? lm_uniform(h.phase) : h.phase;  // scalar register: phase transitions become scalar jumps
    for (;;) {
      (void)LM_PHASE_ENTER(29);  // (diagnostic builds: a trip of the dispatch loop starts)
      if (ONE_LANE) ph = lm_uniform(ph);  // re-assert uniformity: assignments under (formally) divergent branches lose it
      // The phases are a chain of guarded blocks in the order control flows through them, not a switch: a lane (or
      // the one stepping lane) walks through every block its phase reaches in ONE trip of this loop -- only a failed
      // 3x3 solve (B_SOLVE again) needs a second trip.  That matters where 64 machines step side by side in the lanes
      // of one wave (lane_fit.hip): a switch inside a loop executes every phase body once per trip and a lane takes 2-5
      // trips per step, i.e. the expensive bodies (LU, line-search interpolation, pow) would run several times per step.
      // `break` inside a block leaves the block (do { } while (0)), exactly as it left the switch before.
      if (ph <= 0 || ph >= B_DONE) {
        req.kind = RQ_DONE;
        { h.phase = ph; return; }
      }
      LM_PHASE(B_INIT_EVAL) {  // lmbc_core.c:523-540
        h.nfev = 1;
        h.p_e2 = candidate_e2<SPECJ>(h, cool, s);
        if (SPECJ && h.spec_state == 1) h.spec_state = 2;  // the start IS p
        cool.init_e2 = h.p_e2;
        if (!lm_finite(h.p_e2)) h.stop = 7;
        if (c.has_dscl)
          for (int i = M; i-- > 0;) {
            h.p[i] /= c.dscl[i];
            if (c.has_ub && c.ub[i] != LmLimits<Real>::max()) c.ub[i] = c.ub[i] / c.dscl[i];
            if (c.has_lb && c.lb[i] != -LmLimits<Real>::max()) c.lb[i] = c.lb[i] / c.dscl[i];
          }
        ph = B_ITER_TOP;
        break;
      } LM_PHASE_END

      LM_GATE(B_AFTER_JAC)
      if (!GATED || heavy)
      LM_PHASE(B_AFTER_JAC) {
        const Real *sv = s;
        if (SPECJ && h.spec_state == 3) sv = cool.sj;  // the pass that evaluated this point as a candidate left them
        if (SPECJ) h.spec_state = 0;
        unpack_lower<M>(sv, h.jtj);
        for (int i = 0; i < M; ++i) h.jte[i] = sv[SumLayout<M>::NL + i];
        if (c.has_dscl) {  // J <- J*D (lmbc_core.c:562-569) folded into the reduced products
          for (int i = 0; i < M; ++i) {
            h.jte[i] *= c.dscl[i];
            for (int j = 0; j < M; ++j) h.jtj[i * M + j] *= c.dscl[i] * c.dscl[j];
          }
        }
        int nactive = 0, satisfied = 0;  // lmbc_core.c:639-646
        h.p_l2 = h.jte_inf = Real(0.0);
        for (int i = 0; i < M; ++i) {
          if (c.has_ub && h.p[i] == c.ub[i]) {
            ++nactive;
            if (h.jte[i] > Real(0.0)) ++satisfied;
          } else if (c.has_lb && h.p[i] == c.lb[i]) {
            ++nactive;
            if (h.jte[i] < Real(0.0)) ++satisfied;
          } else {
            const Real a = lm_abs(h.jte[i]);
            if (h.jte_inf < a) h.jte_inf = a;
          }
          cool.diag[i] = h.jtj[i * M + i];
          h.p_l2 += h.p[i] * h.p[i];
        }
        if (satisfied == nactive && (h.jte_inf <= c.o.eps1)) {
          h.dp_l2 = Real(0.0);
          h.stop = 1;
          ph = B_FINISH;
          break;
        }
        if (h.k == 0) {  // lmbc_core.c:666-674
          if (!c.has_lb && !c.has_ub) {
            Real m0 = -LmLimits<Real>::max();
            for (int i = 0; i < M; ++i)
              if (cool.diag[i] > m0) m0 = cool.diag[i];
            h.mu = c.o.tau * m0;
          } else
            h.mu = Real(0.5) * c.o.tau * h.p_e2;  // Kanzow's starting damping
        }
        ph = B_SOLVE;
        break;
      } LM_PHASE_END

      LM_GATE(B_SOLVE)
      if (!GATED || heavy)
      LM_PHASE(B_SOLVE) {  // lmbc_core.c:677-734
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] += h.mu;
        const int solved = lu_solve<M>(h.jtj, h.jte, h.dp);
        ++h.nlss;
        if (!solved) {  // :788-804
          h.mu *= h.nu;
          const int nu2 = (int)((unsigned)h.nu << 1);
          if (nu2 <= h.nu) {
            h.stop = 5;
            ph = B_END_ITER;
            break;
          }
          h.nu = nu2;
          for (int i = 0; i < M; ++i) h.jtj[i * M + i] = cool.diag[i];
          break;  // solve again
        }
        Real pc[M], v[M], l2 = Real(0.0);
        for (int i = 0; i < M; ++i) {
          pc[i] = h.p[i];
          v[i] = pc[i] + h.dp[i];
        }
        project(c, v);
        for (int i = 0; i < M; ++i) {
          const Real d = v[i] - pc[i];
          h.pdp[i] = v[i];
          h.dp[i] = d;
          l2 += d * d;
        }
        h.dp_l2 = l2;
        const Real pl2 = h.p_l2;
        if (l2 <= c.o.eps2sq * pl2) {
          h.stop = 2;
          ph = B_END_ITER;
          break;
        }
        if (l2 >= (pl2 + c.o.eps2) / (Real(kEpsilon) * Real(kEpsilon))) {
          h.stop = 4;
          ph = B_END_ITER;
          break;
        }
        if (SPECJ && c.spec_jac)
          request_eval_jac(c, h, req, v);
        else
          request_eval(c, h, req, v);
        ph = B_AFTER_LM_EVAL;
        { h.phase = ph; return; }
      } LM_PHASE_END

      LM_PHASE(B_AFTER_LM_EVAL) {  // overflow guard, lmbc_core.c:748-751
        h.pdp_e2 = candidate_e2<SPECJ>(h, cool, s);
        if (!lm_finite(h.pdp_e2)) {
          if (SPECJ && h.spec_state == 1) {  // the guard wants max |e|, which a Jacobian pass does not form: evaluate plainly
            h.spec_state = 0;
            request_eval(c, h, req, h.pdp);
            --h.nfev;  // (the same evaluation, asked for again)
            { h.phase = ph; return; }
          }
          if (!lm_finite(maxabs)) {
            h.stop = 7;
            ph = B_END_ITER;
            break;
          }
          cool.keep_max = maxabs;
          request_eval(c, h, req, h.pdp, RQ_SCALED);
          --h.nfev;  // not a user-visible function evaluation
          req.scal = maxabs;
          ph = B_AFTER_LM_NORM;
          { h.phase = ph; return; }
        }
        ph = B_LM_JUDGE;
        break;
      } LM_PHASE_END

      LM_PHASE(B_AFTER_LM_NORM) {
        if (!lm_finite(cool.keep_max * sqrt(s[0]))) {
          h.stop = 7;
          ph = B_END_ITER;
          break;
        }
        ph = B_LM_JUDGE;
        break;
      } LM_PHASE_END

      LM_PHASE(B_LM_JUDGE) {
        if (h.pdp_e2 <= gamma * h.p_e2) {  // LM step taken, lmbc_core.c:753-785
          Real dL = Real(0.0);
          for (int i = 0; i < M; ++i) dL += h.dp[i] * (h.mu * h.dp[i] + h.jte[i]);
          if (dL > Real(0.0)) {
            const Real dF = h.p_e2 - h.pdp_e2;
            Real q = (Real(2.0) * dF / dL - Real(1.0));
            q = Real(1.0) - q * q * q;
            h.mu = h.mu * ((q >= Real(kOneThird)) ? q : Real(kOneThird));
          } else {
            const Real q = Real(0.1) * h.pdp_e2;
            h.mu = (h.mu >= q) ? q : h.mu;
          }
          h.nu = 2;
          accept_trial<SPECJ>(h);
          h.gprev = 0;
          ph = B_END_ITER;
          break;
        }
        ph = B_LS_PROLOGUE;
        break;
      } LM_PHASE_END

      LM_GATE(B_LS_PROLOGUE)
      if (!GATED || heavy)
      LM_PHASE(B_LS_PROLOGUE) {  // the LM step was rejected
        h.gdp = Real(0.0);  // lmbc_core.c:811-816
        for (int i = 0; i < M; ++i) {
          h.jte[i] = -h.jte[i];
          h.gdp += h.jte[i] * h.dp[i];
        }
        if (!(h.gdp <= -rho * pow(h.dp_l2, Real(kLsPow) / Real(2.0)))) {
          ph = B_PG_BEGIN;
          break;
        }
        // ---- line-search prologue, lmbc_core.c:227-249 (x = p, f = p_e2, g = jte, step = dp)
        const Real steptl = Real(1e3) * sqrt(LmLimits<Real>::eps());
        Real pn = sqrt(h.p_l2);
        const Real stepmx = Real(1e3) * ((pn >= Real(1.0)) ? pn : Real(1.0));
        cool.ls_f0 = h.p_e2 * Real(0.5);
        Real acc = Real(0.0);
        for (int i = M; i-- > 0;) acc += h.dp[i] * h.dp[i];
        Real sln = sqrt(acc);
        if (sln > stepmx) {
          const Real scl = stepmx / sln;
          for (int i = M; i-- > 0;) h.dp[i] *= scl;
          sln = stepmx;
        }
        Real rln = Real(0.0);
        cool.ls_slp = Real(0.0);
        for (int i = M; i-- > 0;) {
          cool.ls_slp += h.jte[i] * h.dp[i];
          const Real den = (lm_abs(h.p[i]) >= Real(1.0)) ? lm_abs(h.p[i]) : Real(1.0);
          const Real rel = lm_abs(h.dp[i]) / den;
          if (rln < rel) rln = rel;
        }
        cool.ls_rmnlmb = steptl / rln;
        cool.ls_lambda = Real(1.0);
        h.ls_first = 1;
        cool.ls_plmbda = cool.ls_pfpls = cool.ls_tlmbda = Real(0.0);
        h.ls_left = kLsItMax;
        ph = B_LS_ISSUE;
        break;
      } LM_PHASE_END

      LM_PHASE(B_LS_EVAL) {  // lmbc_core.c:269-332
        const Real e2 = candidate_e2<SPECJ>(h, cool, s);
        const Real fpls = Real(0.5) * e2;
        h.pdp_e2 = e2;
        if (fpls <= cool.ls_f0 + cool.ls_slp * alpha * cool.ls_lambda) {  // satisfactory point
          if (!lm_finite(h.pdp_e2)) {  // lmbc_core.c:828
            ph = B_PG_BEGIN;
            break;
          }
          h.gprev = 0;
          ph = B_COMMIT;
          break;
        }
        if (cool.ls_lambda < cool.ls_rmnlmb) {
          ph = B_PG_BEGIN;
          break;
        }
        if (!lm_finite(fpls)) {
          cool.ls_lambda *= Real(0.1);
          h.ls_first = 1;
        } else {
          if (h.ls_first) {
            cool.ls_tlmbda = -cool.ls_lambda * cool.ls_slp / ((fpls - cool.ls_f0 - cool.ls_slp) * Real(2.0));
            h.ls_first = 0;
          } else {
            const Real t1 = fpls - cool.ls_f0 - cool.ls_lambda * cool.ls_slp;
            const Real t2 = cool.ls_pfpls - cool.ls_f0 - cool.ls_plmbda * cool.ls_slp;
            const Real t3 = Real(1.0) / (cool.ls_lambda - cool.ls_plmbda);
            const Real a3 = Real(3.0) * t3 * (t1 / (cool.ls_lambda * cool.ls_lambda) - t2 / (cool.ls_plmbda * cool.ls_plmbda));
            const Real b = t3 * (t2 * cool.ls_lambda / (cool.ls_plmbda * cool.ls_plmbda) - t1 * cool.ls_plmbda / (cool.ls_lambda * cool.ls_lambda));
            const Real disc = b * b - a3 * cool.ls_slp;
            if (disc > b * b)
              cool.ls_tlmbda = (-b + ((a3 < 0) ? -sqrt(disc) : sqrt(disc))) / a3;
            else
              cool.ls_tlmbda = (-b + ((a3 < 0) ? sqrt(disc) : -sqrt(disc))) / a3;
            if (cool.ls_tlmbda > cool.ls_lambda * Real(0.5)) cool.ls_tlmbda = cool.ls_lambda * Real(0.5);
          }
          cool.ls_plmbda = cool.ls_lambda;
          cool.ls_pfpls = fpls;
          if (cool.ls_tlmbda < cool.ls_lambda * Real(0.1))
            cool.ls_lambda *= Real(0.1);
          else
            cool.ls_lambda = cool.ls_tlmbda;
        }
        ph = B_LS_ISSUE;
        break;
      } LM_PHASE_END

      LM_PHASE(B_LS_ISSUE) {  // lmbc_core.c:253-266
        if (h.ls_left-- <= 0) {  // iteration limit: failure -> projected gradient
          ph = B_PG_BEGIN;
          break;
        }
        Real v[M];
        const Real lam = cool.ls_lambda;
        for (int i = M; i-- > 0;) v[i] = h.p[i] + lam * h.dp[i];
        project(c, v);
        if (SPECJ && c.spec_jac) {  // (no dscl with spec_jac)
          for (int i = 0; i < M; ++i) h.pdp[i] = v[i];
          request_eval_jac(c, h, req, v);
          ph = B_LS_EVAL;
          { h.phase = ph; return; }
        }
        req.kind = RQ_EVAL;
        req.scal = Real(1.0);
        if (!c.has_dscl) {
          for (int i = 0; i < M; ++i) {
            req.p[i] = v[i];
            h.pdp[i] = v[i];
          }
        } else {  // the reference multiplies and divides xpls in place, lmbc_core.c:263-265
          for (int i = M; i-- > 0;) {
            v[i] *= c.dscl[i];
            req.p[i] = v[i];
            v[i] /= c.dscl[i];
            h.pdp[i] = v[i];
          }
        }
        ++h.nfev;
        ph = B_LS_EVAL;
        { h.phase = ph; return; }
      } LM_PHASE_END

      LM_PHASE(B_PG_EVAL) {
        h.pdp_e2 = s[0];
        if (!lm_finite(h.pdp_e2)) {  // lmbc_core.c:915-918
          if (!lm_finite(maxabs)) {
            h.stop = 7;
            ph = B_FINISH;
            break;
          }
          cool.keep_max = maxabs;
          Real v[M];
          for (int i = 0; i < M; ++i) v[i] = h.pdp[i];
          request_eval(c, h, req, v, RQ_SCALED);
          --h.nfev;
          req.scal = maxabs;
          ph = B_PG_NORM;
          { h.phase = ph; return; }
        }
        ph = B_PG_JUDGE;
        break;
      } LM_PHASE_END

      LM_PHASE(B_PG_NORM) {
        if (!lm_finite(cool.keep_max * sqrt(s[0]))) {
          h.stop = 7;
          ph = B_FINISH;  // "goto breaknested": k is not advanced
          break;
        }
        ph = B_PG_JUDGE;
        break;
      } LM_PHASE_END

      if constexpr (MULTI)
      LM_PHASE(B_PG_MULTI) {  // judge the candidates of one sweep in the reference's order (lmbc_core.c:886-935)
        Real pc[M], g[M];
        for (int i = 0; i < M; ++i) {
          pc[i] = h.p[i];
          g[i] = h.jte[i];
        }
        const Real fold = h.p_e2;
        const int cnt = h.pg_n;
        Real tt = h.t;
        int next = B_PG_ISSUE;
        for (int j = 0; j < kMaxCand; ++j) {
          if (j >= cnt) break;
          Real v[M], d[M], l2 = Real(0.0), gd = Real(0.0);
          for (int i = 0; i < M; ++i) v[i] = pc[i] - tt * g[i];
          project(c, v);
          for (int i = 0; i < M; ++i) {
            d[i] = v[i] - pc[i];
            l2 += d[i] * d[i];
          }
          const Real fnew = s[j];
          if (!lm_finite(fnew)) {  // overflow guard needs max|e| of this candidate: evaluate it alone
            h.pg_single = 1;
            break;
          }
          ++h.nfev;
          for (int i = 0; i < M; ++i) {
            h.pdp[i] = v[i];
            h.dp[i] = d[i];
            gd += g[i] * d[i];
          }
          h.dp_l2 = l2;
          h.pdp_e2 = fnew;
          h.gdp = gd;
          if (h.gprev && fnew <= fold + Real(2.0) * Real(0.99999) * gd) {  // remembered t was too small
            tt = cool.t0;
            h.gprev = 0;
            tt *= beta;
            break;
          }
          if (fnew <= fold + Real(2.0) * alpha * gd) {
            h.gprev = 1;
            next = B_COMMIT;
            break;
          }
          tt *= beta;
        }
        h.t = tt;
        ph = next;
        break;
      } LM_PHASE_END

      LM_PHASE(B_PG_JUDGE) {  // lmbc_core.c:923-935
        Real g = Real(0.0);
        for (int i = 0; i < M; ++i) g += h.jte[i] * h.dp[i];
        h.gdp = g;
        const Real fnew = h.pdp_e2, fold = h.p_e2;
        if (h.gprev && fnew <= fold + Real(2.0) * Real(0.99999) * g) {  // remembered t was too small
          Real tt = cool.t0;
          h.gprev = 0;
          tt *= beta;  // the reference's `continue` still runs the loop increment
          h.t = tt;
          ph = B_PG_ISSUE;
          break;
        }
        if (fnew <= fold + Real(2.0) * alpha * g) {
          h.gprev = 1;
          ph = B_COMMIT;
          break;
        }
        h.t = h.t * beta;
        ph = B_PG_ISSUE;
        break;
      } LM_PHASE_END

      LM_PHASE(B_PG_BEGIN) {  // lmbc_core.c:877-885 (jte already holds -J^T e)
        Real g2 = Real(0.0);
        for (int i = 0; i < M; ++i) g2 += h.jte[i] * h.jte[i];
        g2 = sqrt(g2);
        g2 = Real(100.0) / (Real(1.0) + g2);
        cool.t0 = (g2 <= tini) ? g2 : tini;
        h.t = h.gprev ? h.t : cool.t0;
        ph = B_PG_ISSUE;
        break;
      } LM_PHASE_END

      LM_PHASE(B_PG_ISSUE) {  // loop head of lmbc_core.c:885
        if (SPECJ) h.spec_state = 0;  // projected-gradient candidates are plain evaluations: nothing kept for them
        if (!(h.t > tming)) {  // search failed, :937-939
          h.gprev = 0;
          ph = B_END_ITER;
          break;
        }
        Real pc[M], g[M];
        for (int i = 0; i < M; ++i) {
          pc[i] = h.p[i];
          g[i] = h.jte[i];
        }
        const int want = MULTI ? (h.pg_single ? 1 : c.multi) : 1;
        h.pg_single = 0;
        Real tt = h.t;
        int cnt = 0;
        Real v0[M];
        for (int j = 0; j < kMaxCand; ++j) {  // t, t*beta, t*beta^2, ... exactly as the loop increment forms them
          if (j >= want || !(tt > tming)) break;
          Real v[M];
          for (int i = 0; i < M; ++i) v[i] = pc[i] - tt * g[i];
          project(c, v);
          for (int i = 0; i < M; ++i) {
            if (j == 0) v0[i] = v[i];
            req.pk[j][i] = c.has_dscl ? v[i] * c.dscl[i] : v[i];
          }
          ++cnt;
          tt *= beta;
        }
        if (cnt <= 1) {  // one candidate: the plain evaluation request
          Real l2 = Real(0.0);
          for (int i = 0; i < M; ++i) {
            const Real d = v0[i] - pc[i];
            h.pdp[i] = v0[i];
            h.dp[i] = d;
            l2 += d * d;
          }
          h.dp_l2 = l2;
          request_eval(c, h, req, v0);
          ph = B_PG_EVAL;
          { h.phase = ph; return; }
        }
        h.pg_n = cnt;
        req.kind = RQ_EVAL_MULTI;
        req.nk = cnt;
        req.scal = Real(1.0);
        ph = B_PG_MULTI;
        { h.phase = ph; return; }
      } LM_PHASE_END

      LM_PHASE(B_COMMIT) {  // lmbc_core.c:950-967
        h.dp_l2 = Real(0.0);
        for (int i = 0; i < M; ++i) {
          const Real d = h.pdp[i] - h.p[i];
          h.dp_l2 += d * d;
        }
        if (h.dp_l2 <= c.o.eps2sq * h.p_l2) {
          h.stop = 2;
          ph = B_END_ITER;
          break;
        }
        accept_trial<SPECJ>(h);
        ph = B_END_ITER;
        break;
      } LM_PHASE_END

      LM_PHASE(B_END_ITER) {
        ++h.k;
        ph = B_ITER_TOP;
        break;
      } LM_PHASE_END

      LM_PHASE(B_ITER_TOP) {
        if (!(h.k < c.itmax && !h.stop)) {
          ph = B_FINISH;
          break;
        }
        if (h.p_e2 <= c.o.eps3) {
          h.stop = 6;
          ph = B_FINISH;
          break;
        }
        if (SPECJ && h.spec_state == 2) {  // this point was evaluated by a Jacobian pass when it was a candidate
          h.spec_state = 3;
          ++h.njev;
          ph = B_AFTER_JAC;
          break;
        }
        clear_req(h, req);  // FD Jacobian at the unscaled point, lmbc_core.c:555-561 + :1043-1054
        req.kind = RQ_JAC;
        req.central = !c.o.forward;
        for (int i = 0; i < M; ++i) req.p[i] = c.has_dscl ? h.p[i] * c.dscl[i] : h.p[i];
        fd_steps<M>(req.p, c.o.delta, req.d);
        ++h.njev;
        ph = B_AFTER_JAC;
        { h.phase = ph; return; }
      } LM_PHASE_END

      LM_GATE(B_FINISH)
      if (!GATED || heavy)
      LM_PHASE(B_FINISH) {  // lmbc_core.c:973-1021, :1119-1124
        if (h.k >= c.itmax) h.stop = 3;
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] = cool.diag[i];
        c.info[0] = cool.init_e2;
        c.info[1] = h.p_e2;
        c.info[2] = h.jte_inf;
        c.info[3] = h.dp_l2;
        Real m0 = -LmLimits<Real>::max();
        for (int i = 0; i < M; ++i)
          if (m0 < h.jtj[i * M + i]) m0 = h.jtj[i * M + i];
        c.info[4] = h.mu / m0;
        c.info[5] = (Real)h.k;
        c.info[6] = (Real)h.stop;
        c.info[7] = (Real)h.nfev + (c.analytic_jac ? Real(0.0) : (Real)h.njev * (c.o.forward ? (M + 1) : (2 * M)));
        c.info[8] = (Real)h.njev;
        c.info[9] = (Real)h.nlss;
        if (c.want_covar) {
          lu_covar<M>(h.jtj, c.covar, h.p_e2, c.n);
          if (c.has_dscl)
            for (int i = M; i-- > 0;)
              for (int j = M; j-- > 0;) c.covar[i * M + j] *= (c.dscl[i] * c.dscl[j]);
        }
        if (c.has_dscl)
          for (int i = 0; i < M; ++i) h.p[i] *= c.dscl[i];
        c.ret = (h.stop != 4 && h.stop != 7) ? h.k : kLmError;
        clear_req(h, req);
        ph = B_DONE;
        { h.phase = ph; return; }
      } LM_PHASE_END
    }
  }
};

// =================================================================================================
// dlevmar_der: unconstrained LM with the caller's analytic Jacobian (lm_core.c:64-432).  Every outer
// iteration asks for the Jacobian's normal equations (RQ_JAC: the pass executor fills J from the caller's
// jacf instead of finite differences), then tries damped steps until one reduces the error.
// =================================================================================================
template <int M, class Real = double>
struct DerMachine {
  enum Phase : int { R_INIT_EVAL = 1, R_ITER_TOP, R_AFTER_JAC, R_SOLVE, R_AFTER_EVAL, R_END_ITER, R_FINISH, R_DONE };
  struct Cold {
    FitOptionsT<Real> o;
    int itmax, n, want_covar;
    Real info[kInfoSz], covar[M * M];
    int ret;
  };
  struct Core {  // everything an LM step reads or writes, except the request it leaves
    int phase, k, stop, nu, nfev, njev, nlss;
    Real p[M], mu, p_e2, init_e2, jte_inf, p_l2, dp_l2, pdp_e2;
    Real jtj[M * M], jte[M], diag[M], dp[M], pdp[M];
  };
  // Core + Cool + the request.  run() takes them separately, so that a kernel can step on a REGISTER copy of Core
  // while the rest stays in LDS (resident_fit.hip: the sweeping waves read the request there; with Cool in registers as
  // well, hipcc spilled ~100 VGPRs of the step to scratch, which cost more than the LDS round trips it saved)
  struct Hot : Core {
    Request<M, Real> req;
  };
  // where ONE machine is stepped by a whole wave (every lane the same values): moves the counters and flags of a
  // register copy into scalar registers, so that the step's integer logic and branches run on the scalar unit
  static LM_HD void uniform_ints(Core &h) {
    h.phase = lm_uniform(h.phase);
    h.k = lm_uniform(h.k);
    h.stop = lm_uniform(h.stop);
    h.nu = lm_uniform(h.nu);
    h.nfev = lm_uniform(h.nfev);
    h.njev = lm_uniform(h.njev);
    h.nlss = lm_uniform(h.nlss);
  }
  Cold c;
  Hot h;

  static LM_HD void clear_req(const Core &h, Request<M, Real> &req) {
    req.kind = RQ_DONE;
    req.central = 0;
    req.sel_hx = req.sel_j = req.aux = 0;
    req.dp_l2 = Real(0.0);
    req.scal = Real(1.0);
    for (int i = 0; i < M; ++i) req.p[i] = req.d[i] = req.q[i] = req.dp[i] = Real(0.0);
  }

  LM_HD void start(const Real *p0, int n_, int itmax_, const Real *opts, int want_covar_) {
    Request<M, Real> &req = h.req;
    c.o = make_options(opts);
    c.itmax = itmax_;
    c.n = n_;
    c.want_covar = want_covar_;
    h.k = h.stop = 0;
    h.nu = 2;
    h.nfev = h.njev = h.nlss = 0;
    h.mu = h.jte_inf = h.p_l2 = h.p_e2 = h.init_e2 = h.pdp_e2 = Real(0.0);
    h.dp_l2 = LmLimits<Real>::max();
    c.ret = kLmError;
    for (int i = 0; i < M; ++i) {
      h.p[i] = p0[i];
      h.jte[i] = h.diag[i] = h.dp[i] = h.pdp[i] = Real(0.0);
    }
    for (int i = 0; i < M * M; ++i) h.jtj[i] = c.covar[i] = Real(0.0);
    for (int i = 0; i < kInfoSz; ++i) c.info[i] = Real(0.0);
    clear_req(h, req);
    if (c.n < M) {  // lm_core.c:121-124
      h.phase = R_DONE;
      return;
    }
    req.kind = RQ_EVAL;
    for (int i = 0; i < M; ++i) req.p[i] = h.p[i];
    h.phase = R_INIT_EVAL;
  }

  template <bool ONE_LANE = false>
  LM_HD void step(const Real *s, Real maxabs) { run<ONE_LANE>(c, h, h.req, s, maxabs); }

  template <bool ONE_LANE>
  static LM_HD void run(Cold &c, Core &h, Request<M, Real> &req, const Real *s, Real /*maxabs*/) {
    int ph = ONE_LANE ? lm_uniform(h.phase) : h.phase;
    for (;;) {
      if (ONE_LANE) ph = lm_uniform(ph);  // re-assert uniformity: assignments under (formally) divergent branches lose it
      // (an ordered chain of guarded blocks, not a switch: see BcMachine::run)
      if (ph <= 0 || ph >= R_DONE) {
        req.kind = RQ_DONE;
        { h.phase = ph; return; }
      }
      LM_PHASE(R_INIT_EVAL) {  // lm_core.c:168-179
        h.nfev = 1;
        h.p_e2 = s[0];
        h.init_e2 = h.p_e2;
        if (!lm_finite(h.p_e2)) h.stop = 7;
        ph = R_ITER_TOP;
        break;
      } LM_PHASE_END

      LM_PHASE(R_AFTER_JAC) {  // lm_core.c:262-291
        unpack_lower<M>(s, h.jtj);
        for (int i = 0; i < M; ++i) h.jte[i] = s[SumLayout<M>::NL + i];
        h.p_l2 = h.jte_inf = Real(0.0);
        for (int i = 0; i < M; ++i) {
          const Real t = lm_abs(h.jte[i]);
          if (h.jte_inf < t) h.jte_inf = t;
          h.diag[i] = h.jtj[i * M + i];
          h.p_l2 += h.p[i] * h.p[i];
        }
        if (h.jte_inf <= c.o.eps1) {
          h.dp_l2 = Real(0.0);
          h.stop = 1;
          ph = R_FINISH;
          break;
        }
        if (h.k == 0) {
          Real t = -LmLimits<Real>::max();
          for (int i = 0; i < M; ++i)
            if (h.diag[i] > t) t = h.diag[i];
          h.mu = c.o.tau * t;
        }
        ph = R_SOLVE;
        break;
      } LM_PHASE_END

      LM_PHASE(R_AFTER_EVAL) {  // lm_core.c:347-396
        h.pdp_e2 = s[0];
        if (!lm_finite(h.pdp_e2)) {
          h.stop = 7;
          ph = R_END_ITER;
          break;
        }
        Real dL = Real(0.0);
        for (int i = 0; i < M; ++i) dL += h.dp[i] * (h.mu * h.dp[i] + h.jte[i]);
        const Real dF = h.p_e2 - h.pdp_e2;
        if (dL > Real(0.0) && dF > Real(0.0)) {
          Real t = (Real(2.0) * dF / dL - Real(1.0));
          t = Real(1.0) - t * t * t;
          h.mu = h.mu * ((t >= Real(kOneThird)) ? t : Real(kOneThird));
          h.nu = 2;
          for (int i = 0; i < M; ++i) h.p[i] = h.pdp[i];
          h.p_e2 = h.pdp_e2;
          ph = R_END_ITER;
          break;
        }
        h.mu *= h.nu;
        {
          const int nu2 = (int)((unsigned)h.nu << 1);
          if (nu2 <= h.nu) {
            h.stop = 5;
            ph = R_END_ITER;
            break;
          }
          h.nu = nu2;
        }
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] = h.diag[i];
        ph = R_SOLVE;
        break;
      } LM_PHASE_END

      LM_PHASE(R_SOLVE) {  // the inner while(1), lm_core.c:294-397
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] += h.mu;
        const int solved = lu_solve<M>(h.jtj, h.jte, h.dp);
        ++h.nlss;
        if (solved) {
          h.dp_l2 = Real(0.0);
          for (int i = 0; i < M; ++i) {
            const Real t = h.dp[i];
            h.pdp[i] = h.p[i] + t;
            h.dp_l2 += t * t;
          }
          if (h.dp_l2 <= c.o.eps2sq * h.p_l2) {
            h.stop = 2;
            ph = R_END_ITER;
            break;
          }
          if (h.dp_l2 >= (h.p_l2 + c.o.eps2) / (Real(kEpsilon) * Real(kEpsilon))) {
            h.stop = 4;
            ph = R_END_ITER;
            break;
          }
          clear_req(h, req);
          req.kind = RQ_EVAL;
          for (int i = 0; i < M; ++i) req.p[i] = h.pdp[i];
          ++h.nfev;
          ph = R_AFTER_EVAL;
          h.phase = ph;
          return;
        }
        // unsolvable system: treat as a rejected step
        h.mu *= h.nu;
        {
          const int nu2 = (int)((unsigned)h.nu << 1);
          if (nu2 <= h.nu) {
            h.stop = 5;
            ph = R_END_ITER;
            break;
          }
          h.nu = nu2;
        }
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] = h.diag[i];
        break;  // solve again
      } LM_PHASE_END

      LM_PHASE(R_END_ITER) {
        ++h.k;
        ph = R_ITER_TOP;
        break;
      } LM_PHASE_END

      LM_PHASE(R_ITER_TOP) {
        if (!(h.k < c.itmax && !h.stop)) {
          ph = R_FINISH;
          break;
        }
        if (h.p_e2 <= c.o.eps3) {
          h.stop = 6;
          ph = R_FINISH;
          break;
        }
        clear_req(h, req);
        req.kind = RQ_JAC;
        for (int i = 0; i < M; ++i) req.p[i] = h.p[i];
        ++h.njev;
        ph = R_AFTER_JAC;
        h.phase = ph;
        return;
      } LM_PHASE_END

      LM_PHASE(R_FINISH) {  // lm_core.c:400-431
        if (h.k >= c.itmax) h.stop = 3;
        for (int i = 0; i < M; ++i) h.jtj[i * M + i] = h.diag[i];
        c.info[0] = h.init_e2;
        c.info[1] = h.p_e2;
        c.info[2] = h.jte_inf;
        c.info[3] = h.dp_l2;
        Real t = -LmLimits<Real>::max();
        for (int i = 0; i < M; ++i)
          if (t < h.jtj[i * M + i]) t = h.jtj[i * M + i];
        c.info[4] = h.mu / t;
        c.info[5] = (Real)h.k;
        c.info[6] = (Real)h.stop;
        c.info[7] = (Real)h.nfev;
        c.info[8] = (Real)h.njev;
        c.info[9] = (Real)h.nlss;
        if (c.want_covar) lu_covar<M>(h.jtj, c.covar, h.p_e2, c.n);
        c.ret = (h.stop != 4 && h.stop != 7) ? h.k : kLmError;
        clear_req(h, req);
        ph = R_DONE;
        h.phase = ph;
        return;
      } LM_PHASE_END
    }
  }
};

}  // namespace brdf
