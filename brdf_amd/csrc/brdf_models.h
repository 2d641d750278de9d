// brdf_models.h -- the BRDF models as per-sample device functions.
//
// Replaces the reference's BRDFFunc callback (brdfdata.cpp:969-989), which levmar invokes once per
// evaluation over all n samples.  Here a model is split into the pieces a fused pass needs:
//
//     f(p; sample) = combine( lin(p), c0, shape( nl(p), sample ) )
//
//   lin(p)   two scalars that enter f linearly           (uniform per pass)
//   nl(p)    scalars feeding the per-sample transcendental (uniform per pass; depends on p[2] only)
//   shape    the one transcendental per sample: pow (Phong, Blinn-Phong) / exp (Ward)
//
// A forward-difference Jacobian needs f at p and at p+d_j e_j (misc_core.c:153-171).  Because lin()
// depends on p[0],p[1] (and p[2] for Phong) while nl() depends on p[2] only, the four evaluations
// share two shape() calls; every value is still bit-identical to evaluating f four times, since each
// evaluation performs the same operations on the same operands in the same order.
//
// Plane layout (struct extraData, brdfdata.cpp:962-966): c0 = cos(L.N), c1 = cos(N.H),
// c2 = cos(R.V) [Phong] or cos(N.V) [Ward].
//
// Model ids: 0 Phong, 1 Blinn-Phong (reference, brdfdata.h:44); 2 Ward (build-defined extension,
// SURVEY.md section 8 row a3 -- not present in the reference).
#pragma once

#include <math.h>

#include "lm_machine.h"

namespace brdf {

constexpr double kPi = 3.1415926535897932384626433832795;  // the reference's CV_PI literal

enum ModelId : int { MODEL_PHONG = 0, MODEL_BLINN_PHONG = 1, MODEL_WARD = 2, MODEL_COUNT = 3 };

struct Lin {
  double a, b;
};
struct Nl {
  double u0, u1;
};
// Per-sample invariants of the transcendental ("prepared sample").  In the FAST path they are computed
// once per fit by the first pass and kept in two HBM planes, so that every later evaluation costs one
// exp per sample instead of a pow (= log + exp) or an exp + two divisions + a square root:
//   Phong / Blinn-Phong   q1 = log(cos)          pow(c, n) == exp(n * log c)
//   Ward                  q1 = tan^2(theta_h) = (1 - c1^2)/c1^2 ,  q2 = 1/sqrt(c0*c2)
// In the EXACT path (FAST = false) q1,q2 are the raw cosines and shape() evaluates the reference's own
// expression (libm-style pow).  Ward's two paths perform identical operations (the invariants are the
// same sub-expressions, cached or not) and are bit-identical; for Phong/Blinn-Phong exp(n*log c) differs
// from pow(c,n) by at most |n log c| * 2^-53 relative on a term that is e^{n log c} small, i.e. below
// one ulp of the model value -- but it needs c > 0, which domain_ok() checks (the host driver re-runs a
// fit on the exact path if any used cosine is <= 0).
struct Prep {
  double q1, q2;
};

template <int MODEL>
struct BrdfModel;

// Phong: x = p0*c0 + ((p2+2)/2*PI)*p1*pow(c2,p2)          brdfdata.cpp:981 (note: *PI, not /(2PI))
template <>
struct BrdfModel<MODEL_PHONG> {
  static constexpr bool uses_c1 = false, uses_c2 = true;
  static constexpr int prep_planes = 1;
  static LM_HD Lin lin(const double *p) { return Lin{p[0], ((p[2] + 2.0) / 2.0 * kPi) * p[1]}; }
  static LM_HD Nl nl(const double *p) { return Nl{p[2], 0.0}; }
  template <bool FAST>
  static LM_HD Prep prepare(double, double, double c2) { return Prep{FAST ? log(c2) : c2, 0.0}; }
  static LM_HD bool domain_ok(double, double, double c2) { return c2 > 0.0; }
  template <bool FAST>
  static LM_HD double shape(const Nl &u, double, const Prep &q) { return FAST ? exp(u.u0 * q.q1) : pow(q.q1, u.u0); }
  static LM_HD double combine(const Lin &l, double c0, double s) { return l.a * c0 + l.b * s; }
  // analytic Jacobian row (not in the reference): with A = (p2+2)/2*PI, s = pow(c2, p2):
  //   df/dp0 = c0,  df/dp1 = A s,  df/dp2 = (1/2*PI p1) s + ((A p1) s) log c2
  static LM_HD void an_scalars(const double *p, double *an) {
    an[0] = (p[2] + 2.0) / 2.0 * kPi;
    an[1] = (1.0 / 2.0 * kPi) * p[1];
  }
  template <bool FAST>
  static LM_HD void an_row(const double *an, const Lin &l, const Nl &, double c0, const Prep &q, double s, double *j) {
    const double lc = FAST ? q.q1 : log(q.q1);
    j[0] = c0;
    j[1] = an[0] * s;
    j[2] = an[1] * s + (l.b * s) * lc;
  }
};

// Blinn-Phong: x = p0*c0 + p1*pow(c1,p2)                    brdfdata.cpp:986
template <>
struct BrdfModel<MODEL_BLINN_PHONG> {
  static constexpr bool uses_c1 = true, uses_c2 = false;
  static constexpr int prep_planes = 1;
  static LM_HD Lin lin(const double *p) { return Lin{p[0], p[1]}; }
  static LM_HD Nl nl(const double *p) { return Nl{p[2], 0.0}; }
  template <bool FAST>
  static LM_HD Prep prepare(double, double c1, double) { return Prep{FAST ? log(c1) : c1, 0.0}; }
  static LM_HD bool domain_ok(double, double c1, double) { return c1 > 0.0; }
  template <bool FAST>
  static LM_HD double shape(const Nl &u, double, const Prep &q) { return FAST ? exp(u.u0 * q.q1) : pow(q.q1, u.u0); }
  static LM_HD double combine(const Lin &l, double c0, double s) { return l.a * c0 + l.b * s; }
  // analytic Jacobian row: s = pow(c1, p2):  df/dp0 = c0,  df/dp1 = s,  df/dp2 = (p1 s) log c1
  static LM_HD void an_scalars(const double *p, double *an) {
    an[0] = p[1];
    an[1] = 0.0;
  }
  template <bool FAST>
  static LM_HD void an_row(const double *an, const Lin &, const Nl &, double c0, const Prep &q, double s, double *j) {
    const double lc = FAST ? q.q1 : log(q.q1);
    j[0] = c0;
    j[1] = s;
    j[2] = (an[0] * s) * lc;
  }
};

// exp(x) for x <= 0 (Ward's lobe: the argument is -tan^2/alpha^2).  On the device: ocml's own algorithm -- k = rint(x log2 e),
// r = x - k ln2 in two pieces, the same degree-11 polynomial with the same coefficients, ldexp -- hence the same bits, minus
// the overflow select at its end, which cannot fire for x <= 0 (the underflow select stays: for arguments like -1e40 -- a
// vanishing alpha -- the two-piece reduction is garbage and only the select makes the result 0): 3 of the ~85 instructions a
// trial sweep spends per sample.  A NaN stays a NaN, exp(-inf) = 0, every other value has ocml's bits.
LM_HD double exp_nonpos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double k = __builtin_rint(x * __longlong_as_double(0x3ff71547652b82feLL));
  double r = fma(k, __longlong_as_double((long long)0xbfe62e42fefa39efULL), x);
  r = fma(k, __longlong_as_double((long long)0xbc7abc9e3b39803fULL), r);
  double p = fma(r, __longlong_as_double(0x3e5ade156a5dcb37LL), __longlong_as_double(0x3e928af3fca7ab0cLL));
  p = fma(r, p, __longlong_as_double(0x3ec71dee623fde64LL));
  p = fma(r, p, __longlong_as_double(0x3efa01997c89e6b0LL));
  p = fma(r, p, __longlong_as_double(0x3f2a01a014761f6eLL));
  p = fma(r, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
  p = fma(r, p, __longlong_as_double(0x3f81111111122322LL));
  p = fma(r, p, __longlong_as_double(0x3fa55555555502a1LL));
  p = fma(r, p, __longlong_as_double(0x3fc5555555555511LL));
  p = fma(r, p, __longlong_as_double(0x3fe000000000000bLL));
  p = fma(r, p, 1.0);
  p = fma(r, p, 1.0);
  const double z = __builtin_ldexp(p, (int)k);
  return x < __longlong_as_double((long long)0xc090cc0000000000ULL) ? 0.0 : z;  // (-1075: ocml's own underflow select; keeps exp(-inf) = 0)
#else
  return exp(x);
#endif
}

// Ward (isotropic, build-defined): with a2 = p2^2, t2 = tan^2(theta_h), rinv = 1/sqrt(c0 c2)
//   x = c0 * ( p0/PI + p1 * ( (1/(4 PI a2)) * exp(-(t2 * (1/a2))) ) * rinv )
// written with reciprocals so that the per-sample invariants t2 and rinv can be cached.
template <>
struct BrdfModel<MODEL_WARD> {
  static constexpr bool uses_c1 = true, uses_c2 = true;
  static constexpr int prep_planes = 2;
  static LM_HD Lin lin(const double *p) { return Lin{p[0] / kPi, p[1]}; }
  static LM_HD Nl nl(const double *p) {
    const double a2 = p[2] * p[2];
    return Nl{1.0 / a2, 1.0 / (4.0 * kPi * a2)};
  }
  static LM_HD Prep ward_invariants(double c0, double c1, double c2) {
    const double ch2 = c1 * c1;
    return Prep{(1.0 - ch2) / ch2, 1.0 / sqrt(c0 * c2)};
  }
  template <bool FAST>
  static LM_HD Prep prepare(double c0, double c1, double c2) { return FAST ? ward_invariants(c0, c1, c2) : Prep{c1, c2}; }
  static LM_HD bool domain_ok(double, double, double) { return true; }
  template <bool FAST>
  static LM_HD double shape(const Nl &u, double c0, const Prep &q) {
    const Prep v = FAST ? q : ward_invariants(c0, q.q1, q.q2);
    const double g = exp_nonpos(-(v.q1 * u.u0));
    return (u.u1 * g) * v.q2;
  }
  static LM_HD double combine(const Lin &l, double c0, double s) { return c0 * (l.a + l.b * s); }
  // analytic Jacobian row: with S = (k g) rinv (what shape() returns), k = 1/(4 PI a2), g = exp(-t2/a2), a2 = p2^2:
  //   df/dp0 = c0/PI,  df/dp1 = c0 S,  df/dp2 = c0 ((p1 S) ((2/p2) (t2/a2 - 1)))        (dk/dp2 = -2k/p2, dg/dp2 = g t2 2/p2^3)
  static LM_HD void an_scalars(const double *p, double *an) {
    an[0] = p[1];
    an[1] = 2.0 / p[2];
  }
  template <bool FAST>
  static LM_HD void an_row(const double *an, const Lin &, const Nl &u, double c0, const Prep &q, double s, double *j) {
    const double t2 = FAST ? q.q1 : ward_invariants(c0, q.q1, q.q2).q1;
    j[0] = c0 / kPi;
    j[1] = c0 * s;
    j[2] = c0 * ((an[0] * s) * (an[1] * (t2 * u.u0 - 1.0)));
  }
};

// ---- per-pass uniforms derived from a Request<3> ---------------------------------------------------
constexpr int kM = 3;  // {kd, ks, n} / {rho_d, rho_s, alpha}: brdfdata.cpp:1046, :1107

template <int MODEL>
struct PassUniforms {
  using Mdl = BrdfModel<MODEL>;
  Lin l0;           // at p
  Nl n0;            // at p
  Lin lp[kM];       // at p + d_j e_j
  Lin lm[kM];       // at p - d_j e_j           (central differences)
  Nl np2, nm2;      // at p[2] + d_2, p[2] - d_2
  double dinv[kM];  // 1/d_j (forward) or 0.5/d_j (central): misc_core.c:167, :206
  Lin lq;           // at q                       (dif trial)
  Nl nq;
  Lin lk[kMaxCand];  // at the candidate points     (RQ_EVAL_MULTI)
  Nl nk[kMaxCand];
  double dp[kM], dp_l2, scal;
  double an[2];      // scalars of the analytic Jacobian (Mdl::an_scalars), RQ_JAC with analytic = 1
  int central, ncand, analytic;

  // only what the request kind reads is computed (each lin()/nl() hides an fp64 division or two, and on the
  // device this runs on a single lane between two passes)
  // need_base = false: the caller keeps f(p) per sample (resident regime), so a dif trial pass evaluates f at q only
  LM_HD void build(const Request<kM> &r, bool need_base = true, bool analytic_jac = false) {
    ncand = 0;
    analytic = analytic_jac ? 1 : 0;
    if (r.kind == RQ_EVAL_MULTI) {
      ncand = r.nk;
      for (int j = 0; j < kMaxCand; ++j)
        if (j < r.nk) {
          lk[j] = Mdl::lin(r.pk[j]);
          nk[j] = Mdl::nl(r.pk[j]);
        }
      return;
    }
    if (r.kind != RQ_DIF_UPDATE && (need_base || r.kind != RQ_DIF_TRIAL)) {
      l0 = Mdl::lin(r.p);
      n0 = Mdl::nl(r.p);
    }
    central = r.central;
    scal = r.scal;
    dp_l2 = r.dp_l2;
    for (int j = 0; j < kM; ++j) dp[j] = r.dp[j];
    if (r.kind == RQ_JAC && analytic_jac) {
      Mdl::an_scalars(r.p, an);
    } else if (r.kind == RQ_JAC || r.kind == RQ_DIF_JAC) {
      for (int j = 0; j < kM; ++j) {
        double pp[kM] = {r.p[0], r.p[1], r.p[2]};
        pp[j] = r.p[j] + r.d[j];  // "p[j]+=d", misc_core.c:161 / "tmp+d", :202
        lp[j] = Mdl::lin(pp);
        if (j == kM - 1) np2 = Mdl::nl(pp);
        double pm[kM] = {r.p[0], r.p[1], r.p[2]};
        pm[j] = r.p[j] - r.d[j];  // "p[j]-=d", misc_core.c:199
        lm[j] = Mdl::lin(pm);
        if (j == kM - 1) nm2 = Mdl::nl(pm);
        dinv[j] = (r.central ? 0.5 : 1.0) / r.d[j];
      }
    }
    if (r.kind == RQ_DIF_TRIAL) {  // (RQ_DIF_UPDATE needs dp, dp_l2 only)
      lq = Mdl::lin(r.q);
      nq = Mdl::nl(r.q);
    }
  }
};

// f(p) for one (prepared) sample
// (U: PassUniforms<MODEL>, or any struct with the fields a function reads -- channels_fit_impl.h keeps per-request copies in scalar registers)
template <int MODEL, bool FAST, class U>
LM_HD double model_value(const U &u, double c0, const Prep &q) {
  using Mdl = BrdfModel<MODEL>;
  return Mdl::combine(u.l0, c0, Mdl::template shape<FAST>(u.n0, c0, q));
}

// f(pk[j]) for one sample (candidate j of a multi-candidate evaluation)
template <int MODEL, bool FAST>
LM_HD double model_value_k(const PassUniforms<MODEL> &u, int j, double c0, const Prep &q) {
  using Mdl = BrdfModel<MODEL>;
  return Mdl::combine(u.lk[j], c0, Mdl::template shape<FAST>(u.nk[j], c0, q));
}

// f(q) for one sample (dif trial point)
template <int MODEL, bool FAST, class U>
LM_HD double model_value_q(const U &u, double c0, const Prep &q) {
  using Mdl = BrdfModel<MODEL>;
  return Mdl::combine(u.lq, c0, Mdl::template shape<FAST>(u.nq, c0, q));
}

// f(p) and one row of the finite-difference Jacobian.  `base` is the value subtracted in the forward
// formula: f(p) recomputed (bc_dif, lmbc_core.c:1049) or the stored hx (dif, lm_core.c:580).
// CENTRAL as a template argument: no branch inside the row, so that a sweep that hoists the (request-uniform) choice out of its
// sample loop gets straight-line sample bodies -- whose exp chains the scheduler can then interleave
template <int MODEL, bool FAST, bool CENTRAL, class U>
LM_HD void model_fd_row_t(const U &u, double c0, const Prep &q, bool need_f0, double &f0, double base_or_nan, bool use_base,
                          double *jrow) {
  using Mdl = BrdfModel<MODEL>;
  const double s0 = Mdl::template shape<FAST>(u.n0, c0, q);
  if (need_f0) f0 = Mdl::combine(u.l0, c0, s0);
  const double sp = Mdl::template shape<FAST>(u.np2, c0, q);
  if (!CENTRAL) {
    const double base = use_base ? base_or_nan : f0;
    jrow[0] = (Mdl::combine(u.lp[0], c0, s0) - base) * u.dinv[0];
    jrow[1] = (Mdl::combine(u.lp[1], c0, s0) - base) * u.dinv[1];
    jrow[2] = (Mdl::combine(u.lp[2], c0, sp) - base) * u.dinv[2];
  } else {
    const double sm = Mdl::template shape<FAST>(u.nm2, c0, q);
    jrow[0] = (Mdl::combine(u.lp[0], c0, s0) - Mdl::combine(u.lm[0], c0, s0)) * u.dinv[0];
    jrow[1] = (Mdl::combine(u.lp[1], c0, s0) - Mdl::combine(u.lm[1], c0, s0)) * u.dinv[1];
    jrow[2] = (Mdl::combine(u.lp[2], c0, sp) - Mdl::combine(u.lm[2], c0, sm)) * u.dinv[2];
  }
}
template <int MODEL, bool FAST, class U>
LM_HD void model_fd_row(const U &u, double c0, const Prep &q, bool need_f0, double &f0,
                        double base_or_nan, bool use_base, double *jrow) {
  if (!u.central)
    model_fd_row_t<MODEL, FAST, false>(u, c0, q, need_f0, f0, base_or_nan, use_base, jrow);
  else
    model_fd_row_t<MODEL, FAST, true>(u, c0, q, need_f0, f0, base_or_nan, use_base, jrow);
}

// f(p) and one row of the analytic Jacobian (the caller's jacf of dlevmar_bc_der, lmbc_core.c:578, for a built-in model)
template <int MODEL, bool FAST, class U>
LM_HD void model_an_row(const U &u, double c0, const Prep &q, double &f0, double *jrow) {
  using Mdl = BrdfModel<MODEL>;
  const double s = Mdl::template shape<FAST>(u.n0, c0, q);
  f0 = Mdl::combine(u.l0, c0, s);
  Mdl::template an_row<FAST>(u.an, u.l0, u.n0, c0, q, s, jrow);
}

// Broyden rank-one update of one Jacobian row, lm_core.c:760-766
LM_HD void broyden_row(const double *jold, double wrk, double hx, const double *dp, double dp_l2, double *jnew) {
  double t = 0.0;
  for (int l = 0; l < kM; ++l) t += jold[l] * dp[l];
  t = (wrk - hx - t) / dp_l2;
  for (int j = 0; j < kM; ++j) jnew[j] = jold[j] + t * dp[j];
}

}  // namespace brdf
