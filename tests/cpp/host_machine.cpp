// tests/cpp/host_machine.cpp -- TEST HARNESS (not product code).
//
// Drives the product's resumable LM machines (brdf_amd/csrc/lm_machine.h) and per-sample model math
// (brdf_amd/csrc/brdf_models.h) on the host, with a pass executor that sums in the REFERENCE's order
// (via liboracle's orc_l2_residual / orc_jtj_jte).  With identical summation order the machines must
// reproduce the oracle bit for bit -- iteration counts, nfev, p, info[] -- which validates the scalar
// logic of the GPU path without a GPU.  The GPU kernels reuse exactly these headers; only the
// (tree-shaped, deterministic) reduction order differs there.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../brdf_amd/csrc/brdf_models.h"
#include "../../oracle/oracle.h"

using namespace brdf;

namespace {

int g_speculative = 1;  // dif protocol under test (see lm_machine.h)
int g_multi = 1;        // bc: candidates per projected-gradient sweep
int g_dif_multi = 1;    // dif: trial points per sweep in a chain of rejections (DifMachine::Cold::multi)
int g_spec_jac = 0;     // bc: candidates evaluated by Jacobian passes (BcMachine::Cold::spec_jac)

template <int MODEL, bool FAST>
struct HostPasses {
  using Mdl = BrdfModel<MODEL>;
  // FAST = false: raw cosines + the reference's pow (bit-exact target).  FAST = true: the prepared-sample
  // path the GPU kernels use (cached log / tan^2 / rsqrt), here with host libm.
  Prep prep(int i) const { return Mdl::template prepare<FAST>(c0[i], c1[i], c2[i]); }
  const double *c0, *c1, *c2, *x;
  int n, bc_rule, speculative = 1;
  std::vector<double> e, e2, jac, hx[2], J[2];

  HostPasses(const double *angles, const double *x_, int n_, int bc)
      : c0(angles), c1(angles + n_), c2(angles + 2 * n_), x(x_), n(n_), bc_rule(bc), e(n_), e2(n_), jac(3 * n_) {
    for (int b = 0; b < 2; ++b) {
      hx[b].assign(n_, 0.0);
      J[b].assign(3 * (size_t)n_, 0.0);
    }
  }

  static void pack_lower(const double *jtj, double *s) {
    int c = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j <= i; ++j) s[c++] = jtj[i * 3 + j];
  }

  bool analytic = false;  // dlevmar_bc_der: RQ_JAC rows from the model's analytic Jacobian

  void run(const Request<3> &r, double *s, double &mx) {
    PassUniforms<MODEL> u;
    u.build(r, true, analytic);
    mx = 0.0;
    std::vector<double> f(n);
    double jtj[9], jte[3];
    switch (r.kind) {
    case RQ_EVAL:
    case RQ_SCALED: {
      for (int i = 0; i < n; ++i) f[i] = model_value<MODEL, FAST>(u, c0[i], prep(i));
      s[0] = orc_l2_residual(e.data(), x, f.data(), n);
      for (int i = 0; i < n; ++i) mx = fmax(mx, fabs(e[i]));
      if (r.kind == RQ_SCALED) {  // lmbc_core.c:163-166, descending
        double acc = 0.0;
        for (int i = n; i-- > 0;) {
          const double t = e[i] / r.scal;
          acc += t * t;
        }
        s[0] = acc;
      }
      break;
    }
    case RQ_EVAL_MULTI: {
      for (int j = 0; j < r.nk; ++j) {
        for (int i = 0; i < n; ++i) f[i] = model_value_k<MODEL, FAST>(u, j, c0[i], prep(i));
        s[j] = orc_l2_residual(e.data(), x, f.data(), n);
      }
      break;
    }
    case RQ_JAC: {
      for (int i = 0; i < n; ++i) {
        double f0 = 0.0;
        if (analytic)
          model_an_row<MODEL, FAST>(u, c0[i], prep(i), f0, &jac[3 * i]);
        else
          model_fd_row<MODEL, FAST>(u, c0[i], prep(i), true, f0, 0.0, false, &jac[3 * i]);
        f[i] = f0;
      }
      s[SumLayout<3>::NL + 3] = orc_l2_residual(e.data(), x, f.data(), n);
      orc_jtj_jte(jac.data(), e.data(), jtj, jte, n, 3, bc_rule);
      pack_lower(jtj, s);
      for (int i = 0; i < 3; ++i) s[SumLayout<3>::NL + i] = jte[i];
      break;
    }
    case RQ_DIF_INIT: {
      std::vector<double> &h = hx[r.sel_hx];
      for (int i = 0; i < n; ++i) h[i] = model_value<MODEL, FAST>(u, c0[i], prep(i));
      s[0] = orc_l2_residual(e.data(), x, h.data(), n);
      break;
    }
    case RQ_DIF_JAC: {
      std::vector<double> &h = hx[r.sel_hx];
      std::vector<double> &Jc = J[r.sel_j];
      for (int i = 0; i < n; ++i) {
        double f0 = 0.0;
        model_fd_row<MODEL, FAST>(u, c0[i], prep(i), false, f0, h[i], true, &Jc[3 * i]);
        e[i] = x[i] - h[i];
      }
      orc_jtj_jte(Jc.data(), e.data(), jtj, jte, n, 3, 0);
      pack_lower(jtj, s);
      for (int i = 0; i < 3; ++i) s[SumLayout<3>::NL + i] = jte[i];
      break;
    }
    case RQ_DIF_TRIAL: {
      if (!speculative) {  // two-step protocol: wrk only
        for (int i = 0; i < n; ++i) hx[1][i] = model_value_q<MODEL, FAST>(u, c0[i], prep(i));
        s[0] = orc_l2_residual(e2.data(), x, hx[1].data(), n);
        break;
      }
      std::vector<double> &h = hx[r.sel_hx];
      std::vector<double> &hn = hx[r.sel_hx ^ 1];
      std::vector<double> &Jc = J[r.sel_j];
      std::vector<double> &Jn = J[r.sel_j ^ 1];
      for (int i = 0; i < n; ++i) {
        hn[i] = model_value_q<MODEL, FAST>(u, c0[i], prep(i));
        broyden_row(&Jc[3 * i], hn[i], h[i], u.dp, u.dp_l2, &Jn[3 * i]);
        e[i] = x[i] - h[i];
      }
      s[0] = orc_l2_residual(e2.data(), x, hn.data(), n);
      orc_jtj_jte(Jn.data(), e2.data(), jtj, jte, n, 3, 0);
      pack_lower(jtj, s + 1);
      for (int i = 0; i < 3; ++i) s[1 + SumLayout<3>::NL + i] = jte[i];
      orc_jtj_jte(Jn.data(), e.data(), jtj, jte, n, 3, 0);
      for (int i = 0; i < 3; ++i) s[1 + SumLayout<3>::NL + 3 + i] = jte[i];
      break;
    }
    case RQ_DIF_UPDATE: {  // two-step protocol: hx[0] = f(p), hx[1] = wrk = f(q) of the preceding trial pass
      std::vector<double> &h = hx[0];
      std::vector<double> &w = hx[1];
      std::vector<double> &Jc = J[0];
      for (int i = 0; i < n; ++i) {
        double jn[3];
        broyden_row(&Jc[3 * i], w[i], h[i], u.dp, u.dp_l2, jn);
        for (int j = 0; j < 3; ++j) Jc[3 * i + j] = jn[j];
        e[i] = x[i] - (r.aux ? w[i] : h[i]);
        if (r.aux) h[i] = w[i];
      }
      orc_jtj_jte(Jc.data(), e.data(), jtj, jte, n, 3, 0);
      pack_lower(jtj, s);
      for (int i = 0; i < 3; ++i) s[SumLayout<3>::NL + i] = jte[i];
      break;
    }
    default:
      break;
    }
  }
};

template <int MODEL, bool FAST>
int fit(int method, double *angles, double *x, int n, double *p, int itmax, double *opts, double *lb,
        double *ub, double *dscl, double *info, double *covar, int *passes) {
  double s[SumLayout<3>::MAX] = {0};
  double mx = 0.0;
  int np = 0;
  if (method == 0) {
    HostPasses<MODEL, FAST> hp(angles, x, n, 0);
    hp.speculative = g_speculative;
    DifMachine<3> m;
    m.start(p, n, itmax, opts, covar != nullptr, g_speculative, g_dif_multi);
    while (m.h.req.kind != RQ_DONE) {
      hp.run(m.h.req, s, mx);
      ++np;
      m.template step<false, true>(s, mx);
    }
    for (int i = 0; i < 3; ++i) p[i] = m.h.p[i];
    if (info) for (int i = 0; i < 10; ++i) info[i] = m.c.info[i];
    if (covar) for (int i = 0; i < 9; ++i) covar[i] = m.c.covar[i];
    if (passes) *passes = np;
    return m.c.ret;
  }
  if (method == 3) {  // dlevmar_der with the analytic Jacobian
    HostPasses<MODEL, FAST> hp(angles, x, n, 1);  // J^T J blocking rule of lm_core.c:221 (n*m < 1024), as in lmbc_core.c
    hp.analytic = true;
    DerMachine<3> m;
    m.start(p, n, itmax, opts, covar != nullptr);
    while (m.h.req.kind != RQ_DONE) {
      hp.run(m.h.req, s, mx);
      ++np;
      m.step(s, mx);
    }
    for (int i = 0; i < 3; ++i) p[i] = m.h.p[i];
    if (info) for (int i = 0; i < 10; ++i) info[i] = m.c.info[i];
    if (covar) for (int i = 0; i < 9; ++i) covar[i] = m.c.covar[i];
    if (passes) *passes = np;
    return m.c.ret;
  }
  HostPasses<MODEL, FAST> hp(angles, x, n, 1);
  hp.analytic = (method == 2);
  BcMachine<3> m;
  m.start(p, n, lb, ub, dscl, itmax, opts, covar != nullptr, g_multi, g_spec_jac);
  m.c.analytic_jac = (method == 2) ? 1 : 0;
  while (m.h.req.kind != RQ_DONE) {
    hp.run(m.h.req, s, mx);
    ++np;
    m.template step<false, true, false, true>(s, mx);
  }
  for (int i = 0; i < 3; ++i) p[i] = m.h.p[i];
  if (info) for (int i = 0; i < 10; ++i) info[i] = m.c.info[i];
  if (covar) for (int i = 0; i < 9; ++i) covar[i] = m.c.covar[i];
  if (passes) *passes = np;
  return m.c.ret;
}

}  // namespace

extern "C" int hm_brdf_fit(int method, int model, double *angles, double *x, int n, double *p, int itmax,
                           double *opts, double *lb, double *ub, double *dscl, double *info,
                           double *covar, int *passes) {
  switch (model) {
  case 0: return fit<0, false>(method, angles, x, n, p, itmax, opts, lb, ub, dscl, info, covar, passes);
  case 1: return fit<1, false>(method, angles, x, n, p, itmax, opts, lb, ub, dscl, info, covar, passes);
  case 2: return fit<2, false>(method, angles, x, n, p, itmax, opts, lb, ub, dscl, info, covar, passes);
  }
  return -1;
}

// the same with the prepared-sample (FAST) model path
extern "C" int hm_brdf_fit_fast(int method, int model, double *angles, double *x, int n, double *p, int itmax,
                                double *opts, double *lb, double *ub, double *dscl, double *info,
                                double *covar, int *passes) {
  switch (model) {
  case 0: return fit<0, true>(method, angles, x, n, p, itmax, opts, lb, ub, dscl, info, covar, passes);
  case 1: return fit<1, true>(method, angles, x, n, p, itmax, opts, lb, ub, dscl, info, covar, passes);
  case 2: return fit<2, true>(method, angles, x, n, p, itmax, opts, lb, ub, dscl, info, covar, passes);
  }
  return -1;
}

extern "C" void hm_set_dif_protocol(int speculative) { g_speculative = speculative; }
extern "C" void hm_set_bc_multi(int k) { g_multi = k; }
extern "C" void hm_set_dif_multi(int k) { g_dif_multi = k; }
extern "C" void hm_set_bc_spec_jac(int on) { g_spec_jac = on; }
