/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Linked into oracle/_ref/liblevmar_ref.so next to the reference's own levmar objects (compiled from
 * /root/reference/levmar/ in place; see Makefile).  It drives the REFERENCE solver
 * (levmar.h:112-127) with the BRDF model callback, reproducing the call sites
 * brdfdata.cpp:1058 / :1119, so tests and bench.py's cpu_baseline("reference") can time / compare the
 * reference CPU path itself.
 */
#include "levmar.h" /* the reference's header, via -I$(REF)/levmar */
#include "oracle.h"

/* The reference's OWN callback, compiled from its text (brdfdata.cpp:962-989 through ref_brdffunc.cpp; struct extraData has
 * the layout of orc_extra_data): Phong and Blinn-Phong fits of the reference are driven by it.  Ward (modelInfo 2) is not in
 * the reference: its callback is ours. */
void ref_BRDFFunc(double *p, double *x, int m, int n, void *data);
typedef void (*brdf_cb)(double *, double *, int, int, void *);
static brdf_cb callback_of(int model) { return (model == 0 || model == 1) ? ref_BRDFFunc : orc_brdf_func; }

int ref_brdf_fit(int method, int model, double *angles, double *x, int n, double *p, int itmax,
                 double *opts, double *lb, double *ub, double *info)
{
  struct orc_extra_data d;
  brdf_cb f = callback_of(model);
  d.angles = angles;
  d.modelInfo = model;
  if (method == 0)
    return dlevmar_dif(f, p, x, 3, n, itmax, opts, info, 0, 0, &d);
  if (method == 3) /* the reference's dlevmar_der driven by our analytic Jacobian */
    return dlevmar_der(f, orc_brdf_jac, p, x, 3, n, itmax, opts, info, 0, 0, &d);
  if (method == 2) /* the reference's dlevmar_bc_der driven by our analytic Jacobian */
    return dlevmar_bc_der(f, orc_brdf_jac, p, x, 3, n, lb, ub, 0, itmax, opts, info, 0, 0, &d);
  return dlevmar_bc_dif(f, p, x, 3, n, lb, ub, 0, itmax, opts, info, 0, 0, &d);
}

/* model values through the reference's own callback (models 0 / 1) */
void ref_brdf_values(int model, double *angles, int n, double *p, double *hx)
{
  struct orc_extra_data d;
  d.angles = angles;
  d.modelInfo = model;
  callback_of(model)(p, hx, 3, n, &d);
}

/* S independent fits, one after the other (the shape of CalcBRDFEquation's pixel loop,
 * brdfdata.cpp:1195-1220): angles[S][3][n], x[S][n], p[S][3] in/out, info[S][10]. */
int ref_brdf_fit_batch(int method, int model, double *angles, double *x, int S, int n, double *p,
                       int itmax, double *opts, double *lb, double *ub, double *info)
{
  int s, bad = 0;
  for (s = 0; s < S; ++s)
    if (ref_brdf_fit(method, model, angles + (long)s * 3 * n, x + (long)s * n, n, p + 3 * s, itmax, opts,
                     lb, ub, info + 10 * s) < 0)
      ++bad;
  return bad;
}

/* the reference's own dlevmar_chkjac (misc_core.c:250-321) on the BRDF callbacks: err[n] */
void ref_brdf_chkjac(int model, double *angles, int n, double *p, double *err)
{
  struct orc_extra_data d;
  d.angles = angles;
  d.modelInfo = model;
  dlevmar_chkjac(callback_of(model), orc_brdf_jac, p, 3, n, &d, err);
}

/* ---- single-precision test problems (ours: textbook NLLS functions written for float callbacks; the reference's
 * lmdemo.c problems are double-only).  They drive the REFERENCE's slevmar_* (tests/golden/gen_golden.py ->
 * tests/golden/slevmar_kat.json) and, by address, the product's slevmar_* on the GPU box. */
#include <math.h>
void sp_rosenbrock(float *p, float *x, int m, int n, void *d)
{
  (void)m; (void)n; (void)d;
  x[0] = 10.0f * (p[1] - p[0] * p[0]);
  x[1] = 1.0f - p[0];
}
void sp_rosenbrock_jac(float *p, float *j, int m, int n, void *d)
{
  (void)m; (void)n; (void)d;
  j[0] = -20.0f * p[0]; j[1] = 10.0f;
  j[2] = -1.0f;         j[3] = 0.0f;
}
void sp_wood(float *p, float *x, int m, int n, void *d)
{
  (void)m; (void)n; (void)d;
  x[0] = 10.0f * (p[1] - p[0] * p[0]);
  x[1] = 1.0f - p[0];
  x[2] = sqrtf(90.0f) * (p[3] - p[2] * p[2]);
  x[3] = 1.0f - p[2];
  x[4] = sqrtf(10.0f) * (p[1] + p[3] - 2.0f);
  x[5] = (p[1] - p[3]) / sqrtf(10.0f);
}
void sp_meyer(float *p, float *x, int m, int n, void *d)
{
  int i;
  (void)m; (void)d;
  for (i = 0; i < n; ++i) {
    const float u = 0.45f + 0.05f * (float)i;
    x[i] = p[0] * expf(10.0f * p[1] / (u + p[2]) - 13.0f);
  }
}
void sp_helval(float *p, float *x, int m, int n, void *d)
{
  const float pi = 3.14159265358979f;
  float theta;
  (void)m; (void)n; (void)d;
  if (p[0] < 0.0f)
    theta = atanf(p[1] / p[0]) / (2.0f * pi) + 0.5f;
  else if (p[0] > 0.0f)
    theta = atanf(p[1] / p[0]) / (2.0f * pi);
  else
    theta = (p[1] >= 0.0f) ? 0.25f : -0.25f;
  x[0] = 10.0f * (p[2] - 10.0f * theta);
  x[1] = 10.0f * (sqrtf(p[0] * p[0] + p[1] * p[1]) - 1.0f);
  x[2] = p[2];
}
void sp_helval_jac(float *p, float *j, int m, int n, void *d)
{
  const float pi = 3.14159265358979f;
  const float r2 = p[0] * p[0] + p[1] * p[1], r = sqrtf(r2);
  (void)m; (void)n; (void)d;
  j[0] = 50.0f * p[1] / (pi * r2); j[1] = -50.0f * p[0] / (pi * r2); j[2] = 10.0f;
  j[3] = 10.0f * p[0] / r;         j[4] = 10.0f * p[1] / r;          j[5] = 0.0f;
  j[6] = 0.0f;                     j[7] = 0.0f;                      j[8] = 1.0f;
}
void sp_hatfldb(float *p, float *x, int m, int n, void *d)
{
  int i;
  (void)n; (void)d;
  x[0] = p[0] - 1.0f;
  for (i = 1; i < m; ++i) x[i] = p[i - 1] - sqrtf(p[i]);
}
void sp_hatfldb_jac(float *p, float *j, int m, int n, void *d)
{
  int i, k;
  (void)d;
  for (i = 0; i < n * m; ++i) j[i] = 0.0f;
  j[0] = 1.0f;
  for (k = 1; k < m; ++k) {
    j[k * m + k - 1] = 1.0f;
    j[k * m + k] = -0.5f / sqrtf(p[k]);
  }
}

/* ---- a wide problem (ours): m up to 16 unknowns, for the host-callback path's m = 9..16 instantiations (levmar takes any m,
 * lm_core.c:528-548; lmdemo.c's problems stop at m = 5).  hx_i = sum_j p_j T_j(t_i) + 0.05 sin(p_0 t_i), t_i in [-1, 1],
 * T_j the Chebyshev polynomials; n measurements; nonlinear through the sine only, well conditioned through the basis. */
void wide_cheb(double *p, double *x, int m, int n, void *d)
{
  int i, j;
  (void)d;
  for (i = 0; i < n; ++i) {
    const double t = (n > 1) ? -1.0 + 2.0 * i / (n - 1) : 0.0;
    double tjm1 = 1.0, tj = t, s = p[0];
    for (j = 1; j < m; ++j) {
      const double next = 2.0 * t * tj - tjm1;
      s += p[j] * tj;
      tjm1 = tj;
      tj = next;
    }
    x[i] = s + 0.05 * sin(p[0] * t);
  }
}
void wide_cheb_jac(double *p, double *jac, int m, int n, void *d)
{
  int i, j;
  (void)d;
  for (i = 0; i < n; ++i) {
    const double t = (n > 1) ? -1.0 + 2.0 * i / (n - 1) : 0.0;
    double tjm1 = 1.0, tj = t;
    jac[i * m] = 1.0 + 0.05 * t * cos(p[0] * t);
    for (j = 1; j < m; ++j) {
      const double next = 2.0 * t * tj - tjm1;
      jac[i * m + j] = tj;
      tjm1 = tj;
      tj = next;
    }
  }
}
