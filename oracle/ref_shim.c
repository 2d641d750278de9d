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

int ref_brdf_fit(int method, int model, double *angles, double *x, int n, double *p, int itmax,
                 double *opts, double *lb, double *ub, double *info)
{
  struct orc_extra_data d;
  d.angles = angles;
  d.modelInfo = model;
  if (method == 0)
    return dlevmar_dif(orc_brdf_func, p, x, 3, n, itmax, opts, info, 0, 0, &d);
  if (method == 3) /* the reference's dlevmar_der driven by our analytic Jacobian */
    return dlevmar_der(orc_brdf_func, orc_brdf_jac, p, x, 3, n, itmax, opts, info, 0, 0, &d);
  if (method == 2) /* the reference's dlevmar_bc_der driven by our analytic Jacobian */
    return dlevmar_bc_der(orc_brdf_func, orc_brdf_jac, p, x, 3, n, lb, ub, 0, itmax, opts, info, 0, 0, &d);
  return dlevmar_bc_dif(orc_brdf_func, p, x, 3, n, lb, ub, 0, itmax, opts, info, 0, 0, &d);
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
  dlevmar_chkjac(orc_brdf_func, orc_brdf_jac, p, 3, n, &d, err);
}
