/*
 * oracle/brdf_models_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the BRDF model callback the reference hands to levmar
 * (brdfdata.cpp:969-989) plus the build-defined Ward model.
 *
 * Parity notes, kept deliberately:
 *  - Phong's specular factor is ((n+2)/2)*PI -- operator precedence at brdfdata.cpp:981 -- and NOT
 *    (n+2)/(2*PI) as the viewer uses (glutcallbacks.cpp:420).
 *  - A modelInfo other than 0/1/2 leaves hx untouched, as the reference does (no else branch).
 *  - PI is the reference's CV_PI literal.
 */
#include <math.h>
#include "oracle.h"

#define ORC_PI 3.1415926535897932384626433832795

void orc_brdf_func(double *p, double *hx, int m, int n, void *adata)
{
  const struct orc_extra_data *d = (const struct orc_extra_data *)adata;
  const double *c_ln = d->angles;         /* plane 0 */
  const double *c_nh = d->angles + n;     /* plane 1 */
  const double *c_p2 = d->angles + 2 * n; /* plane 2 */
  int i;
  (void)m;

  switch (d->modelInfo) {
  case 0: /* Phong, brdfdata.cpp:978-982 */
    for (i = 0; i < n; ++i)
      hx[i] = p[0] * c_ln[i] + ((p[2] + 2.0) / 2.0 * ORC_PI) * p[1] * (pow(c_p2[i], p[2]));
    break;
  case 1: /* Blinn-Phong, brdfdata.cpp:983-987 */
    for (i = 0; i < n; ++i)
      hx[i] = p[0] * c_ln[i] + p[1] * (pow(c_nh[i], p[2]));
    break;
  case 2: { /* Ward (isotropic, 1992), build-defined: p = (rho_d, rho_s, alpha).  Written with
             * reciprocals so the sample invariants tan^2(theta_h) and 1/sqrt(ci*co) are separable. */
    const double a2 = p[2] * p[2];
    const double ia2 = 1.0 / a2;
    const double k = 1.0 / (4.0 * ORC_PI * a2);
    const double dterm = p[0] / ORC_PI;
    for (i = 0; i < n; ++i) {
      const double ci = c_ln[i], ch = c_nh[i], co = c_p2[i];
      const double ch2 = ch * ch;
      const double t2 = (1.0 - ch2) / ch2; /* tan^2(theta_h) */
      const double rinv = 1.0 / sqrt(ci * co);
      const double g = exp(-(t2 * ia2));
      const double spec = (k * g) * rinv;
      hx[i] = ci * (dterm + p[1] * spec);
    }
    break;
  }
  default:
    break;
  }
}

/* Analytic Jacobian of the three models, n x 3 row-major like every levmar jacf (NOT in
 * the reference, which only ever differentiates numerically; checked with the reference's own dlevmar_chkjac in
 * tests/test_oracle_brdf.py).  Same sub-expressions, same order as orc_brdf_func. */
void orc_brdf_jac(double *p, double *jac, int m, int n, void *adata)
{
  const struct orc_extra_data *d = (const struct orc_extra_data *)adata;
  const double *c_ln = d->angles, *c_nh = d->angles + n, *c_p2 = d->angles + 2 * n;
  int i;
  (void)m;
  switch (d->modelInfo) {
  case 0: {
    const double A = (p[2] + 2.0) / 2.0 * ORC_PI, dA = (1.0 / 2.0 * ORC_PI) * p[1], b = A * p[1];
    for (i = 0; i < n; ++i) {
      const double s = pow(c_p2[i], p[2]), lc = log(c_p2[i]);
      jac[3 * i] = c_ln[i];
      jac[3 * i + 1] = A * s;
      jac[3 * i + 2] = dA * s + (b * s) * lc;
    }
    break;
  }
  case 1:
    for (i = 0; i < n; ++i) {
      const double s = pow(c_nh[i], p[2]), lc = log(c_nh[i]);
      jac[3 * i] = c_ln[i];
      jac[3 * i + 1] = s;
      jac[3 * i + 2] = (p[1] * s) * lc;
    }
    break;
  case 2: {
    const double a2 = p[2] * p[2], ia2 = 1.0 / a2, k = 1.0 / (4.0 * ORC_PI * a2), two_over = 2.0 / p[2];
    for (i = 0; i < n; ++i) {
      const double ci = c_ln[i], ch = c_nh[i], co = c_p2[i];
      const double ch2 = ch * ch;
      const double t2 = (1.0 - ch2) / ch2;
      const double rinv = 1.0 / sqrt(ci * co);
      const double g = exp(-(t2 * ia2));
      const double spec = (k * g) * rinv;
      jac[3 * i] = ci / ORC_PI;
      jac[3 * i + 1] = ci * spec;
      jac[3 * i + 2] = ci * ((p[1] * spec) * (two_over * (t2 * ia2 - 1.0)));
    }
    break;
  }
  default:
    break;
  }
}
