/*
 * oracle/lm_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the double-precision levmar 2.6 routines on the BRDF fitting hot path, in
 * the no-LAPACK configuration the reference ships (levmar/levmar.h:31).  The floating-point operation
 * ORDER of every routine follows the cited reference lines so that, compiled with the same compiler
 * and without FP contraction, results are bit-identical to oracle/_ref -- that is what
 * tests/test_oracle_kat.py asserts.  Control flow and data structures are this build's own.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* constants: levmar/lm.c:35-36, lmbc.c:35-38, levmar.h:98-100 */
#define K_EPSILON 1E-12
#define K_ONE_THIRD 0.3333333334
#define K_LS_ITMAX 150
#define K_LS_POW 2.1
#define K_INIT_MU 1E-03
#define K_STOP_THRESH 1E-17
#define K_DIFF_DELTA 1E-06
#define K_BLOCK 32 /* misc.h:54 */

static double absd(double v) { return (v >= 0.0) ? v : -v; } /* misc.h FABS */

/* ------------------------------------------------------------------------------------------------
 * e = x - y, returns sum e^2.   misc_core.c:721-807: four interleaved accumulators, the 8-aligned
 * body walked from the top index down, the ragged tail walked upwards with a fixed accumulator map.
 * x == NULL means the zero vector.
 * ---------------------------------------------------------------------------------------------- */
double orc_l2_residual(double *e, const double *x, const double *y, int n)
{
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int body = (n >> 3) << 3;
  int top, k, t;

  for (top = body - 1; top > 0; top -= 8) {
    for (k = 0; k < 8; ++k) {
      const int j = top - k;
      e[j] = x ? x[j] - y[j] : -y[j];
      acc[k & 3] += e[j] * e[j];
    }
  }
  for (t = body; t < n; ++t) {
    const int left = n - t; /* 7..1 */
    e[t] = x ? x[t] - y[t] : -y[t];
    acc[(7 - left) & 3] += e[t] * e[t];
  }
  return acc[0] + acc[1] + acc[2] + acc[3];
}

/* forward differences, misc_core.c:137-172.  p is perturbed in place and restored. */
void orc_fdif_forward(orc_func_t f, double *p, const double *hx, double *hxx, double delta, double *jac,
                      int m, int n, void *adata)
{
  int i, j;
  for (j = 0; j < m; ++j) {
    double d = 1E-04 * p[j];
    const double keep = p[j];
    d = absd(d);
    if (d < delta) d = delta;
    p[j] += d;
    f(p, hxx, m, n, adata);
    p[j] = keep;
    d = 1.0 / d;
    for (i = 0; i < n; ++i) jac[i * m + j] = (hxx[i] - hx[i]) * d;
  }
}

/* central differences, misc_core.c:175-211 */
void orc_fdif_central(orc_func_t f, double *p, double *hxm, double *hxp, double delta, double *jac,
                      int m, int n, void *adata)
{
  int i, j;
  for (j = 0; j < m; ++j) {
    double d = 1E-04 * p[j];
    const double keep = p[j];
    d = absd(d);
    if (d < delta) d = delta;
    p[j] -= d;
    f(p, hxm, m, n, adata);
    p[j] = keep + d;
    f(p, hxp, m, n, adata);
    p[j] = keep;
    d = 0.5 / d;
    for (i = 0; i < n; ++i) jac[i * m + j] = (hxp[i] - hxm[i]) * d;
  }
}

/* b = a^T a for n x m row-major a; upper triangle by 32-row blocks, then mirrored.
 * misc_core.c:95-128 */
void orc_jtj_blocked(const double *a, double *b, int n, int m)
{
  int i, j, k, jj, kk;
  for (jj = 0; jj < m; jj += K_BLOCK) {
    const int jend = (jj + K_BLOCK <= m) ? jj + K_BLOCK : m;
    for (i = 0; i < m; ++i)
      for (j = (jj >= i) ? jj : i; j < jend; ++j) b[i * m + j] = 0.0;
    for (kk = 0; kk < n; kk += K_BLOCK) {
      const int kend = (kk + K_BLOCK <= n) ? kk + K_BLOCK : n;
      for (i = 0; i < m; ++i)
        for (j = (jj >= i) ? jj : i; j < jend; ++j) {
          double s = 0.0;
          for (k = kk; k < kend; ++k) s += a[k * m + i] * a[k * m + j];
          b[i * m + j] += s;
        }
    }
  }
  for (i = 0; i < m; ++i)
    for (j = 0; j < i; ++j) b[i * m + j] = b[j * m + i];
}

/* J^T J (lower triangle, then mirrored) and J^T e in one descending sweep: the "small problem"
 * path, lm_core.c:617-637 / lmbc_core.c:595-615 */
static void jtj_jte_small(const double *jac, const double *e, double *jtj, double *jte, int n, int m)
{
  int i, j, l;
  for (i = m * m; i-- > 0;) jtj[i] = 0.0;
  for (i = m; i-- > 0;) jte[i] = 0.0;
  for (l = n; l-- > 0;) {
    const double *row = jac + l * m;
    for (i = m; i-- > 0;) {
      const double alpha = row[i];
      for (j = i + 1; j-- > 0;) jtj[i * m + j] += row[j] * alpha;
      jte[i] += alpha * e[l];
    }
  }
  for (i = m; i-- > 0;)
    for (j = i + 1; j < m; ++j) jtj[i * m + j] = jtj[j * m + i];
}

/* the "large problem" path: blocked J^T J + ascending J^T e, lm_core.c:642-653 */
static void jtj_jte_large(const double *jac, const double *e, double *jtj, double *jte, int n, int m)
{
  int i, l;
  orc_jtj_blocked(jac, jtj, n, m);
  for (i = 0; i < m; ++i) jte[i] = 0.0;
  for (i = 0; i < n; ++i) {
    const double *row = jac + i * m;
    const double ei = e[i];
    for (l = 0; l < m; ++l) jte[l] += row[l] * ei;
  }
}

/* Crout LU with implicit row scaling + partial pivoting; factors into caller scratch.
 * Shared by the solver (Axb_core.c:1197-1247) and the inverse (misc_core.c:458-506).
 * Returns 0 when a row is entirely zero. */
static int lu_factor(double *a, int *perm, double *scale, int m)
{
  int i, j, k, pivot = -1;
  for (i = 0; i < m; ++i) {
    double big = 0.0;
    for (j = 0; j < m; ++j) {
      const double t = absd(a[i * m + j]);
      if (t > big) big = t;
    }
    if (big == 0.0) return 0;
    scale[i] = 1.0 / big;
  }
  for (j = 0; j < m; ++j) {
    double big = 0.0;
    for (i = 0; i < j; ++i) {
      double s = a[i * m + j];
      for (k = 0; k < i; ++k) s -= a[i * m + k] * a[k * m + j];
      a[i * m + j] = s;
    }
    for (i = j; i < m; ++i) {
      double s = a[i * m + j], t;
      for (k = 0; k < j; ++k) s -= a[i * m + k] * a[k * m + j];
      a[i * m + j] = s;
      if ((t = scale[i] * absd(s)) >= big) {
        big = t;
        pivot = i;
      }
    }
    if (j != pivot) {
      for (k = 0; k < m; ++k) {
        const double t = a[pivot * m + k];
        a[pivot * m + k] = a[j * m + k];
        a[j * m + k] = t;
      }
      scale[pivot] = scale[j];
    }
    perm[j] = pivot;
    if (a[j * m + j] == 0.0) a[j * m + j] = DBL_EPSILON;
    if (j != m - 1) {
      const double t = 1.0 / a[j * m + j];
      for (i = j + 1; i < m; ++i) a[i * m + j] *= t;
    }
  }
  return 1;
}

/* forward + back substitution on a factored system, Axb_core.c:1252-1270 */
static void lu_substitute(const double *a, const int *perm, double *x, int m)
{
  int i, j, first = 0;
  for (i = 0; i < m; ++i) {
    double s;
    j = perm[i];
    s = x[j];
    x[j] = x[i];
    if (first != 0) {
      for (j = first - 1; j < i; ++j) s -= a[i * m + j] * x[j];
    } else if (s != 0.0) {
      first = i + 1;
    }
    x[i] = s;
  }
  for (i = m - 1; i >= 0; --i) {
    double s = x[i];
    for (j = i + 1; j < m; ++j) s -= a[i * m + j] * x[j];
    x[i] = s / a[i * m + i];
  }
}

/* solve A x = B without touching A, B.  Axb_core.c:1140-1277 (reentrant here: no retained buffer) */
int orc_lu_solve(const double *A, const double *B, double *x, int m)
{
  double *a = (double *)malloc((size_t)(m * m + m) * sizeof(double));
  int *perm = (int *)malloc((size_t)m * sizeof(int));
  int ok;
  if (!a || !perm) {
    free(a);
    free(perm);
    return 0;
  }
  memcpy(a, A, (size_t)(m * m) * sizeof(double));
  memcpy(x, B, (size_t)m * sizeof(double));
  ok = lu_factor(a, perm, a + m * m, m);
  if (ok) lu_substitute(a, perm, x, m);
  free(a);
  free(perm);
  return ok;
}

/* C = sumsq/(n-m) * inverse(JtJ) by LU, misc_core.c:426-591 (no-LAPACK branch).  Returns rank m or 0. */
int orc_covar(const double *JtJ, double *C, double sumsq, int m, int n)
{
  double *a = (double *)malloc((size_t)(m * m + 2 * m) * sizeof(double));
  int *perm = (int *)malloc((size_t)m * sizeof(int));
  double *col, fact;
  int i, l, ok;
  if (!a || !perm) {
    free(a);
    free(perm);
    return 0;
  }
  col = a + m * m + m;
  memcpy(a, JtJ, (size_t)(m * m) * sizeof(double));
  ok = lu_factor(a, perm, a + m * m, m);
  if (!ok) {
    fprintf(stderr, "orc_covar(): singular matrix\n");
    free(a);
    free(perm);
    return 0;
  }
  for (l = 0; l < m; ++l) {
    for (i = 0; i < m; ++i) col[i] = 0.0;
    col[l] = 1.0;
    lu_substitute(a, perm, col, m);
    for (i = 0; i < m; ++i) C[i * m + l] = col[i];
  }
  free(a);
  free(perm);
  fact = sumsq / (double)(n - m);
  for (i = 0; i < m * m; ++i) C[i] *= fact;
  return m;
}

/* the reference's size switch between the two paths: n*m <= 1024 in dif (lm_core.c:594) but
 * n*m < 1024 in bc (lmbc_core.c:573).  Exported for the host-side state-machine harness in tests/. */
void orc_jtj_jte(const double *jac, const double *e, double *jtj, double *jte, int n, int m, int bc_rule)
{
  const int nm = n * m;
  const int small = bc_rule ? (nm < K_BLOCK * K_BLOCK) : (nm <= K_BLOCK * K_BLOCK);
  if (small)
    jtj_jte_small(jac, e, jtj, jte, n, m);
  else
    jtj_jte_large(jac, e, jtj, jte, n, m);
}

static double max_diag(const double *jtj, int m)
{
  double t = -DBL_MAX;
  int i;
  for (i = 0; i < m; ++i)
    if (t < jtj[i * m + i]) t = jtj[i * m + i];
  return t;
}

static void read_opts(const double *opts, double *tau, double *e1, double *e2, double *e2sq, double *e3)
{
  if (opts) {
    *tau = opts[0];
    *e1 = opts[1];
    *e2 = opts[2];
    *e2sq = opts[2] * opts[2];
    *e3 = opts[3];
  } else {
    *tau = K_INIT_MU;
    *e1 = K_STOP_THRESH;
    *e2 = K_STOP_THRESH;
    *e2sq = K_STOP_THRESH * K_STOP_THRESH;
    *e3 = K_STOP_THRESH;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Unconstrained LM with a secant (Broyden-updated) finite-difference Jacobian.  lm_core.c:438-842.
 * ---------------------------------------------------------------------------------------------- */
int orc_dlevmar_dif(orc_func_t f, double *p, double *x, int m, int n, int itmax, double *opts,
                    double *info, double *work, double *covar, void *adata)
{
  double *e, *hx, *jte, *jac, *jtj, *dp, *diag, *pdp, *wrk, *wrk2;
  double tau, eps1, eps2, eps2sq, eps3, delta;
  double mu = 0.0, tmp, p_e2, jte_inf = 0.0, pdp_e2, p_l2 = 0.0, dp_l2 = DBL_MAX, dF, dL, init_e2;
  int i, j, k, l, own_work = 0, forward = 1, solved;
  int nu, stop = 0, nfev, njap = 0, nlss = 0, updjac = 0, updp = 1, newjac = 0;
  const int refresh = (m >= 10) ? m : 10; /* lm_core.c:495 "K" */
  const int nm = n * m;

  if (n < m) {
    fprintf(stderr, "orc_dlevmar_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]\n", n, m);
    return ORC_ERROR;
  }
  read_opts(opts, &tau, &eps1, &eps2, &eps2sq, &eps3);
  delta = opts ? opts[4] : K_DIFF_DELTA;
  if (delta < 0.0) {
    delta = -delta;
    forward = 0;
  }

  if (!work) {
    work = (double *)malloc((size_t)(4 * n + 4 * m + n * m + m * m) * sizeof(double)); /* levmar.h:69 */
    if (!work) return ORC_ERROR;
    own_work = 1;
  }
  /* layout is observable through a caller-supplied work: lm_core.c:539-548 */
  e = work;
  hx = e + n;
  jte = hx + n;
  jac = jte + m;
  jtj = jac + nm;
  dp = jtj + m * m;
  diag = dp + m;
  pdp = diag + m;
  wrk = pdp + m;
  wrk2 = wrk + n;

  f(p, hx, m, n, adata);
  nfev = 1;
  p_e2 = orc_l2_residual(e, x, hx, n);
  init_e2 = p_e2;
  if (!isfinite(p_e2)) stop = 7;
  nu = 20; /* forces a Jacobian on the first pass */

  for (k = 0; k < itmax && !stop; ++k) {
    if (p_e2 <= eps3) {
      stop = 6;
      break;
    }
    if ((updp && nu > 16) || updjac == refresh) {
      if (forward) {
        orc_fdif_forward(f, p, hx, wrk, delta, jac, m, n, adata);
        ++njap;
        nfev += m;
      } else {
        orc_fdif_central(f, p, wrk, wrk2, delta, jac, m, n, adata);
        ++njap;
        nfev += 2 * m;
      }
      nu = 2;
      updjac = 0;
      updp = 0;
      newjac = 1;
    }
    if (newjac) {
      newjac = 0;
      if (nm <= K_BLOCK * K_BLOCK)
        jtj_jte_small(jac, e, jtj, jte, n, m);
      else
        jtj_jte_large(jac, e, jtj, jte, n, m);
      for (i = 0, p_l2 = jte_inf = 0.0; i < m; ++i) {
        if (jte_inf < (tmp = absd(jte[i]))) jte_inf = tmp;
        diag[i] = jtj[i * m + i];
        p_l2 += p[i] * p[i];
      }
    }
    if (jte_inf <= eps1) {
      dp_l2 = 0.0;
      stop = 1;
      break;
    }
    if (k == 0) {
      for (i = 0, tmp = -DBL_MAX; i < m; ++i)
        if (diag[i] > tmp) tmp = diag[i];
      mu = tau * tmp;
    }
    for (i = 0; i < m; ++i) jtj[i * m + i] += mu;
    solved = orc_lu_solve(jtj, jte, dp, m);
    ++nlss;
    if (solved) {
      for (i = 0, dp_l2 = 0.0; i < m; ++i) {
        pdp[i] = p[i] + (tmp = dp[i]);
        dp_l2 += tmp * tmp;
      }
      if (dp_l2 <= eps2sq * p_l2) {
        stop = 2;
        break;
      }
      if (dp_l2 >= (p_l2 + eps2) / (K_EPSILON * K_EPSILON)) {
        stop = 4;
        break;
      }
      f(pdp, wrk, m, n, adata);
      ++nfev;
      pdp_e2 = orc_l2_residual(wrk2, x, wrk, n);
      if (!isfinite(pdp_e2)) {
        stop = 7;
        break;
      }
      dF = p_e2 - pdp_e2;
      if (updp || dF > 0) { /* Broyden rank-one secant update, lm_core.c:759-769 */
        for (i = 0; i < n; ++i) {
          for (l = 0, tmp = 0.0; l < m; ++l) tmp += jac[i * m + l] * dp[l];
          tmp = (wrk[i] - hx[i] - tmp) / dp_l2;
          for (j = 0; j < m; ++j) jac[i * m + j] += tmp * dp[j];
        }
        ++updjac;
        newjac = 1;
      }
      for (i = 0, dL = 0.0; i < m; ++i) dL += dp[i] * (mu * dp[i] + jte[i]);
      if (dL > 0.0 && dF > 0.0) {
        tmp = (2.0 * dF / dL - 1.0);
        tmp = 1.0 - tmp * tmp * tmp;
        mu = mu * ((tmp >= K_ONE_THIRD) ? tmp : K_ONE_THIRD);
        nu = 2;
        for (i = 0; i < m; ++i) p[i] = pdp[i];
        for (i = 0; i < n; ++i) {
          e[i] = wrk2[i];
          hx[i] = wrk[i];
        }
        p_e2 = pdp_e2;
        updp = 1;
        continue;
      }
    }
    /* rejected step or unsolvable system */
    mu *= nu;
    {
      const int nu2 = nu << 1;
      if (nu2 <= nu) {
        stop = 5;
        break;
      }
      nu = nu2;
    }
    for (i = 0; i < m; ++i) jtj[i * m + i] = diag[i];
  }
  if (k >= itmax) stop = 3;
  for (i = 0; i < m; ++i) jtj[i * m + i] = diag[i];

  if (info) {
    info[0] = init_e2;
    info[1] = p_e2;
    info[2] = jte_inf;
    info[3] = dp_l2;
    info[4] = mu / max_diag(jtj, m);
    info[5] = (double)k;
    info[6] = (double)stop;
    info[7] = (double)nfev;
    info[8] = (double)njap;
    info[9] = (double)nlss;
  }
  if (covar) orc_covar(jtj, covar, p_e2, m, n);
  if (own_work) free(work);
  return (stop != 4 && stop != 7) ? k : ORC_ERROR;
}

/* ------------------------------------------------------------------------------------------------
 * Unconstrained LM with the caller's analytic Jacobian.  lm_core.c:64-432.
 * ---------------------------------------------------------------------------------------------- */
int orc_dlevmar_der(orc_func_t f, orc_jacf_t jf, double *p, double *x, int m, int n, int itmax, double *opts,
                    double *info, double *work, double *covar, void *adata)
{
  double *e, *hx, *jte, *jac, *jtj, *dp, *diag, *pdp;
  double tau, eps1, eps2, eps2sq, eps3;
  double mu = 0.0, tmp, p_e2, jte_inf = 0.0, pdp_e2, p_l2 = 0.0, dp_l2 = DBL_MAX, dF, dL, init_e2;
  int i, k, own_work = 0, solved, nu = 2, stop = 0, nfev, njev = 0, nlss = 0;
  const int nm = n * m;

  if (n < m) {
    fprintf(stderr, "orc_dlevmar_der(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]\n", n, m);
    return ORC_ERROR;
  }
  if (!jf) return ORC_ERROR;
  read_opts(opts, &tau, &eps1, &eps2, &eps2sq, &eps3);
  if (!work) {
    work = (double *)malloc((size_t)(2 * n + 4 * m + n * m + m * m) * sizeof(double)); /* levmar.h:68 */
    if (!work) return ORC_ERROR;
    own_work = 1;
  }
  e = work;
  hx = e + n;
  jte = hx + n;
  jac = jte + m;
  jtj = jac + nm;
  dp = jtj + m * m;
  diag = dp + m;
  pdp = diag + m;

  f(p, hx, m, n, adata);
  nfev = 1;
  p_e2 = orc_l2_residual(e, x, hx, n);
  init_e2 = p_e2;
  if (!isfinite(p_e2)) stop = 7;

  for (k = 0; k < itmax && !stop; ++k) {
    if (p_e2 <= eps3) {
      stop = 6;
      break;
    }
    jf(p, jac, m, n, adata);
    ++njev;
    if (nm < K_BLOCK * K_BLOCK)
      jtj_jte_small(jac, e, jtj, jte, n, m);
    else
      jtj_jte_large(jac, e, jtj, jte, n, m);
    for (i = 0, p_l2 = jte_inf = 0.0; i < m; ++i) {
      if (jte_inf < (tmp = absd(jte[i]))) jte_inf = tmp;
      diag[i] = jtj[i * m + i];
      p_l2 += p[i] * p[i];
    }
    if (jte_inf <= eps1) {
      dp_l2 = 0.0;
      stop = 1;
      break;
    }
    if (k == 0) {
      for (i = 0, tmp = -DBL_MAX; i < m; ++i)
        if (diag[i] > tmp) tmp = diag[i];
      mu = tau * tmp;
    }
    for (;;) { /* damped steps until one reduces the error */
      for (i = 0; i < m; ++i) jtj[i * m + i] += mu;
      solved = orc_lu_solve(jtj, jte, dp, m);
      ++nlss;
      if (solved) {
        for (i = 0, dp_l2 = 0.0; i < m; ++i) {
          pdp[i] = p[i] + (tmp = dp[i]);
          dp_l2 += tmp * tmp;
        }
        if (dp_l2 <= eps2sq * p_l2) {
          stop = 2;
          break;
        }
        if (dp_l2 >= (p_l2 + eps2) / (K_EPSILON * K_EPSILON)) {
          stop = 4;
          break;
        }
        f(pdp, hx, m, n, adata);
        ++nfev;
        pdp_e2 = orc_l2_residual(hx, x, hx, n);
        if (!isfinite(pdp_e2)) {
          stop = 7;
          break;
        }
        for (i = 0, dL = 0.0; i < m; ++i) dL += dp[i] * (mu * dp[i] + jte[i]);
        dF = p_e2 - pdp_e2;
        if (dL > 0.0 && dF > 0.0) {
          tmp = (2.0 * dF / dL - 1.0);
          tmp = 1.0 - tmp * tmp * tmp;
          mu = mu * ((tmp >= K_ONE_THIRD) ? tmp : K_ONE_THIRD);
          nu = 2;
          for (i = 0; i < m; ++i) p[i] = pdp[i];
          for (i = 0; i < n; ++i) e[i] = hx[i];
          p_e2 = pdp_e2;
          break;
        }
      }
      mu *= nu;
      {
        const int nu2 = nu << 1;
        if (nu2 <= nu) {
          stop = 5;
          break;
        }
        nu = nu2;
      }
      for (i = 0; i < m; ++i) jtj[i * m + i] = diag[i];
    }
  }
  if (k >= itmax) stop = 3;
  for (i = 0; i < m; ++i) jtj[i * m + i] = diag[i];
  if (info) {
    info[0] = init_e2;
    info[1] = p_e2;
    info[2] = jte_inf;
    info[3] = dp_l2;
    info[4] = mu / max_diag(jtj, m);
    info[5] = (double)k;
    info[6] = (double)stop;
    info[7] = (double)nfev;
    info[8] = (double)njev;
    info[9] = (double)nlss;
  }
  if (covar) orc_covar(jtj, covar, p_e2, m, n);
  if (own_work) free(work);
  return (stop != 4 && stop != 7) ? k : ORC_ERROR;
}

/* ------------------------------------------------------------------------------------------------
 * Box-constrained LM.  lmbc_core.c.
 * ---------------------------------------------------------------------------------------------- */
static double median3(double a, double b, double c) /* lmbc_core.c:59-61 */
{
  return (a >= b) ? ((c >= a) ? a : ((c <= b) ? b : c)) : ((c >= b) ? b : ((c <= a) ? a : c));
}

static void box_project(double *p, const double *lb, const double *ub, int m) /* lmbc_core.c:68-88 */
{
  int i;
  if (!lb && !ub) return;
  for (i = m; i-- > 0;) {
    if (lb && ub)
      p[i] = median3(lb[i], p[i], ub[i]);
    else if (ub) {
      if (p[i] > ub[i]) p[i] = ub[i];
    } else {
      if (p[i] < lb[i]) p[i] = lb[i];
    }
  }
}

static void box_scale(double *lb, double *ub, const double *scl, int m, int divide) /* :94-142 */
{
  int i;
  for (i = m; i-- > 0;) {
    if (ub && ub[i] != DBL_MAX) ub[i] = divide ? ub[i] / scl[i] : ub[i] * scl[i];
    if (lb && lb[i] != -DBL_MAX) lb[i] = divide ? lb[i] / scl[i] : lb[i] * scl[i];
  }
}

static double scaled_norm(const double *v, int n) /* lmbc_core.c:156-168 (Blue's method, no LAPACK) */
{
  double big = 0.0, s = 0.0;
  int i;
  for (i = n; i-- > 0;) {
    if (v[i] > big)
      big = v[i];
    else if (v[i] < -big)
      big = -v[i];
  }
  for (i = n; i-- > 0;) {
    const double t = v[i] / big;
    s += t * t;
  }
  return big * sqrt(s);
}

struct bc_eval { /* what the line search needs in order to evaluate the objective */
  orc_func_t f;
  int n, *nfev;
  double *hx, *x, *lb, *ub;
  void *adata;
};

/* Schnabel/Koontz/Weiss backtracking line search with box projection.  lmbc_core.c:179-337.
 * Returns the uncmin-style code (0 ok, 1 failed); *f_new receives ||e||^2 at xpls; ev->hx holds e. */
static int line_search(int m, const double *x0, double f0, const double *g, double *step, double alpha,
                       double *xpls, double *f_new, struct bc_eval *ev, double stepmx, double steptl,
                       const double *sx)
{
  int i, j, first_back = 1;
  double lambda, tlmbda = 0.0, rmnlmb, sln, slp, rln, t;
  double fpls, pfpls = 0.0, plmbda = 0.0;

  f0 *= 0.5;
  t = 0.0;
  for (i = m; i-- > 0;) t += step[i] * step[i];
  sln = sqrt(t);
  if (sln > stepmx) {
    const double scl = stepmx / sln;
    for (i = m; i-- > 0;) step[i] *= scl;
    sln = stepmx;
  }
  for (i = m, slp = rln = 0.0; i-- > 0;) {
    double denom, rel;
    slp += g[i] * step[i];
    denom = (absd(x0[i]) >= 1.0) ? absd(x0[i]) : 1.0;
    rel = absd(step[i]) / denom;
    if (rln < rel) rln = rel;
  }
  rmnlmb = steptl / rln;
  lambda = 1.0;

  for (j = K_LS_ITMAX; j-- > 0;) {
    for (i = m; i-- > 0;) xpls[i] = x0[i] + lambda * step[i];
    box_project(xpls, ev->lb, ev->ub, m);
    if (!sx) {
      ev->f(xpls, ev->hx, m, ev->n, ev->adata);
      ++(*ev->nfev);
    } else {
      for (i = m; i-- > 0;) xpls[i] *= sx[i];
      ev->f(xpls, ev->hx, m, ev->n, ev->adata);
      ++(*ev->nfev);
      for (i = m; i-- > 0;) xpls[i] /= sx[i];
    }
    t = orc_l2_residual(ev->hx, ev->x, ev->hx, ev->n);
    fpls = 0.5 * t;
    *f_new = t;

    if (fpls <= f0 + slp * alpha * lambda) return 0;
    if (lambda < rmnlmb) return 1;

    if (!isfinite(fpls)) {
      lambda *= 0.1;
      first_back = 1;
    } else {
      if (first_back) { /* quadratic model */
        tlmbda = -lambda * slp / ((fpls - f0 - slp) * 2.0);
        first_back = 0;
      } else { /* cubic model */
        const double t1 = fpls - f0 - lambda * slp;
        const double t2 = pfpls - f0 - plmbda * slp;
        const double t3 = 1.0 / (lambda - plmbda);
        const double a3 = 3.0 * t3 * (t1 / (lambda * lambda) - t2 / (plmbda * plmbda));
        const double b = t3 * (t2 * lambda / (plmbda * plmbda) - t1 * plmbda / (lambda * lambda));
        const double disc = b * b - a3 * slp;
        if (disc > b * b)
          tlmbda = (-b + ((a3 < 0) ? -sqrt(disc) : sqrt(disc))) / a3;
        else
          tlmbda = (-b + ((a3 < 0) ? sqrt(disc) : -sqrt(disc))) / a3;
        if (tlmbda > lambda * 0.5) tlmbda = lambda * 0.5;
      }
      plmbda = lambda;
      pfpls = fpls;
      if (tlmbda < lambda * 0.1)
        lambda *= 0.1;
      else
        lambda = tlmbda;
    }
  }
  return 1;
}

int orc_dlevmar_bc_der(orc_func_t f, orc_jacf_t jf, double *p, double *x, int m, int n, double *lb,
                       double *ub, double *dscl, int itmax, double *opts, double *info, double *work,
                       double *covar, void *adata)
{
  double *e, *hx, *jte, *jac, *jtj, *dp, *diag, *pdp, *sp = NULL;
  double tau, eps1, eps2, eps2sq, eps3;
  double mu = 0.0, tmp, p_e2, jte_inf = 0.0, pdp_e2 = 0.0, p_l2 = 0.0, dp_l2 = DBL_MAX, dF, dL, init_e2;
  const double alpha = 1e-4, beta = 0.9, gamma = 0.99995, rho = 1e-8, tming = 1e-18, tini = 1.0;
  double t = 0.0, t0, gdp;
  int i, j, k, own_work = 0, solved, nu = 2, stop = 0, nfev, njev = 0, nlss = 0;
  int gprev = 0, nactive;
  const int nm = n * m;
  struct bc_eval ev;

  if (n < m) {
    fprintf(stderr, "orc_dlevmar_bc_der(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]\n", n, m);
    return ORC_ERROR;
  }
  if (!jf) return ORC_ERROR;
  if (lb && ub)
    for (i = 0; i < m; ++i)
      if (lb[i] > ub[i]) {
        fprintf(stderr, "orc_dlevmar_bc_der(): at least one lower bound exceeds the upper one\n");
        return ORC_ERROR;
      }
  if (dscl) {
    for (i = m; i-- > 0;)
      if (dscl[i] <= 0.0) return ORC_ERROR;
    sp = (double *)malloc((size_t)m * sizeof(double));
    if (!sp) return ORC_ERROR;
  }
  read_opts(opts, &tau, &eps1, &eps2, &eps2sq, &eps3);

  if (!work) {
    work = (double *)malloc((size_t)(2 * n + 4 * m + n * m + m * m) * sizeof(double)); /* levmar.h:74 */
    if (!work) {
      free(sp);
      return ORC_ERROR;
    }
    own_work = 1;
  }
  e = work;
  hx = e + n;
  jte = hx + n;
  jac = jte + m;
  jtj = jac + nm;
  dp = jtj + m * m;
  diag = dp + m;
  pdp = diag + m;

  ev.f = f;
  ev.n = n;
  ev.hx = hx;
  ev.x = x;
  ev.lb = lb;
  ev.ub = ub;
  ev.adata = adata;
  ev.nfev = &nfev;

  for (i = 0; i < m; ++i) pdp[i] = p[i];
  box_project(p, lb, ub, m);
  for (i = 0; i < m; ++i)
    if (pdp[i] != p[i])
      fprintf(stderr, "Warning: component %d of starting point not feasible in orc_dlevmar_bc_der()! [%g projected to %g]\n",
              i, pdp[i], p[i]);

  f(p, hx, m, n, adata);
  nfev = 1;
  p_e2 = orc_l2_residual(e, x, hx, n);
  init_e2 = p_e2;
  if (!isfinite(p_e2)) stop = 7;

  if (dscl) {
    for (i = m; i-- > 0;) p[i] /= dscl[i];
    box_scale(lb, ub, dscl, m, 1);
  }

  for (k = 0; k < itmax && !stop; ++k) {
    if (p_e2 <= eps3) {
      stop = 6;
      break;
    }
    if (!dscl) {
      jf(p, jac, m, n, adata);
      ++njev;
    } else {
      for (i = m; i-- > 0;) sp[i] = p[i] * dscl[i];
      jf(sp, jac, m, n, adata);
      ++njev;
      for (i = n; i-- > 0;)
        for (j = m; j-- > 0;) jac[i * m + j] *= dscl[j];
    }
    if (nm < K_BLOCK * K_BLOCK) /* strict < here, <= in dif: lmbc_core.c:573 vs lm_core.c:594 */
      jtj_jte_small(jac, e, jtj, jte, n, m);
    else
      jtj_jte_large(jac, e, jtj, jte, n, m);

    /* gradient norm over free variables only, lmbc_core.c:639-646 */
    for (i = j = nactive = 0, p_l2 = jte_inf = 0.0; i < m; ++i) {
      if (ub && p[i] == ub[i]) {
        ++nactive;
        if (jte[i] > 0.0) ++j;
      } else if (lb && p[i] == lb[i]) {
        ++nactive;
        if (jte[i] < 0.0) ++j;
      } else if (jte_inf < (tmp = absd(jte[i])))
        jte_inf = tmp;
      diag[i] = jtj[i * m + i];
      p_l2 += p[i] * p[i];
    }
    if (j == nactive && (jte_inf <= eps1)) {
      dp_l2 = 0.0;
      stop = 1;
      break;
    }
    if (k == 0) {
      if (!lb && !ub) {
        for (i = 0, tmp = -DBL_MAX; i < m; ++i)
          if (diag[i] > tmp) tmp = diag[i];
        mu = tau * tmp;
      } else
        mu = 0.5 * tau * p_e2; /* Kanzow's starting mu */
    }

    for (;;) { /* step selection: LM step, else line search, else projected gradient */
      int take_gradproj = 0;
      for (i = 0; i < m; ++i) jtj[i * m + i] += mu;
      solved = orc_lu_solve(jtj, jte, dp, m);
      ++nlss;
      if (!solved) {
        const int nu2 = nu << 1;
        mu *= nu;
        if (nu2 <= nu) {
          stop = 5;
          break;
        }
        nu = nu2;
        for (i = 0; i < m; ++i) jtj[i * m + i] = diag[i];
        continue;
      }
      for (i = 0; i < m; ++i) pdp[i] = p[i] + dp[i];
      box_project(pdp, lb, ub, m);
      for (i = 0, dp_l2 = 0.0; i < m; ++i) {
        dp[i] = tmp = pdp[i] - p[i];
        dp_l2 += tmp * tmp;
      }
      if (dp_l2 <= eps2sq * p_l2) {
        stop = 2;
        break;
      }
      if (dp_l2 >= (p_l2 + eps2) / (K_EPSILON * K_EPSILON)) {
        stop = 4;
        break;
      }
      if (!dscl) {
        f(pdp, hx, m, n, adata);
        ++nfev;
      } else {
        for (i = m; i-- > 0;) sp[i] = pdp[i] * dscl[i];
        f(sp, hx, m, n, adata);
        ++nfev;
      }
      pdp_e2 = orc_l2_residual(hx, x, hx, n);
      if (!isfinite(pdp_e2) && !isfinite(scaled_norm(hx, n))) {
        stop = 7;
        break;
      }
      if (pdp_e2 <= gamma * p_e2) { /* LM step accepted */
        for (i = 0, dL = 0.0; i < m; ++i) dL += dp[i] * (mu * dp[i] + jte[i]);
        if (dL > 0.0) {
          dF = p_e2 - pdp_e2;
          tmp = (2.0 * dF / dL - 1.0);
          tmp = 1.0 - tmp * tmp * tmp;
          mu = mu * ((tmp >= K_ONE_THIRD) ? tmp : K_ONE_THIRD);
        } else {
          tmp = 0.1 * pdp_e2;
          mu = (mu >= tmp) ? tmp : mu;
        }
        nu = 2;
        for (i = 0; i < m; ++i) p[i] = pdp[i];
        for (i = 0; i < n; ++i) e[i] = hx[i];
        p_e2 = pdp_e2;
        gprev = 0;
        break;
      }

      /* LM step did not reduce the error enough: is it at least a descent direction? */
      for (i = 0, gdp = 0.0; i < m; ++i) {
        jte[i] = -jte[i];
        gdp += jte[i] * dp[i];
      }
      if (gdp <= -rho * pow(dp_l2, K_LS_POW / 2.0)) {
        const double steptl = 1e3 * sqrt(DBL_EPSILON);
        double stepmx;
        int rc;
        tmp = sqrt(p_l2);
        stepmx = 1e3 * ((tmp >= 1.0) ? tmp : 1.0);
        rc = line_search(m, p, p_e2, jte, dp, alpha, pdp, &pdp_e2, &ev, stepmx, steptl, dscl);
        if (rc != 0 || !isfinite(pdp_e2))
          take_gradproj = 1;
        else
          gprev = 0;
      } else
        take_gradproj = 1;

      if (take_gradproj) { /* projected gradient search, lmbc_core.c:871-946 */
        int found = 0;
        for (i = 0, tmp = 0.0; i < m; ++i) tmp += jte[i] * jte[i];
        tmp = sqrt(tmp);
        tmp = 100.0 / (1.0 + tmp);
        t0 = (tmp <= tini) ? tmp : tini;
        for (t = (gprev) ? t : t0; t > tming; t *= beta) {
          for (i = 0; i < m; ++i) pdp[i] = p[i] - t * jte[i];
          box_project(pdp, lb, ub, m);
          for (i = 0, dp_l2 = 0.0; i < m; ++i) {
            dp[i] = tmp = pdp[i] - p[i];
            dp_l2 += tmp * tmp;
          }
          if (!dscl) {
            f(pdp, hx, m, n, adata);
            ++nfev;
          } else {
            for (i = m; i-- > 0;) sp[i] = pdp[i] * dscl[i];
            f(sp, hx, m, n, adata);
            ++nfev;
          }
          pdp_e2 = orc_l2_residual(hx, x, hx, n);
          if (!isfinite(pdp_e2) && !isfinite(scaled_norm(hx, n))) {
            stop = 7;
            goto finish;
          }
          for (i = 0, gdp = 0.0; i < m; ++i) gdp += jte[i] * dp[i];
          if (gprev && pdp_e2 <= p_e2 + 2.0 * 0.99999 * gdp) { /* remembered t was too small */
            t = t0;
            gprev = 0;
            continue; /* NB: the loop increment still applies t*=beta, as in the reference */
          }
          if (pdp_e2 <= p_e2 + 2.0 * alpha * gdp) {
            found = 1;
            break;
          }
        }
        if (!found) {
          gprev = 0;
          break; /* search failed: leave the step loop with p unchanged */
        }
        gprev = 1;
      }

      /* commit the line-search / projected-gradient point */
      for (i = 0, dp_l2 = 0.0; i < m; ++i) {
        tmp = pdp[i] - p[i];
        dp_l2 += tmp * tmp;
      }
      if (dp_l2 <= eps2sq * p_l2) {
        stop = 2;
        break;
      }
      for (i = 0; i < m; ++i) p[i] = pdp[i];
      for (i = 0; i < n; ++i) e[i] = hx[i];
      p_e2 = pdp_e2;
      break;
    }
  }

finish:
  if (k >= itmax) stop = 3;
  for (i = 0; i < m; ++i) jtj[i * m + i] = diag[i];
  if (info) {
    info[0] = init_e2;
    info[1] = p_e2;
    info[2] = jte_inf;
    info[3] = dp_l2;
    info[4] = mu / max_diag(jtj, m);
    info[5] = (double)k;
    info[6] = (double)stop;
    info[7] = (double)nfev;
    info[8] = (double)njev;
    info[9] = (double)nlss;
  }
  if (covar) {
    orc_covar(jtj, covar, p_e2, m, n);
    if (dscl)
      for (i = m; i-- > 0;)
        for (j = m; j-- > 0;) covar[i * m + j] *= (dscl[i] * dscl[j]);
  }
  if (own_work) free(work);
  if (dscl) {
    for (i = 0; i < m; ++i) p[i] *= dscl[i];
    box_scale(lb, ub, dscl, m, 0);
    free(sp);
  }
  return (stop != 4 && stop != 7) ? k : ORC_ERROR;
}

/* finite-difference shim over bc_der, lmbc_core.c:1027-1129 */
struct fd_shim {
  int forward;
  orc_func_t f;
  double *hx, *hxx, delta;
  void *adata;
};
static void fd_shim_func(double *p, double *hx, int m, int n, void *data)
{
  struct fd_shim *s = (struct fd_shim *)data;
  s->f(p, hx, m, n, s->adata);
}
static void fd_shim_jac(double *p, double *jac, int m, int n, void *data)
{
  struct fd_shim *s = (struct fd_shim *)data;
  if (s->forward) {
    s->f(p, s->hx, m, n, s->adata); /* re-evaluated at p on every Jacobian: lmbc_core.c:1049 */
    orc_fdif_forward(s->f, p, s->hx, s->hxx, s->delta, jac, m, n, s->adata);
  } else
    orc_fdif_central(s->f, p, s->hx, s->hxx, s->delta, jac, m, n, s->adata);
}

int orc_dlevmar_bc_dif(orc_func_t f, double *p, double *x, int m, int n, double *lb, double *ub,
                       double *dscl, int itmax, double *opts, double *info, double *work,
                       double *covar, void *adata)
{
  struct fd_shim s;
  int ret;
  s.forward = !opts || opts[4] >= 0.0;
  s.f = f;
  s.hx = (double *)malloc((size_t)(2 * n) * sizeof(double));
  if (!s.hx) return ORC_ERROR;
  s.hxx = s.hx + n;
  s.adata = adata;
  s.delta = opts ? absd(opts[4]) : K_DIFF_DELTA;
  ret = orc_dlevmar_bc_der(fd_shim_func, fd_shim_jac, p, x, m, n, lb, ub, dscl, itmax, opts, info, work,
                           covar, &s);
  if (info) info[7] += info[8] * (s.forward ? (m + 1) : (2 * m)); /* lmbc_core.c:1119-1124 */
  free(s.hx);
  return ret;
}

int orc_brdf_fit(int method, int model, double *angles, double *x, int n, double *p, int itmax,
                 double *opts, double *lb, double *ub, double *info)
{
  struct orc_extra_data d;
  d.angles = angles;
  d.modelInfo = model;
  if (method == 0) return orc_dlevmar_dif(orc_brdf_func, p, x, 3, n, itmax, opts, info, NULL, NULL, &d);
  if (method == 3) /* dlevmar_der with the analytic Jacobian */
    return orc_dlevmar_der(orc_brdf_func, orc_brdf_jac, p, x, 3, n, itmax, opts, info, NULL, NULL, &d);
  if (method == 2) /* dlevmar_bc_der with the analytic Jacobian */
    return orc_dlevmar_bc_der(orc_brdf_func, orc_brdf_jac, p, x, 3, n, lb, ub, NULL, itmax, opts, info, NULL, NULL, &d);
  return orc_dlevmar_bc_dif(orc_brdf_func, p, x, 3, n, lb, ub, NULL, itmax, opts, info, NULL, NULL, &d);
}

/* S independent fits one after the other -- the shape of CalcBRDFEquation's pixel loop (brdfdata.cpp:1195-1220):
 * angles[S][3][n], x[S][n], p[S][3] in/out, info[S][10], ret[S].  Returns the number of fits that failed.  (Callers run
 * several ranges of surfels on several threads: the restatement keeps no static state.) */
int orc_brdf_fit_batch(int method, int model, double *angles, double *x, long S, int n, double *p, int itmax,
                       double *opts, double *lb, double *ub, double *info, int *ret)
{
  long s;
  int bad = 0;
  for (s = 0; s < S; ++s) {
    const int r = orc_brdf_fit(method, model, angles + s * 3 * n, x + s * n, n, p + 3 * s, itmax, opts, lb, ub, info + 10 * s);
    if (ret) ret[s] = r;
    if (r < 0) ++bad;
  }
  return bad;
}
