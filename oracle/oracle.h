/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("port") of the reference hot path, used as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under brdf_amd/ may include,
 * link or call anything declared here; the product path is HIP-only and fails loudly without its
 * extension.
 *
 * What is restated (reference = /root/reference, levmar 2.6 as vendored by ccalantzis/BRDF):
 *   orc_dlevmar_der      <- levmar/lm_core.c:64-432      (LM with the caller's analytic Jacobian)
 *   orc_dlevmar_dif      <- levmar/lm_core.c:438-842     (LM, FD Jacobian + Broyden rank-1 updates)
 *   orc_dlevmar_bc_der   <- levmar/lmbc_core.c:369-1022  (projected LM + line search + proj. gradient)
 *   orc_dlevmar_bc_dif   <- levmar/lmbc_core.c:1062-1129 (FD shim over bc_der)
 *   orc_l2_residual      <- levmar/misc_core.c:721-807
 *   orc_fdif_forward/central <- levmar/misc_core.c:137-211
 *   orc_jtj_blocked      <- levmar/misc_core.c:82-134
 *   orc_lu_solve         <- levmar/Axb_core.c:1140-1277
 *   orc_covar            <- levmar/misc_core.c:426-591
 *   orc_brdf_func        <- brdfdata.cpp:962-989 (Phong, Blinn-Phong) + build-defined Ward (model 2)
 *
 * Pinning: tests/test_oracle_kat.py replays the reference's own known answers (lmdemo.c problems, the
 * table in SURVEY.md section 4) through both this restatement and oracle/_ref (the reference's
 * sources compiled here) and requires bit-identical p / info[] between the two.
 */
#ifndef BRDF_ORACLE_H
#define BRDF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*orc_func_t)(double *p, double *hx, int m, int n, void *adata);
typedef void (*orc_jacf_t)(double *p, double *jac, int m, int n, void *adata);

#define ORC_INFO_SZ 10
#define ORC_ERROR   (-1)

/* adata payload of the BRDF callback, layout of brdfdata.cpp:962-966 */
struct orc_extra_data {
  double *angles; /* SoA: [0,n) cos(L.N)  [n,2n) cos(N.H)  [2n,3n) cos(R.V) (Phong) / cos(N.V) (Ward) */
  int modelInfo;  /* 0 Phong, 1 Blinn-Phong, 2 Ward (build-defined) */
};

void orc_brdf_func(double *p, double *hx, int m, int n, void *adata);
void orc_brdf_jac(double *p, double *jac, int m, int n, void *adata); /* analytic, n x 3 row-major; not in the reference */

double orc_l2_residual(double *e, const double *x, const double *y, int n);
void orc_fdif_forward(orc_func_t f, double *p, const double *hx, double *hxx, double delta,
                      double *jac, int m, int n, void *adata);
void orc_fdif_central(orc_func_t f, double *p, double *hxm, double *hxp, double delta,
                      double *jac, int m, int n, void *adata);
void orc_jtj_blocked(const double *a, double *b, int n, int m);
void orc_jtj_jte(const double *jac, const double *e, double *jtj, double *jte, int n, int m, int bc_rule);
int orc_lu_solve(const double *A, const double *B, double *x, int m);
int orc_covar(const double *JtJ, double *C, double sumsq, int m, int n);

int orc_dlevmar_dif(orc_func_t f, double *p, double *x, int m, int n, int itmax, double *opts,
                    double *info, double *work, double *covar, void *adata);
int orc_dlevmar_der(orc_func_t f, orc_jacf_t jf, double *p, double *x, int m, int n, int itmax, double *opts,
                    double *info, double *work, double *covar, void *adata);
int orc_dlevmar_bc_der(orc_func_t f, orc_jacf_t jf, double *p, double *x, int m, int n, double *lb,
                       double *ub, double *dscl, int itmax, double *opts, double *info, double *work,
                       double *covar, void *adata);
int orc_dlevmar_bc_dif(orc_func_t f, double *p, double *x, int m, int n, double *lb, double *ub,
                       double *dscl, int itmax, double *opts, double *info, double *work,
                       double *covar, void *adata);

/* convenience for ctypes callers: one BRDF fit with the reference call-site conventions
 * (brdfdata.cpp:1085-1119).  method 0 = dlevmar_dif, 1 = dlevmar_bc_dif, 2 = dlevmar_bc_der, 3 = dlevmar_der (both with orc_brdf_jac).
 * Returns the solver's return. */
int orc_brdf_fit(int method, int model, double *angles, double *x, int n, double *p, int itmax,
                 double *opts, double *lb, double *ub, double *info);

/* S fits one after the other (angles[S][3][n], x[S][n], p[S][3] in/out, info[S][10], ret[S]); returns the number of failed fits */
int orc_brdf_fit_batch(int method, int model, double *angles, double *x, long S, int n, double *p, int itmax,
                       double *opts, double *lb, double *ub, double *info, int *ret);

#ifdef __cplusplus
}
#endif
#endif
