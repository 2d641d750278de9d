// capi.hip -- the C ABI of libbrdf_hip.so (include/brdf_levmar.h).
//
// Host-side mirror of the reference's solver interface: dlevmar_dif / dlevmar_bc_dif keep the exact
// levmar.h:112-127 signatures and semantics (return value, info[], NULL-able opts/info/work/covar,
// stderr diagnostics, LM_ERROR instead of exit()).  Everything n-sized runs in the HIP kernels of
// stream_fit.hip / batch_fit.hip; there is no CPU evaluation path in this library.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>

#include <cfloat>
#include <vector>

#include "../../include/brdf_levmar.h"
#include "batch_fit.h"
#include "stream_fit.h"

namespace brdf {
int generic_fit_run(int method, void (*func)(double *, double *, int, int, void *),
                    void (*jacf)(double *, double *, int, int, void *), double *p, double *x, int m, int n, double *lb,
                    double *ub, double *dscl, int itmax, double *opts, double *info, double *covar, void *adata);
int model_eval_run(int model, const double *d_angles, int n, const double *p, double *d_hx, hipStream_t stream);
int model_jac_run(int model, const double *d_angles, int n, const double *p, double *d_jac, hipStream_t stream);
int chkjac_err_run(const double *fvec, const double *fjac, const double *fvecp, const double *p, int m, int n, double *err);
int r2_run(const double *x, const double *hx, int n, double *r2);
int generic_fit_run_f(int method, void (*func)(float *, float *, int, int, void *), void (*jacf)(float *, float *, int, int, void *),
                      float *p, float *x, int m, int n, float *lb, float *ub, float *dscl, int itmax, float *opts, float *info,
                      float *covar, void *adata);
int chkjac_err_run_f(const float *fvec, const float *fjac, const float *fvecp, const float *p, int m, int n, float *err);
int r2_run_f(const float *x, const float *hx, int n, float *r2);
}

using namespace brdf;

namespace {

// levmar documents a FOUR-element opts array for the analytic-Jacobian entry points (lm_core.c:70-75, lmbc_core.c:380):
// they are widened here so that nothing below ever reads a fifth element of the caller's array
struct Opts4 {
  double o[5];
  double *ptr;
  explicit Opts4(double *opts) : ptr(opts ? o : nullptr) {
    if (opts) {
      for (int i = 0; i < 4; ++i) o[i] = opts[i];
      o[4] = LM_DIFF_DELTA;
    }
  }
};

typedef void (*model_func_t)(double *, double *, int, int, void *);
constexpr int kMaxRegistered = 16;
model_func_t g_registered[kMaxRegistered] = {nullptr};
std::mutex g_reg_mutex;

bool is_registered(model_func_t f) {
  if (f == &BRDFFunc_hip) return true;
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  for (int i = 0; i < kMaxRegistered; ++i)
    if (g_registered[i] == f) return true;
  return false;
}

// Grow-only device staging of the drop-in entry points (one per host thread): a dlevmar_* call with host pointers
// uploads its planes and measurements here.  The first version hipMalloc'ed and hipFree'd two buffers per call -- at the
// application's call site (brdfdata.cpp:1119: n = 16, once per pixel and colour channel) that pair cost more than the fit.
struct HostStage {
  double *d = nullptr;
  size_t cap = 0;
  int dev = -1;
  // the block belongs to device `dev`: work queued THERE has to drain before it is freed, whatever device is current now
  void release() {
    if (!d) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    (void)hipFree(d);
    if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
    d = nullptr;
    cap = 0;
  }
  ~HostStage() { release(); }  // a host thread that ends gives its staging block back
  double *get(size_t count) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (d && cur == dev && cap >= count) return d;
    release();
    const size_t want = count + count / 2 + 1024;
    if (hipMalloc(&d, want * sizeof(double)) != hipSuccess) {
      set_error("hipMalloc(%zu doubles) failed", want);
      d = nullptr;
      return nullptr;
    }
    cap = want;
    dev = cur;
    return d;
  }
};
thread_local HostStage g_stage;

// uploads the planes `modelInfo` reads (Blinn-Phong never reads plane 3, which the reference's per-surfel caller
// allocates and fills incorrectly, brdfdata.cpp:1095, :1102) into d[0, 3n): adjacent planes travel in one copy
hipError_t upload_planes(double *d, const double *angles, int n, int model) {
  if (model == MODEL_WARD) return hipMemcpy(d, angles, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice);
  if (model == MODEL_BLINN_PHONG) return hipMemcpy(d, angles, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice);
  hipError_t e = hipMemcpy(d, angles, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + 2 * (size_t)n, angles + 2 * (size_t)n, sizeof(double) * n, hipMemcpyHostToDevice);
  return e;
}

// scoped device buffer
struct DevBuf {
  double *ptr = nullptr;
  ~DevBuf() {
    if (ptr) (void)hipFree(ptr);
  }
  int alloc(size_t count) {
    hipError_t e = hipMalloc(&ptr, count * sizeof(double));
    if (e != hipSuccess) {
      set_error("hipMalloc(%zu doubles) failed: %s", count, hipGetErrorString(e));
      ptr = nullptr;
      return -1;
    }
    return 0;
  }
};

int host_fit(int method, const char *who, model_func_t func, double *p, double *x, int m, int n, double *lb,
             double *ub, double *dscl, int itmax, double *opts, double *info, double *covar, void *adata, int analytic = 0) {
  if (!func) {
    set_error("%s(): func is NULL", who);
    return LM_ERROR;
  }
  if (!is_registered(func))  // arbitrary host callback: evaluated on the host, all n-sized algebra on the device
    return generic_fit_run(method, func, nullptr, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  if (m != kM) {
    set_error("%s(): the BRDF models have exactly 3 parameters (got m=%d)", who, m);
    return LM_ERROR;
  }
  if (n < m) {  // lm_core.c:502-505, lmbc_core.c:440-443
    set_error("%s(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", who, n, m);
    return LM_ERROR;
  }
  if (!adata || !p) {
    set_error("%s(): adata (struct extraData) and p must not be NULL", who);
    return LM_ERROR;
  }
  const brdf_extra_data *ed = static_cast<const brdf_extra_data *>(adata);
  if (!ed->angles) {
    set_error("%s(): extraData.angles is NULL", who);
    return LM_ERROR;
  }
  if (ed->modelInfo < 0 || ed->modelInfo >= MODEL_COUNT) {
    set_error("%s(): extraData.modelInfo=%d is not a known model (the reference would leave hx unwritten)", who,
              ed->modelInfo);
    return LM_ERROR;
  }
  double *stage = g_stage.get(4 * (size_t)n);
  if (!stage) return LM_ERROR;
  double *d_angles = stage, *d_x = stage + 3 * (size_t)n;
  hipError_t e = upload_planes(d_angles, ed->angles, n, ed->modelInfo);
  if (e == hipSuccess)  // "NULL implies a zero vector", lm_core.c:441 / misc_core.c:770
    e = x ? hipMemcpy(d_x, x, sizeof(double) * n, hipMemcpyHostToDevice) : hipMemset(d_x, 0, sizeof(double) * n);
  if (e != hipSuccess) {
    set_error("%s(): host->device copy failed: %s", who, hipGetErrorString(e));
    return LM_ERROR;
  }
  StreamFitArgs a;
  a.method = method;
  a.analytic = analytic;
  a.model = ed->modelInfo;
  a.d_angles = d_angles;
  a.d_x = d_x;
  a.n = n;
  a.p = p;
  a.lb = lb;
  a.ub = ub;
  a.dscl = dscl;
  a.itmax = itmax;
  a.opts = opts;
  a.info = info;
  a.covar = covar;
  a.stream = nullptr;
  // (the fit has finished when this returns: launches of the chain still queued behind it only read its `done` word,
  // never the staging buffer, so the next call may overwrite the buffer at once)
  return stream_fit_run(a);
}

}  // namespace

namespace {
/* Axb_core.c:1197-1270 for a run-time m: Crout LU with implicit row scaling + partial pivoting, zero pivot -> LmLimits<Real>::eps(),
 * forward and back substitution.  Same operations in the same order as lm_machine.h: lu_solve<M> (which the fitter uses in
 * registers and which the reference's known answers pin bit for bit). */
template <class Real>
static int lu_noLapack(Real *A, Real *B, Real *x, int m, const char *who) {
  if (!A) return 1;  // Axb_core.c:1149-1157: "release the retained buffer" -- there is none here
  if (!B || !x || m <= 0) {
    set_error("%s(): bad arguments", who);
    return 0;
  }
  std::vector<Real> a(A, A + (size_t)m * m), scale(m);
  std::vector<int> perm(m);
  for (int i = 0; i < m; ++i) x[i] = B[i];
  for (int i = 0; i < m; ++i) {
    Real big = Real(0.0);
    for (int j = 0; j < m; ++j) {
      const Real t = std::fabs(a[(size_t)i * m + j]);
      if (t > big) big = t;
    }
    if (big == Real(0.0)) return 0;  // silently, as the reference does (its message is commented out, Axb_core.c:1202-1208)
    scale[i] = Real(1.0) / big;
  }
  for (int j = 0; j < m; ++j) {
    int pivot = j;
    Real big = Real(0.0);
    for (int i = 0; i < j; ++i) {
      Real s = a[(size_t)i * m + j];
      for (int k = 0; k < i; ++k) s -= a[(size_t)i * m + k] * a[(size_t)k * m + j];
      a[(size_t)i * m + j] = s;
    }
    for (int i = j; i < m; ++i) {
      Real s = a[(size_t)i * m + j];
      for (int k = 0; k < j; ++k) s -= a[(size_t)i * m + k] * a[(size_t)k * m + j];
      a[(size_t)i * m + j] = s;
      const Real t = scale[i] * std::fabs(s);
      if (t >= big) {
        big = t;
        pivot = i;
      }
    }
    if (j != pivot) {
      for (int k = 0; k < m; ++k) std::swap(a[(size_t)pivot * m + k], a[(size_t)j * m + k]);
      scale[pivot] = scale[j];
    }
    perm[j] = pivot;
    if (a[(size_t)j * m + j] == Real(0.0)) a[(size_t)j * m + j] = LmLimits<Real>::eps();
    if (j != m - 1) {
      const Real t = Real(1.0) / a[(size_t)j * m + j];
      for (int i = j + 1; i < m; ++i) a[(size_t)i * m + j] *= t;
    }
  }
  int first = 0;
  for (int i = 0; i < m; ++i) {
    const int ip = perm[i];
    Real s = x[ip];
    x[ip] = x[i];
    if (first != 0) {
      for (int jj = first - 1; jj < i; ++jj) s -= a[(size_t)i * m + jj] * x[jj];
    } else if (s != Real(0.0)) {
      first = i + 1;
    }
    x[i] = s;
  }
  for (int i = m - 1; i >= 0; --i) {
    Real s = x[i];
    for (int j = i + 1; j < m; ++j) s -= a[(size_t)i * m + j] * x[j];
    x[i] = s / a[(size_t)i * m + i];
  }
  return 1;
}

}  // namespace

extern "C" {

int dlevmar_dif(void (*func)(double *, double *, int, int, void *), double *p, double *x, int m, int n, int itmax,
                double *opts, double *info, double * /*work*/, double *covar, void *adata) {
  return host_fit(BRDF_METHOD_DIF, "dlevmar_dif", func, p, x, m, n, nullptr, nullptr, nullptr, itmax, opts, info,
                  covar, adata);
}

int dlevmar_bc_dif(void (*func)(double *, double *, int, int, void *), double *p, double *x, int m, int n,
                   double *lb, double *ub, double *dscl, int itmax, double *opts, double *info, double * /*work*/,
                   double *covar, void *adata) {
  return host_fit(BRDF_METHOD_BC_DIF, "dlevmar_bc_dif", func, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar,
                  adata);
}

int dlevmar_der(void (*func)(double *, double *, int, int, void *), void (*jacf)(double *, double *, int, int, void *),
                double *p, double *x, int m, int n, int itmax, double *opts, double *info, double * /*work*/, double *covar,
                void *adata) {
  if (!func) {
    set_error("dlevmar_der(): func is NULL");
    return LM_ERROR;
  }
  if (!jacf) {  // lm_core.c:126-130
    set_error("No function specified for computing the Jacobian in dlevmar_der(). If no such function is available, use "
              "dlevmar_dif() rather than dlevmar_der()");
    return LM_ERROR;
  }
  Opts4 o4(opts);
  opts = o4.ptr;
  if (is_registered(func) && jacf == &BRDFJac_hip)  // a built-in model with its own analytic Jacobian: all on the device
    return host_fit(2, "dlevmar_der", func, p, x, m, n, nullptr, nullptr, nullptr, itmax, opts, info, covar, adata, 1);
  return generic_fit_run(2, func, jacf, p, x, m, n, nullptr, nullptr, nullptr, itmax, opts, info, covar, adata);
}

int dlevmar_bc_der(void (*func)(double *, double *, int, int, void *), void (*jacf)(double *, double *, int, int, void *),
                   double *p, double *x, int m, int n, double *lb, double *ub, double *dscl, int itmax, double *opts,
                   double *info, double * /*work*/, double *covar, void *adata) {
  if (!func) {
    set_error("dlevmar_bc_der(): func is NULL");
    return LM_ERROR;
  }
  if (!jacf) {  // lmbc_core.c:445-449
    set_error("No function specified for computing the Jacobian in dlevmar_bc_der(). If no such function is available, "
              "use dlevmar_bc_dif() rather than dlevmar_bc_der()");
    return LM_ERROR;
  }
  Opts4 o4(opts);
  opts = o4.ptr;
  if (is_registered(func) && jacf == &BRDFJac_hip)  // a built-in model with its own analytic Jacobian: all on the device
    return host_fit(BRDF_METHOD_BC_DIF, "dlevmar_bc_der", func, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata,
                    /*analytic=*/1);
  return generic_fit_run(BRDF_METHOD_BC_DIF, func, jacf, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata);
}

/* scalar post-processing of the covariance the solvers return (misc_core.c:598-611); no n-sized work */
double dlevmar_stddev(double *covar, int m, int i) { return sqrt(covar[i * m + i]); }
double dlevmar_corcoef(double *covar, int m, int i, int j) { return covar[i * m + j] / sqrt(covar[i * m + i] * covar[j * m + j]); }

double dlevmar_R2(void (*func)(double *, double *, int, int, void *), double *p, double *x, int m, int n, void *adata) {
  if (!func || !p || m <= 0 || n <= 0) {
    set_error("dlevmar_R2(): bad arguments");
    return NAN;
  }
  std::vector<double> hx(n);
  (*func)(p, hx.data(), m, n, adata);  // misc_core.c:632
  double r2 = NAN;
  (void)r2_run(x, hx.data(), n, &r2);
  return r2;
}

int dAx_eq_b_LU_noLapack(double *A, double *B, double *x, int m) { return lu_noLapack<double>(A, B, x, m, "dAx_eq_b_LU_noLapack"); }
int sAx_eq_b_LU_noLapack(float *A, float *B, float *x, int m) { return lu_noLapack<float>(A, B, x, m, "sAx_eq_b_LU_noLapack"); }

/* ---- the single-precision twins, levmar/levmar.h:208-310 (instantiated in the reference from the same *_core.c files with
 * LM_REAL = float, lm.c:43-63; here: the same machines and kernels with Real = float).  The callbacks are the caller's host
 * code and run on the host; everything n-sized around them runs on the device, in float, in the reference's summation
 * order for n*m <= 65536.  (There is no registered-model shortcut: the BRDF application is double-only.) */
int slevmar_dif(void (*func)(float *, float *, int, int, void *), float *p, float *x, int m, int n, int itmax, float *opts,
                float *info, float * /*work*/, float *covar, void *adata) {
  return generic_fit_run_f(0, func, nullptr, p, x, m, n, nullptr, nullptr, nullptr, itmax, opts, info, covar, adata);
}
int slevmar_bc_dif(void (*func)(float *, float *, int, int, void *), float *p, float *x, int m, int n, float *lb, float *ub,
                   float *dscl, int itmax, float *opts, float *info, float * /*work*/, float *covar, void *adata) {
  return generic_fit_run_f(1, func, nullptr, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata);
}
int slevmar_der(void (*func)(float *, float *, int, int, void *), void (*jacf)(float *, float *, int, int, void *), float *p, float *x,
                int m, int n, int itmax, float *opts, float *info, float * /*work*/, float *covar, void *adata) {
  if (!jacf) {  // lm_core.c:126-130
    set_error("No function specified for computing the Jacobian in slevmar_der(). If no such function is available, use "
              "slevmar_dif() rather than slevmar_der()");
    return LM_ERROR;
  }
  float o5[5] = {0, 0, 0, 0, (float)LM_DIFF_DELTA};  // four documented elements (see Opts4)
  if (opts)
    for (int i = 0; i < 4; ++i) o5[i] = opts[i];
  return generic_fit_run_f(2, func, jacf, p, x, m, n, nullptr, nullptr, nullptr, itmax, opts ? o5 : nullptr, info, covar, adata);
}
int slevmar_bc_der(void (*func)(float *, float *, int, int, void *), void (*jacf)(float *, float *, int, int, void *), float *p,
                   float *x, int m, int n, float *lb, float *ub, float *dscl, int itmax, float *opts, float *info, float * /*work*/,
                   float *covar, void *adata) {
  if (!jacf) {  // lmbc_core.c:445-449
    set_error("No function specified for computing the Jacobian in slevmar_bc_der(). If no such function is available, "
              "use slevmar_bc_dif() rather than slevmar_bc_der()");
    return LM_ERROR;
  }
  float o5[5] = {0, 0, 0, 0, (float)LM_DIFF_DELTA};
  if (opts)
    for (int i = 0; i < 4; ++i) o5[i] = opts[i];
  return generic_fit_run_f(1, func, jacf, p, x, m, n, lb, ub, dscl, itmax, opts ? o5 : nullptr, info, covar, adata);
}
float slevmar_stddev(float *covar, int m, int i) { return (float)sqrt(covar[i * m + i]); }  /* misc_core.c:598-602 */
float slevmar_corcoef(float *covar, int m, int i, int j) { return (float)(covar[i * m + j] / sqrt(covar[i * m + i] * covar[j * m + j])); }
float slevmar_R2(void (*func)(float *, float *, int, int, void *), float *p, float *x, int m, int n, void *adata) {
  if (!func || !p || m <= 0 || n <= 0) {
    set_error("slevmar_R2(): bad arguments");
    return NAN;
  }
  std::vector<float> hx(n);
  (*func)(p, hx.data(), m, n, adata);
  float r2 = NAN;
  (void)r2_run_f(x, hx.data(), n, &r2);
  return r2;
}
void slevmar_chkjac(void (*func)(float *, float *, int, int, void *), void (*jacf)(float *, float *, int, int, void *), float *p, int m,
                    int n, void *adata, float *err) {
  if (!func || !jacf || !p || !err || m <= 0 || n <= 0) {
    set_error("slevmar_chkjac(): bad arguments");
    return;
  }
  std::vector<float> fvec(n), fjac((size_t)n * m), pp(m), fvecp(n);
  const float eps = sqrtf(FLT_EPSILON);
  (*func)(p, fvec.data(), m, n, adata);
  (*jacf)(p, fjac.data(), m, n, adata);
  for (int j = 0; j < m; ++j) {
    float temp = eps * fabsf(p[j]);
    if (temp == 0.0f) temp = eps;
    pp[j] = p[j] + temp;
  }
  (*func)(pp.data(), fvecp.data(), m, n, adata);
  (void)chkjac_err_run_f(fvec.data(), fjac.data(), fvecp.data(), p, m, n, err);
}

int brdf_hip_register_model(void (*func)(double *, double *, int, int, void *)) {
  if (!func) return -1;
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  for (int i = 0; i < kMaxRegistered; ++i)
    if (g_registered[i] == func) return 0;
  for (int i = 0; i < kMaxRegistered; ++i)
    if (!g_registered[i]) {
      g_registered[i] = func;
      return 0;
    }
  set_error("brdf_hip_register_model(): table full (%d entries)", kMaxRegistered);
  return -1;
}

int brdf_hip_unregister_model(void (*func)(double *, double *, int, int, void *)) {
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  for (int i = 0; i < kMaxRegistered; ++i)
    if (g_registered[i] == func) {
      g_registered[i] = nullptr;
      return 0;
    }
  return -1;
}

void BRDFFunc_hip(double *p, double *hx, int m, int n, void *adata) {
  const brdf_extra_data *ed = static_cast<const brdf_extra_data *>(adata);
  if (!p || !hx || !ed || !ed->angles || m != kM || n <= 0) {
    set_error("BRDFFunc_hip(): bad arguments");
    return;
  }
  if (ed->modelInfo < 0 || ed->modelInfo >= MODEL_COUNT) return;  // reference: hx left unwritten
  double *stage = g_stage.get(4 * (size_t)n);
  if (!stage) return;
  double *d_out = stage + 3 * (size_t)n;
  hipError_t e = upload_planes(stage, ed->angles, n, ed->modelInfo);
  if (e != hipSuccess) {
    set_error("BRDFFunc_hip(): host->device copy failed: %s", hipGetErrorString(e));
    return;
  }
  if (model_eval_run(ed->modelInfo, stage, n, p, d_out, nullptr) != 0) return;
  e = hipMemcpy(hx, d_out, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (e != hipSuccess) set_error("BRDFFunc_hip(): device->host copy failed: %s", hipGetErrorString(e));
}

void BRDFJac_hip(double *p, double *jac, int m, int n, void *adata) {
  const brdf_extra_data *ed = static_cast<const brdf_extra_data *>(adata);
  if (!p || !jac || !ed || !ed->angles || m != kM || n <= 0) {
    set_error("BRDFJac_hip(): bad arguments");
    return;
  }
  if (ed->modelInfo < 0 || ed->modelInfo >= MODEL_COUNT) return;
  double *stage = g_stage.get(6 * (size_t)n);
  if (!stage) return;
  double *d_out = stage + 3 * (size_t)n;
  hipError_t e = upload_planes(stage, ed->angles, n, ed->modelInfo);
  if (e != hipSuccess) {
    set_error("BRDFJac_hip(): host->device copy failed: %s", hipGetErrorString(e));
    return;
  }
  if (model_jac_run(ed->modelInfo, stage, n, p, d_out, nullptr) != 0) return;
  e = hipMemcpy(jac, d_out, sizeof(double) * 3 * n, hipMemcpyDeviceToHost);
  if (e != hipSuccess) set_error("BRDFJac_hip(): device->host copy failed: %s", hipGetErrorString(e));
}

void dlevmar_chkjac(void (*func)(double *, double *, int, int, void *), void (*jacf)(double *, double *, int, int, void *),
                    double *p, int m, int n, void *adata, double *err) {
  if (!func || !jacf || !p || !err || m <= 0 || n <= 0) {
    set_error("dlevmar_chkjac(): bad arguments");
    return;
  }
  // misc_core.c:268-301: f(p), J(p), f(p + eps |p|) through the caller's callbacks; the n-sized comparison on the device
  std::vector<double> fvec(n), fjac((size_t)n * m), pp(m), fvecp(n);
  const double eps = sqrt(DBL_EPSILON);
  (*func)(p, fvec.data(), m, n, adata);
  (*jacf)(p, fjac.data(), m, n, adata);
  for (int j = 0; j < m; ++j) {
    double temp = eps * fabs(p[j]);
    if (temp == 0.0) temp = eps;
    pp[j] = p[j] + temp;
  }
  (*func)(pp.data(), fvecp.data(), m, n, adata);
  (void)chkjac_err_run(fvec.data(), fjac.data(), fvecp.data(), p, m, n, err);
}

int brdf_hip_fit_dev(int method, int model, const double *d_angles, const double *d_x, int n, double *p,
                     const double *lb, const double *ub, const double *dscl, int itmax, const double *opts,
                     double *info, double *covar, void *stream) {
  StreamFitArgs a;
  a.method = (method == BRDF_METHOD_BC_DER) ? 1 : (method == BRDF_METHOD_DER ? 2 : method);  // internal: 2 = DerMachine
  a.analytic = (method == BRDF_METHOD_BC_DER || method == BRDF_METHOD_DER) ? 1 : 0;
  a.model = model;
  a.d_angles = d_angles;
  a.d_x = d_x;
  a.n = n;
  a.p = p;
  a.lb = lb;
  a.ub = ub;
  a.dscl = dscl;
  a.itmax = itmax;
  a.opts = opts;
  a.info = info;
  a.covar = covar;
  a.stream = static_cast<hipStream_t>(stream);
  return stream_fit_run(a);
}

int brdf_hip_fit_channels_dev(int method, int model, const double *d_angles, const double *d_x, long long x_stride, int n, int channels,
                              double *p, const double *lb, const double *ub, const double *dscl, int itmax, const double *opts,
                              double *info, double *covar, void *stream) {
  if (!d_angles || !d_x || !p || n <= 0 || channels < 1 || channels > 16 || (channels > 1 && x_stride < n)) {
    set_error("brdf_hip_fit_channels_dev(): bad arguments");
    return LM_ERROR;
  }
  if (method < 0 || method > 3 || model < 0 || model >= MODEL_COUNT) {
    set_error("brdf_hip_fit_channels_dev(): unknown method %d or model %d", method, model);
    return LM_ERROR;
  }
  return channels_fit_run(method, model, d_angles, d_x, x_stride, n, channels, p, lb, ub, dscl, itmax, opts, info, covar,
                          static_cast<hipStream_t>(stream));
}

/* diagnostic builds (-DBRDF_STAMPS) only: cycles per section of channel `channel`'s control wave, summed over its passes */
int brdf_hip_last_channels_stamps(int channel, long long *out8) {
  const FitStats st = channels_last_stats(channel);
  for (int i = 0; i < 8; ++i) out8[i] = st.stamps[i];
  return 0;
}

int brdf_hip_last_channels_stats(int channel, int *shared_launch, long long *passes, long long *jac_passes, double *device_us) {
  const FitStats st = channels_last_stats(channel);
  if (shared_launch) *shared_launch = channels_last_shared();
  if (passes) *passes = st.passes;
  if (jac_passes) *jac_passes = st.jac_passes;
  if (device_us) *device_us = st.device_us;
  return 0;
}

int brdf_hip_fit_batch_dev(int method, int model, const double *d_angles, const double *d_x, int S, int n,
                           double *d_p, const double *lb, const double *ub, int itmax, const double *opts,
                           double *d_info, int *d_ret, void *stream) {
  BatchFitArgs a;
  a.method = method;
  a.model = model;
  a.d_angles = d_angles;
  a.d_x = d_x;
  a.S = S;
  a.n = n;
  a.d_p = d_p;
  a.lb = lb;
  a.ub = ub;
  a.itmax = itmax;
  a.opts = opts;
  a.d_info = d_info;
  a.d_ret = d_ret;
  a.stream = static_cast<hipStream_t>(stream);
  return batch_fit_enqueue(a);
}

int brdf_hip_fit_batch(int method, int model, const double *angles, const double *x, int S, int n, double *p,
                       const double *lb, const double *ub, int itmax, const double *opts, double *info, int *ret) {
  if (!angles || !x || !p || S <= 0 || n <= 0) {
    set_error("brdf_hip_fit_batch(): bad arguments");
    return LM_ERROR;
  }
  const size_t sn = (size_t)S * n;
  DevBuf d_angles, d_x, d_p, d_info, d_ret;
  if (d_angles.alloc(3 * sn) || d_x.alloc(sn) || d_p.alloc(3 * (size_t)S) || d_info.alloc(10 * (size_t)S) ||
      d_ret.alloc(((size_t)S + 1) / 2 + 1))
    return LM_ERROR;
  hipError_t e = hipMemcpy(d_angles.ptr, angles, sizeof(double) * 3 * sn, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_x.ptr, x, sizeof(double) * sn, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_p.ptr, p, sizeof(double) * 3 * S, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("brdf_hip_fit_batch(): host->device copy failed: %s", hipGetErrorString(e));
    return LM_ERROR;
  }
  int *d_ret_i = reinterpret_cast<int *>(d_ret.ptr);
  if (brdf_hip_fit_batch_dev(method, model, d_angles.ptr, d_x.ptr, S, n, d_p.ptr, lb, ub, itmax, opts, d_info.ptr,
                             d_ret_i, nullptr) != 0)
    return LM_ERROR;
  e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(p, d_p.ptr, sizeof(double) * 3 * S, hipMemcpyDeviceToHost);
  if (e == hipSuccess && info) e = hipMemcpy(info, d_info.ptr, sizeof(double) * 10 * S, hipMemcpyDeviceToHost);
  int *host_ret = ret;
  int *tmp = nullptr;
  if (!host_ret) host_ret = tmp = new int[S];
  if (e == hipSuccess) e = hipMemcpy(host_ret, d_ret_i, sizeof(int) * S, hipMemcpyDeviceToHost);
  int bad = 0;
  if (e == hipSuccess)
    for (int s = 0; s < S; ++s) bad += host_ret[s] < 0;
  delete[] tmp;
  if (e != hipSuccess) {
    set_error("brdf_hip_fit_batch(): %s", hipGetErrorString(e));
    return LM_ERROR;
  }
  return bad;
}

int brdf_hip_model_eval_dev(int model, const double *d_angles, int n, const double *p, double *d_hx, void *stream) {
  return model_eval_run(model, d_angles, n, p, d_hx, static_cast<hipStream_t>(stream));
}

int brdf_hip_synth_dev(int model, unsigned long long seed, long long first, int count, int n,
                       const double *d_truth, double *d_angles, double *d_x, void *stream) {
  return synth_enqueue(model, seed, first, count, n, d_truth, d_angles, d_x, static_cast<hipStream_t>(stream));
}

int brdf_hip_cosines_dev(const double *d_vertices, const int *d_faces, const double *d_face_normals, const int *d_surfels,
                         long long S, const double *leds, int L, const double *view_origin, int rv_mode, double *d_angles,
                         void *stream) {
  return cosines_run(d_vertices, d_faces, d_face_normals, d_surfels, S, leds, L, view_origin, rv_mode, d_angles,
                     static_cast<hipStream_t>(stream));
}

void brdf_hip_led_table(double *leds16x3) { led_table(leds16x3); }

int brdf_hip_fit_capture_dev(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                             const double *d_vertices, const int *d_faces, const double *d_face_normals, int nf,
                             const double *leds, const double *view_origin, int rv_mode, const double *p0, const double *lb,
                             const double *ub, int itmax, const double *opts, double *d_brdf_surfaces, double *avg,
                             long long *n_pixels, void *stream) {
  return capture_fit_run(model, d_images, L, H, W, d_pixel_map, d_vertices, d_faces, d_face_normals, nf, leds, view_origin,
                         rv_mode, p0, lb, ub, itmax, opts, d_brdf_surfaces, avg, n_pixels, static_cast<hipStream_t>(stream));
}

int brdf_hip_fit_capture_single_dev(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                                    const double *d_vertices, const int *d_faces, const double *d_face_normals, int nf,
                                    const double *leds, const double *view_origin, int rv_mode, const double *p0,
                                    const double *lb, const double *ub, int itmax, const double *opts, double *single_brdf,
                                    double *info, long long *n_faces_used, void *stream) {
  return capture_fit_single_run(model, d_images, L, H, W, d_pixel_map, d_vertices, d_faces, d_face_normals, nf, leds, view_origin,
                                rv_mode, p0, lb, ub, itmax, opts, single_brdf, info, n_faces_used, static_cast<hipStream_t>(stream));
}

int brdf_hip_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

const char *brdf_hip_last_error(void) { return get_error(); }

int brdf_hip_last_fit_stats(long long *passes, long long *jac_passes, long long *eval_passes, double *device_us) {
  const FitStats s = stream_fit_last_stats();
  if (passes) *passes = s.passes;
  if (jac_passes) *jac_passes = s.jac_passes;
  if (eval_passes) *eval_passes = s.eval_passes;
  if (device_us) *device_us = s.device_us;
  return 0;
}

long long brdf_hip_last_fit_launches(void) { return stream_fit_last_stats().launches; }

void brdf_hip_set_launch_timing(int on) { set_launch_timing(on != 0); }
double brdf_hip_last_fit_kernel_us(void) { return stream_fit_last_stats().kernel_us; }
double brdf_hip_last_channels_kernel_us(void) { return channels_last_shared() ? channels_last_stats(0).kernel_us : -1.0; }

/* diagnostic builds (-DBRDF_STAMPS) only: one epoch's timeline of every workgroup of the last resident fit */
int brdf_hip_last_fit_trace(long long *out, int max_rows) { return resident_fit_last_trace(out, max_rows); }

/* diagnostic builds (-DBRDF_STAMPS) only: cycles per section of the pass kernel, summed over passes */
int brdf_hip_last_fit_stamps(long long *out8) {
  const FitStats s = stream_fit_last_stats();
  for (int k = 0; k < 8; ++k) out8[k] = s.stamps[k];
  return 0;
}

}  // extern "C"
