// generic_fit.hip -- dlevmar_dif / dlevmar_bc_dif for an ARBITRARY host callback (levmar.h:112-127).
//
// A host function pointer cannot run on the GPU, and it is the caller's code: it is evaluated on the host,
// exactly when and where the reference would call it (same points, same order, same in-place perturbation
// pattern is not needed because the callback only ever sees a private copy of p).  Everything n-sized the
// reference does around those calls runs on the device:
//   e = x - hx and ||e||^2                     misc_core.c:721-807   (K2)
//   forward / central difference Jacobian fill   misc_core.c:153-171, :191-210   (K3)
//   J^T J and J^T e                              lm_core.c:617-653, misc_core.c:82-134   (K4)
//   Broyden rank-one update of J                 lm_core.c:760-766   (K5)
// and the LM state machines of lm_machine.h (instantiated for the caller's m) sequence the passes from the
// host, since every pass needs a host callback anyway.
//
// Summation order.  For problems with n*m <= 65536 the sums are formed by ONE lane in the reference's own
// order (4-accumulator descending residual norm; descending "small problem" loop or 32-row blocked loop for
// J^T J, with the reference's n*m <= 1024 / < 1024 switch): results are then bit-identical to levmar's, which
// is what lets the reference's own known answers (lmdemo.c) be replayed through this ABI.
// Larger problems use a deterministic workgroup tree (one workgroup; the callback dominates there anyway).
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include "stream_fit.h"

namespace brdf {

constexpr int kGenMaxM = 16;  // (levmar takes any m, lm_core.c:528-548; the machines are instantiated per m: 1..16 here)
constexpr int kGenSums = kGenMaxM * (kGenMaxM + 1) / 2 + kGenMaxM + 2;  // JtJ lower + Jte + ||e||^2 + max
constexpr int kGenThreads = 256;
constexpr int kGenExactLimit = 65536;

template <class Real>
struct GenArgs {
  const Real *x;    // measurements (zeros if the caller passed NULL)
  const Real *hx;   // f(p)                       (device copy kept across passes for dif)
  const Real *aux;  // RQ_EVAL: f(point); RQ_JAC/RQ_DIF_JAC: hxx planes [j][n] (central: [2j] minus, [2j+1] plus)
  Real *wrk;        // dif: f(p + Dp) of the last trial
  Real *hx_rw;      // writable alias of hx (accept: hx <- wrk)
  Real *jac;        // n x m row-major, as levmar stores it (jac[i*m + j])
  Real *out;        // kGenSums doubles
  int n, m, kind, central, exact, accepted, bc_rule, store_hx, user_jac;
  Real dinv[kGenMaxM], dp[kGenMaxM], dp_l2, scal;
};

// ---- exact (reference-order) single-lane routines ---------------------------------------------------------
template <class Real>
__device__ Real ref_l2(const Real *x, const Real *y, int n, Real scal, bool scaled, Real *mx_out) {
  Real acc[4] = {Real(0.0), Real(0.0), Real(0.0), Real(0.0)};
  Real mx = Real(0.0);
  if (scaled) {  // lmbc_core.c:163-166, descending
    Real s = Real(0.0);
    for (int i = n; i-- > 0;) {
      const Real t = (x[i] - y[i]) / scal;
      s += t * t;
    }
    *mx_out = Real(0.0);
    return s;
  }
  const int body = (n >> 3) << 3;  // misc_core.c:732-768
  for (int top = body - 1; top > 0; top -= 8)
    for (int k = 0; k < 8; ++k) {
      const Real e = x[top - k] - y[top - k];
      acc[k & 3] += e * e;
      mx = fmax(mx, fabs(e));
    }
  for (int t = body; t < n; ++t) {
    const Real e = x[t] - y[t];
    acc[(7 - (n - t)) & 3] += e * e;
    mx = fmax(mx, fabs(e));
  }
  *mx_out = mx;
  return acc[0] + acc[1] + acc[2] + acc[3];
}

template <class Real>
__device__ void ref_jtj_jte(const Real *jac, const Real *x, const Real *h, int n, int m, bool small, Real *jtj /*m*m*/,
                            Real *jte) {
  if (small) {  // lm_core.c:617-637 / lmbc_core.c:595-615
    for (int i = m * m; i-- > 0;) jtj[i] = Real(0.0);
    for (int i = m; i-- > 0;) jte[i] = Real(0.0);
    for (int l = n; l-- > 0;) {
      const Real *row = jac + (size_t)l * m;
      const Real el = x[l] - h[l];
      for (int i = m; i-- > 0;) {
        const Real alpha = row[i];
        for (int j = i + 1; j-- > 0;) jtj[i * m + j] += row[j] * alpha;
        jte[i] += alpha * el;
      }
    }
  } else {  // misc_core.c:103-128 (32-row blocks, upper triangle) + lm_core.c:645-653
    for (int i = 0; i < m; ++i)
      for (int j = i; j < m; ++j) jtj[i * m + j] = Real(0.0);
    for (int kk = 0; kk < n; kk += 32) {
      const int kend = (kk + 32 <= n) ? kk + 32 : n;
      for (int i = 0; i < m; ++i)
        for (int j = i; j < m; ++j) {
          Real s = Real(0.0);
          for (int k = kk; k < kend; ++k) s += jac[(size_t)k * m + i] * jac[(size_t)k * m + j];
          jtj[i * m + j] += s;
        }
    }
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < i; ++j) jtj[i * m + j] = jtj[j * m + i];
    for (int i = 0; i < m; ++i) jte[i] = Real(0.0);
    for (int i = 0; i < n; ++i) {
      const Real ei = x[i] - h[i];
      for (int l = 0; l < m; ++l) jte[l] += jac[(size_t)i * m + l] * ei;
    }
  }
}

template <class Real>
__device__ void pack_sums(const Real *jtj, const Real *jte, int m, Real *out) {
  int c = 0;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j) out[c++] = jtj[i * m + j];
  for (int i = 0; i < m; ++i) out[c++] = jte[i];
}

template <class Real>
__global__ __launch_bounds__(kGenThreads) void gen_pass_kernel(GenArgs<Real> a) {
  __shared__ Real part[kGenThreads];
  __shared__ Real sh_out[kGenSums];
  const int tid = threadIdx.x;
  const int n = a.n, m = a.m;
  const int nl = m * (m + 1) / 2;

  // ---- element-wise stages, all lanes -------------------------------------------------------------------------
  if (a.kind == RQ_DIF_INIT || (a.kind == RQ_EVAL && a.store_hx)) {
    for (int i = tid; i < n; i += kGenThreads) a.hx_rw[i] = a.aux[i];
  } else if (a.kind == RQ_DIF_TRIAL) {
    for (int i = tid; i < n; i += kGenThreads) a.wrk[i] = a.aux[i];
  } else if ((a.kind == RQ_JAC || a.kind == RQ_DIF_JAC) && !a.user_jac) {  // misc_core.c:167-170 / :206-209
    for (int i = tid; i < n; i += kGenThreads)
      for (int j = 0; j < m; ++j) {
        const Real v = a.central ? (a.aux[(size_t)(2 * j + 1) * n + i] - a.aux[(size_t)(2 * j) * n + i])
                                   : (a.aux[(size_t)j * n + i] - a.hx[i]);
        a.jac[(size_t)i * m + j] = v * a.dinv[j];
      }
  } else if (a.kind == RQ_DIF_UPDATE) {  // lm_core.c:760-766
    for (int i = tid; i < n; i += kGenThreads) {
      Real *row = a.jac + (size_t)i * m;
      Real t = Real(0.0);
      for (int l = 0; l < m; ++l) t += row[l] * a.dp[l];
      t = (a.wrk[i] - a.hx[i] - t) / a.dp_l2;
      for (int j = 0; j < m; ++j) row[j] += t * a.dp[j];
    }
  }
  __syncthreads();

  const Real *y = (a.kind == RQ_EVAL || a.kind == RQ_SCALED) ? a.aux : ((a.kind == RQ_DIF_TRIAL) ? a.wrk : a.hx);
  const Real *h_for_e = (a.kind == RQ_DIF_UPDATE && a.accepted) ? a.wrk : a.hx;  // residual paired with J^T e
  const bool wants_norm = (a.kind == RQ_EVAL || a.kind == RQ_SCALED || a.kind == RQ_DIF_INIT || a.kind == RQ_DIF_TRIAL || a.kind == RQ_JAC);
  const bool wants_jtj = (a.kind == RQ_JAC || a.kind == RQ_DIF_JAC || a.kind == RQ_DIF_UPDATE);

  if (a.exact) {
    if (tid == 0) {
      Real mx = Real(0.0);
      if (wants_jtj) {
        Real jtj[kGenMaxM * kGenMaxM], jte[kGenMaxM];
        const int nm = n * m;
        const bool small = a.bc_rule ? (nm < 1024) : (nm <= 1024);
        ref_jtj_jte(a.jac, a.x, h_for_e, n, m, small, jtj, jte);
        pack_sums(jtj, jte, m, sh_out);
        if (a.kind == RQ_JAC) sh_out[nl + m] = ref_l2(a.x, a.hx, n, Real(1.0), false, &mx);
      } else if (wants_norm) {
        sh_out[0] = ref_l2(a.x, y, n, a.scal, a.kind == RQ_SCALED, &mx);
      }
      sh_out[kGenSums - 1] = mx;
    }
    __syncthreads();
  } else {
    // deterministic tree: strided per-lane accumulation, then a fixed LDS fold per sum
    const int nsum = wants_jtj ? (nl + m + (a.kind == RQ_JAC ? 1 : 0)) : 1;
    Real acc[kGenSums];
    for (int k = 0; k < kGenSums; ++k) acc[k] = Real(0.0);
    Real mx = Real(0.0);
    for (int i = tid; i < n; i += kGenThreads) {
      if (wants_jtj) {
        const Real *row = a.jac + (size_t)i * m;
        const Real e = a.x[i] - h_for_e[i];
        int c = 0;
        for (int r = 0; r < m; ++r)
          for (int j = 0; j <= r; ++j) acc[c++] += row[r] * row[j];
        for (int r = 0; r < m; ++r) acc[c++] += row[r] * e;
        if (a.kind == RQ_JAC) {
          const Real e0 = a.x[i] - a.hx[i];
          acc[c] += e0 * e0;
        }
      } else {
        Real e = a.x[i] - y[i];
        if (a.kind == RQ_SCALED) e /= a.scal;
        acc[0] += e * e;
        mx = fmax(mx, fabs(e));
      }
    }
    for (int k = 0; k <= nsum; ++k) {
      part[tid] = (k < nsum) ? acc[k] : mx;
      __syncthreads();
      if (tid == 0) {
        Real s = part[0];
        if (k < nsum)
          for (int t = 1; t < kGenThreads; ++t) s += part[t];
        else
          for (int t = 1; t < kGenThreads; ++t) s = fmax(s, part[t]);
        sh_out[(k < nsum) ? k : kGenSums - 1] = s;
      }
      __syncthreads();
    }
  }
  if (a.kind == RQ_DIF_UPDATE && a.accepted) {  // e, hx <- trial values (lm_core.c:783-786)
    for (int i = tid; i < n; i += kGenThreads) a.hx_rw[i] = a.wrk[i];
  }
  if (tid < kGenSums) a.out[tid] = sh_out[tid];
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return kLmError;                                                                \
    }                                                                                 \
  } while (0)

template <class Real>
using user_func_of = void (*)(Real *p, Real *hx, int m, int n, void *adata);
template <class Real>
using user_jacf_of = void (*)(Real *p, Real *jac, int m, int n, void *adata);
typedef user_func_of<double> user_func_t;
typedef user_jacf_of<double> user_jacf_t;

namespace {

template <class Real>
struct GenBuffers {
  Real *d = nullptr;
  size_t cap = 0;
  ~GenBuffers() {
    if (d) (void)hipFree(d);
  }
};

template <int M, int METHOD, class Real>
int generic_run(user_func_of<Real> func, user_jacf_of<Real> jacf, Real *p, Real *x, int n, Real *lb, Real *ub, Real *dscl, int itmax,
                Real *opts, Real *info, Real *covar, void *adata) {
  // METHOD: 0 dlevmar_dif, 1 dlevmar_bc_dif / dlevmar_bc_der, 2 dlevmar_der
  using Machine = typename std::conditional<METHOD == 0, DifMachine<M, Real>,
                                            typename std::conditional<METHOD == 1, BcMachine<M, Real>, DerMachine<M, Real>>::type>::type;
  Machine mach;
  if constexpr (METHOD == 0)
    mach.start(p, n, itmax, opts, covar != nullptr, /*speculative=*/0);
  else if constexpr (METHOD == 1)
    mach.start(p, n, lb, ub, dscl, itmax, opts, covar != nullptr);
  else
    mach.start(p, n, itmax, opts, covar != nullptr);
  if constexpr (METHOD == 1) mach.c.analytic_jac = jacf ? 1 : 0;
  if (mach.h.req.kind == RQ_DONE) {
    if constexpr (METHOD == 1) {
      if (mach.c.bad_input == 2) {
        set_error("dlevmar_bc_dif(): at least one lower bound exceeds the upper one");
        return kLmError;
      }
      if (mach.c.bad_input == 3) {
        set_error("dlevmar_bc_dif(): scaling constants should be positive");
        return kLmError;
      }
    }
    set_error("%clevmar_%s(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", sizeof(Real) == 4 ? 's' : 'd',
              METHOD == 0 ? "dif" : (METHOD == 1 ? "bc_dif" : "der"), n, M);
    return kLmError;
  }
  if constexpr (METHOD == 1)
    for (int i = 0; i < M; ++i)
      if (mach.c.infeasible_mask & (1 << i))
        fprintf(stderr, "Warning: component %d of starting point not feasible in dlevmar_bc_dif()! [%g projected to %g]\n", i,
                mach.c.p_start[i], mach.h.p[i]);

  // device buffers: x | hx | wrk | aux (2M planes) | jac (n*M) | out
  const size_t need = (size_t)n * (3 + 2 * M + M) + kGenSums;
  GenBuffers<Real> buf;
  HIP_OK(hipMalloc(&buf.d, need * sizeof(Real)));
  Real *d_x = buf.d, *d_hx = d_x + n, *d_wrk = d_hx + n, *d_aux = d_wrk + n, *d_jac = d_aux + (size_t)2 * M * n,
         *d_out = d_jac + (size_t)n * M;
  std::vector<Real> host_aux((size_t)2 * M * n);
  if (x) {
    HIP_OK(hipMemcpy(d_x, x, sizeof(Real) * n, hipMemcpyHostToDevice));
  } else {  // "NULL implies a zero vector", lm_core.c:441
    HIP_OK(hipMemset(d_x, 0, sizeof(Real) * n));
  }
  Real sums_host[kGenSums];
  Real sums[SumLayout<M>::MAX + 2];
  const bool exact = (long long)n * M <= kGenExactLimit;

  long long guard = 0;
  const long long cap = (long long)(itmax > 0 ? itmax : 1) * 700 + 64;
  while (mach.h.req.kind != RQ_DONE) {
    if (++guard > cap) {
      set_error("pass budget exhausted without termination");
      return kLmError;
    }
    const Request<M, Real> &r = mach.h.req;
    GenArgs<Real> a;
    memset(&a, 0, sizeof a);
    a.x = d_x;
    a.hx = d_hx;
    a.hx_rw = d_hx;
    a.aux = d_aux;
    a.wrk = d_wrk;
    a.jac = d_jac;
    a.out = d_out;
    a.n = n;
    a.m = M;
    a.kind = r.kind;
    a.central = r.central;
    a.exact = exact ? 1 : 0;
    a.accepted = r.aux;
    a.bc_rule = (METHOD != 0);  // n*m < 1024 in bc_der and der (lmbc_core.c:573, lm_core.c:197), <= 1024 in dif (:594)
    a.scal = r.scal;
    a.dp_l2 = r.dp_l2;
    for (int j = 0; j < M; ++j) a.dp[j] = r.dp[j];
    Real pt[M];
    size_t planes = 0;
    switch (r.kind) {
    case RQ_EVAL:
    case RQ_SCALED:
    case RQ_DIF_INIT:
      for (int j = 0; j < M; ++j) pt[j] = r.p[j];
      func(pt, host_aux.data(), M, n, adata);
      planes = 1;
      break;
    case RQ_DIF_TRIAL:
      for (int j = 0; j < M; ++j) pt[j] = r.q[j];
      func(pt, host_aux.data(), M, n, adata);
      planes = 1;
      break;
    case RQ_JAC:
      if (jacf) {  // dlevmar_bc_der: the caller's analytic Jacobian (lmbc_core.c:555-557).  The residual paired with
                   // J^T e is x - f(p) of the accepted point; f is deterministic, so it is simply evaluated again here
        std::vector<Real> jh((size_t)n * M);
        for (int j = 0; j < M; ++j) pt[j] = r.p[j];
        func(pt, host_aux.data(), M, n, adata);
        HIP_OK(hipMemcpy(d_hx, host_aux.data(), sizeof(Real) * n, hipMemcpyHostToDevice));
        for (int j = 0; j < M; ++j) pt[j] = r.p[j];
        jacf(pt, jh.data(), M, n, adata);
        HIP_OK(hipMemcpy(d_jac, jh.data(), sizeof(Real) * (size_t)n * M, hipMemcpyHostToDevice));
        a.user_jac = 1;
        break;
      }
      // bc_dif re-evaluates f(p) on every Jacobian (lmbc_core.c:1049); forward only
      if (!r.central) {
        for (int j = 0; j < M; ++j) pt[j] = r.p[j];
        func(pt, host_aux.data() + (size_t)M * n, M, n, adata);  // parked behind the M difference planes
        HIP_OK(hipMemcpy(d_hx, host_aux.data() + (size_t)M * n, sizeof(Real) * n, hipMemcpyHostToDevice));
      }
      [[fallthrough]];
    case RQ_DIF_JAC:
      for (int j = 0; j < M; ++j) {
        a.dinv[j] = (r.central ? Real(0.5) : Real(1.0)) / r.d[j];
        for (int k = 0; k < M; ++k) pt[k] = r.p[k];
        if (!r.central) {
          pt[j] = r.p[j] + r.d[j];
          func(pt, host_aux.data() + (size_t)j * n, M, n, adata);
        } else {
          pt[j] = r.p[j] - r.d[j];
          func(pt, host_aux.data() + (size_t)(2 * j) * n, M, n, adata);
          pt[j] = r.p[j] + r.d[j];
          func(pt, host_aux.data() + (size_t)(2 * j + 1) * n, M, n, adata);
        }
      }
      planes = r.central ? 2 * M : M;
      break;
    default: break;  // RQ_DIF_UPDATE: no evaluation
    }
    if (planes) HIP_OK(hipMemcpy(d_aux, host_aux.data(), sizeof(Real) * planes * n, hipMemcpyHostToDevice));
    // (bc with central differences never evaluates f(p): the residual paired with J^T e is the one of the last
    // accepted point, which the device keeps in hx -- refreshed below whenever an evaluation is accepted)
    a.store_hx = 0;
    hipLaunchKernelGGL(gen_pass_kernel<Real>, dim3(1), dim3(kGenThreads), 0, 0, a);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(sums_host, d_out, sizeof(Real) * kGenSums, hipMemcpyDeviceToHost));
    const int ns = (r.kind == RQ_JAC) ? SumLayout<M>::JAC : ((r.kind == RQ_DIF_JAC || r.kind == RQ_DIF_UPDATE) ? SumLayout<M>::DIF_JAC : 1);
    for (int k = 0; k < ns; ++k) sums[k] = sums_host[k];
    const int kind_done = r.kind;
    mach.step(sums, sums_host[kGenSums - 1]);
    if constexpr (METHOD == 1) if (kind_done == RQ_EVAL) {
      // bc keeps e of the last ACCEPTED point for central-difference Jacobians: mirror "e <- hx" (lmbc_core.c:779,
      // :964) by keeping f(accepted point) on the device.  The machine accepted iff its p now equals the point
      // that was just evaluated.
      bool same = true;
      for (int j = 0; j < M; ++j) {
        const Real cur = mach.c.has_dscl ? mach.h.p[j] * mach.c.dscl[j] : mach.h.p[j];
        if (cur != pt[j]) same = false;
      }
      if (same) HIP_OK(hipMemcpy(d_hx, d_aux, sizeof(Real) * n, hipMemcpyDeviceToDevice));
    }
  }
  for (int i = 0; i < M; ++i) p[i] = mach.h.p[i];
  if (info)
    for (int i = 0; i < kInfoSz; ++i) info[i] = mach.c.info[i];
  if (covar)
    for (int i = 0; i < M * M; ++i) covar[i] = mach.c.covar[i];
  return mach.c.ret;
}

template <int METHOD, class Real>
int generic_dispatch(user_func_of<Real> func, user_jacf_of<Real> jacf, Real *p, Real *x, int m, int n, Real *lb, Real *ub, Real *dscl, int itmax,
                     Real *opts, Real *info, Real *covar, void *adata) {
  switch (m) {
  case 1: return generic_run<1, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 2: return generic_run<2, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 3: return generic_run<3, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 4: return generic_run<4, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 5: return generic_run<5, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 6: return generic_run<6, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 7: return generic_run<7, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 8: return generic_run<8, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 9: return generic_run<9, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 10: return generic_run<10, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 11: return generic_run<11, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 12: return generic_run<12, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 13: return generic_run<13, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 14: return generic_run<14, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 15: return generic_run<15, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  case 16: return generic_run<16, METHOD, Real>(func, jacf, p, x, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  }
  set_error("generic callback path supports 1 <= m <= %d parameters (got %d)", kGenMaxM, m);
  return kLmError;
}

}  // namespace

int generic_fit_run(int method, user_func_t func, user_jacf_t jacf, double *p, double *x, int m, int n, double *lb, double *ub,
                    double *dscl, int itmax, double *opts, double *info, double *covar, void *adata) {
  if (!func || !p || n <= 0) {
    set_error("generic fit: null callback / parameter vector or n <= 0");
    return kLmError;
  }
  (void)hipGetLastError();
  if (method == 0) return generic_dispatch<0, double>(func, nullptr, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  if (method == 2) return generic_dispatch<2, double>(func, jacf, p, x, m, n, nullptr, nullptr, nullptr, itmax, opts, info, covar, adata);
  return generic_dispatch<1, double>(func, jacf, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata);
}

// the single-precision twins (slevmar_*, levmar.h:208-310): the same machines and kernels instantiated with Real = float
int generic_fit_run_f(int method, user_func_of<float> func, user_jacf_of<float> jacf, float *p, float *x, int m, int n, float *lb,
                      float *ub, float *dscl, int itmax, float *opts, float *info, float *covar, void *adata) {
  if (!func || !p || n <= 0) {
    set_error("generic fit: null callback / parameter vector or n <= 0");
    return kLmError;
  }
  (void)hipGetLastError();
  if (method == 0) return generic_dispatch<0, float>(func, nullptr, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata);
  if (method == 2) return generic_dispatch<2, float>(func, jacf, p, x, m, n, nullptr, nullptr, nullptr, itmax, opts, info, covar, adata);
  return generic_dispatch<1, float>(func, jacf, p, x, m, n, lb, ub, dscl, itmax, opts, info, covar, adata);
}


// dlevmar_chkjac's comparison of the n rows (misc_core.c:286-318), one row per lane
template <class Real>
__global__ __launch_bounds__(256) void chkjac_kernel(const Real *__restrict__ fvec, const Real *__restrict__ fjac,
                                                     const Real *__restrict__ fvecp, const Real *__restrict__ pabs, int m, int n,
                                                     Real *__restrict__ err) {
  const Real epsmch = LmLimits<Real>::eps(), eps = sqrt(epsmch), epsf = Real(100.0) * epsmch, epslog = log10(eps);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    Real e = Real(0.0);
    for (int j = 0; j < m; ++j) e += pabs[j] * fjac[(size_t)i * m + j];  // temp = |p[j]| (1 if zero), :290-296
    Real temp = Real(1.0);
    const Real f = fvec[i], fp = fvecp[i];
    if (f != Real(0.0) && fp != Real(0.0) && fabs(fp - f) >= epsf * fabs(f)) temp = eps * fabs((fp - f) / eps - e) / (fabs(f) + fabs(fp));
    Real r = Real(1.0);
    if (temp > epsmch && temp < eps) r = (log10(temp) - epslog) / epslog;
    if (temp >= eps) r = Real(0.0);
    err[i] = r;
  }
}

template <class Real>
int chkjac_err_run_t(const Real *fvec, const Real *fjac, const Real *fvecp, const Real *p, int m, int n, Real *err) {
  (void)hipGetLastError();
  Real *d = nullptr;
  const size_t total = (size_t)n * (m + 3) + m;
  if (hipMalloc(&d, total * sizeof(Real)) != hipSuccess) {
    set_error("dlevmar_chkjac(): hipMalloc failed");
    return kLmError;
  }
  Real *d_fvec = d, *d_fvecp = d + n, *d_err = d + 2 * (size_t)n, *d_p = d + 3 * (size_t)n, *d_fjac = d_p + m;
  std::vector<Real> pabs(m);
  for (int j = 0; j < m; ++j) {
    pabs[j] = fabs(p[j]);
    if (pabs[j] == Real(0.0)) pabs[j] = Real(1.0);
  }
  hipError_t e = hipMemcpy(d_fvec, fvec, sizeof(Real) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_fvecp, fvecp, sizeof(Real) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_fjac, fjac, sizeof(Real) * (size_t)n * m, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_p, pabs.data(), sizeof(Real) * m, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(chkjac_kernel<Real>, dim3(blocks), dim3(256), 0, nullptr, d_fvec, d_fjac, d_fvecp, d_p, m, n, d_err);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(err, d_err, sizeof(Real) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) {
    set_error("dlevmar_chkjac(): %s", hipGetErrorString(e));
    return kLmError;
  }
  return 0;
}


// dlevmar_R2's three sums (misc_core.c:634-654): sum x, then SS_err and SS_tot, each walked from the top index down.
// Small problems: one lane, the reference's order.  Large ones: per-lane strided partial sums folded by a fixed tree.
template <class Real>
__global__ __launch_bounds__(256) void r2_kernel(const Real *__restrict__ x, const Real *__restrict__ hx, int n, int exact,
                                                 Real *__restrict__ out) {
  __shared__ Real part[3][256];
  const int tid = threadIdx.x;
  if (exact) {
    if (tid == 0) {
      Real sx = Real(0.0);
      for (int i = n; i-- > 0;) sx += x[i];
      const Real xavg = sx / (Real)n;
      Real sse = Real(0.0), sst = Real(0.0);
      for (int i = n; i-- > 0;) {
        Real t = x[i] - hx[i];
        sse += t * t;
        t = x[i] - xavg;
        sst += t * t;
      }
      out[0] = Real(1.0) - sse / sst;
    }
    return;
  }
  auto fold = [&](int k) {
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) part[k][tid] += part[k][tid + s];
      __syncthreads();
    }
  };
  Real sx = Real(0.0);
  for (int i = tid; i < n; i += 256) sx += x[i];
  part[0][tid] = sx;
  fold(0);
  const Real xavg = part[0][0] / (Real)n;
  Real sse = Real(0.0), sst = Real(0.0);
  for (int i = tid; i < n; i += 256) {
    Real t = x[i] - hx[i];
    sse += t * t;
    t = x[i] - xavg;
    sst += t * t;
  }
  part[1][tid] = sse;
  part[2][tid] = sst;
  fold(1);
  fold(2);
  if (tid == 0) out[0] = Real(1.0) - part[1][0] / part[2][0];
}

template <class Real>
int r2_run_t(const Real *x, const Real *hx, int n, Real *r2) {
  (void)hipGetLastError();
  Real *d = nullptr;
  if (hipMalloc(&d, (2 * (size_t)n + 1) * sizeof(Real)) != hipSuccess) {
    set_error("dlevmar_R2(): hipMalloc failed");
    return kLmError;
  }
  hipError_t e = x ? hipMemcpy(d, x, sizeof(Real) * n, hipMemcpyHostToDevice) : hipMemset(d, 0, sizeof(Real) * n);
  if (e == hipSuccess) e = hipMemcpy(d + n, hx, sizeof(Real) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(r2_kernel<Real>, dim3(1), dim3(256), 0, nullptr, d, d + n, n, n <= kGenExactLimit ? 1 : 0, d + 2 * (size_t)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(r2, d + 2 * (size_t)n, sizeof(Real), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) {
    set_error("dlevmar_R2(): %s", hipGetErrorString(e));
    return kLmError;
  }
  return 0;
}

int chkjac_err_run(const double *fvec, const double *fjac, const double *fvecp, const double *p, int m, int n, double *err) {
  return chkjac_err_run_t<double>(fvec, fjac, fvecp, p, m, n, err);
}
int chkjac_err_run_f(const float *fvec, const float *fjac, const float *fvecp, const float *p, int m, int n, float *err) {
  return chkjac_err_run_t<float>(fvec, fjac, fvecp, p, m, n, err);
}
int r2_run(const double *x, const double *hx, int n, double *r2) { return r2_run_t<double>(x, hx, n, r2); }
int r2_run_f(const float *x, const float *hx, int n, float *r2) { return r2_run_t<float>(x, hx, n, r2); }

}  // namespace brdf
