// stream_fit.hip -- streamed regime kernels + host driver (see stream_fit.h).
//
// Replaces, for one large fit: the per-evaluation model loop (brdfdata.cpp:975-988), e=x-hx / ||e||^2
// (misc_core.c:721-807), the FD Jacobian fill (misc_core.c:153-171, :191-210), J^T J / J^T e
// (lm_core.c:617-653, misc_core.c:103-128) and the Broyden update (lm_core.c:760-766), fused into
// one HBM/L2 sweep per LM evaluation.
//
// Data layout in HBM: the caller's SoA planes c0|c1|c2 (n doubles each) and x[n] are read in place,
// 8 B per lane coalesced.  dif keeps two SoA secant Jacobians (3 planes of n each) so the trial pass can
// write the Broyden-updated Jacobian speculatively and the state machine commits by flipping an index;
// f(p) (levmar's hx vector) is re-evaluated where needed instead of being stored.  Workgroup b sweeps one contiguous tile; tiles are dealt so that the eight
// workgroups {b, b+8, ...} that share an XCD own one contiguous eighth of the arrays, i.e. every
// XCD's private L2 keeps seeing the same 1/8 slice on every pass.
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#if defined(BRDF_STAMPS)
#include <hip/hip_runtime.h>
__device__ long long g_lm_stamps[8];  // diagnostic: cycles per section of DifMachine::run, block 0 only
__device__ long long g_lm_last;
#endif
#if defined(BRDF_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define LM_STAMP(i) do { if (blockIdx.x == 0) { const long long now_ = clock64(); if ((i) != 0) g_lm_stamps[i] += now_ - g_lm_last; g_lm_last = now_; } } while (0)
#endif
#include "stream_fit.h"

namespace brdf {

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// METHOD 0 dlevmar_dif, 1 dlevmar_bc_dif / bc_der, 2 dlevmar_der (the model's analytic Jacobian, lm_core.c:64-432)
template <int METHOD>
using MachineOf = typename std::conditional<METHOD == 0, DifMachine<kM>, typename std::conditional<METHOD == 1, BcMachine<kM>, DerMachine<kM>>::type>::type;

template <int METHOD>
__device__ __forceinline__ MachineOf<METHOD> *machine_slot(StreamCtx *ctx, int which) {
  if constexpr (METHOD == 0)
    return &ctx->m[which].dif;
  else if constexpr (METHOD == 1)
    return &ctx->m[which].bc;
  else
    return &ctx->m[which].der;
}

template <int METHOD>
__device__ void publish_result(StreamCtx *ctx, const MachineOf<METHOD> &sm, int pass) {
  Mailbox *mb = ctx->mbox;
  mb->ret = sm.c.ret;
  mb->passes = pass;
  if constexpr (METHOD == 1)
    mb->infeasible_mask = sm.c.infeasible_mask;
  else
    mb->infeasible_mask = 0;
  mb->t_first = ctx->t_first;
  mb->t_last = (long long)wall_clock64();
  mb->n_jac = ctx->n_jac;
  mb->n_eval = ctx->n_eval;
  mb->domain_bad = ctx->domain_bad;
  for (int k = 0; k < 8; ++k) mb->stamps[k] = ctx->stamps[k];
  for (int i = 0; i < kM; ++i) mb->p[i] = sm.h.p[i];
  for (int i = 0; i < kInfoSz; ++i) mb->info[i] = sm.c.info[i];
  for (int i = 0; i < kM * kM; ++i) mb->covar[i] = sm.c.covar[i];
  ctx->done = 1;
  __threadfence_system();
  __hip_atomic_store(&mb->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int MODEL, int METHOD, bool FAST>
__global__ __launch_bounds__(kStreamThreads, kStreamWavesPerSimd) void stream_pass(StreamCtx *ctx, int pass) {
  using Machine = MachineOf<METHOD>;
  using Mdl = BrdfModel<MODEL>;
  __shared__ Machine sm;
  __shared__ PassUniforms<MODEL> su;
  __shared__ double red[reduce_buf_doubles<kStreamThreads>()];
  __shared__ double sums[kSlots];

  const int tid = threadIdx.x;
  const int par = pass & 1;
  // The scalar LM step is executed by the WHOLE first wave behind a wave-uniform branch (all 64 lanes compute and
  // store identical values): inside a branch the compiler knows to be uniform, the step's own control flow is
  // compiled to scalar branches; under `if (tid == 0)` it is structurised into exec-mask form and runs ~25 % slower.
  const bool first_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < kWave;
#ifdef BRDF_STAMPS
  long long st_[8]; int sti_ = 0;
#define STAMP() do { if (blockIdx.x == 0 && tid == 0) st_[sti_++] = clock64(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  STAMP();

  // Everything this launch needs from the previous one is requested up front, in one burst, so that a
  // single global-memory round trip is paid: the sticky `done` word, the machine state and this thread's
  // column of the previous launch's per-workgroup partial rows (written by other CUs: they miss in L2).
  static_assert(sizeof(Machine) % 4 == 0 && sizeof(Machine) / 4 <= 2 * kStreamThreads, "machine copy layout");
  const int done = ctx->done;
  const unsigned *msrc = reinterpret_cast<const unsigned *>(machine_slot<METHOD>(ctx, par));
  constexpr int kWords = (int)(sizeof(Machine) / 4);
  const unsigned w0 = (tid < kWords) ? msrc[tid] : 0u;
  const unsigned w1 = (tid + kStreamThreads < kWords) ? msrc[tid + kStreamThreads] : 0u;
  double pv[kSlots];
  {
    const double *part = ctx->partials + (size_t)par * kSlots * kStreamMaxBlocks;
    const bool have = pass > 0 && tid < ctx->nb;
#pragma unroll
    for (int k = 0; k < kSlots; ++k) pv[k] = have ? part[k * kStreamMaxBlocks + tid] : 0.0;
  }
  if (done) return;  // the fit finished in an earlier launch: queued run-ahead launches fall through
  {
    unsigned *dst = reinterpret_cast<unsigned *>(&sm);
    if (tid < kWords) dst[tid] = w0;
    if (tid + kStreamThreads < kWords) dst[tid + kStreamThreads] = w1;
  }
  __syncthreads();
  STAMP();

  if (pass > 0) {  // fold the previous launch's per-workgroup partials and advance the LM state machine
    switch (sm.h.req.kind) {
    case RQ_JAC: block_reduce<SumLayout<kM>::JAC, kStreamThreads>(pv, pv[kSums], red, sums); break;
    case RQ_DIF_JAC: block_reduce<SumLayout<kM>::DIF_JAC, kStreamThreads>(pv, pv[kSums], red, sums); break;
    case RQ_DIF_TRIAL: block_reduce<SumLayout<kM>::DIF_TRIAL, kStreamThreads>(pv, pv[kSums], red, sums); break;
    case RQ_EVAL_MULTI: block_reduce<kMaxCand, kStreamThreads>(pv, pv[kSums], red, sums); break;
    default: block_reduce<1, kStreamThreads>(pv, pv[kSums], red, sums); break;
    }
    STAMP();
#ifdef BRDF_STAMPS
    const int kind_before_ = sm.h.req.kind, phase_before_ = sm.h.phase;
    const long long ts0_ = clock64();
#endif
    if (first_wave) {  // (dif: + chains of rejections sharing a sweep; bc: + candidates evaluated by Jacobian passes -- lm_machine.h)
      if constexpr (METHOD == 0)
        sm.template step<true, true>(sums, sums[kSums]);
      else if constexpr (METHOD == 1)
        sm.template step<true, true, false, true>(sums, sums[kSums]);
      else
        sm.template step<true>(sums, sums[kSums]);
    }
#ifdef BRDF_STAMPS
    if (blockIdx.x == 0 && tid == 0 && pass < 4096) {
      ctx->dbg[pass * 4 + 0] = (int)(clock64() - ts0_);
      ctx->dbg[pass * 4 + 1] = kind_before_ * 100 + sm.h.req.kind;
      ctx->dbg[pass * 4 + 2] = phase_before_ * 100 + sm.h.phase;
    }
#endif
    __syncthreads();
    STAMP();
  }

  const int kind = sm.h.req.kind;
  if (kind == RQ_DONE) {
    if (blockIdx.x == 0 && tid == 0) publish_result<METHOD>(ctx, sm, pass);
    return;
  }
  if (first_wave) {
    if constexpr (METHOD == 1)
      su.build(sm.h.req, true, sm.c.analytic_jac != 0);
    else
      su.build(sm.h.req, true, METHOD == 2);
  }
  STAMP();
  if (blockIdx.x == 0) {  // persist the advanced machine for the next launch
    const unsigned *src = reinterpret_cast<const unsigned *>(&sm);
    unsigned *dst = reinterpret_cast<unsigned *>(machine_slot<METHOD>(ctx, par ^ 1));
    for (int w = tid; w < (int)(sizeof(Machine) / 4); w += kStreamThreads) dst[w] = src[w];
    if (tid == 0) {
      if (pass == 0) ctx->t_first = (long long)wall_clock64();
      if (kind == RQ_JAC || kind == RQ_DIF_JAC)
        ctx->n_jac += 1;
      else
        ctx->n_eval += 1;
      __hip_atomic_store(&ctx->mbox->progress, pass + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  STAMP();

  // ---- the sweep -----------------------------------------------------------------------------------
  const int n = ctx->n;
  const int nb = gridDim.x;
  int vb = blockIdx.x;  // workgroups b, b+8, .. share an XCD: give each XCD a contiguous run of tiles
  if ((nb & 7) == 0) vb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  const int tile = (n + nb - 1) / nb;
  const int begin = vb * tile;
  const int end = min(n, begin + tile);
  const double *__restrict__ c0 = ctx->c0;
  const double *__restrict__ c1 = ctx->c1;
  const double *__restrict__ c2 = ctx->c2;
  const double *__restrict__ x = ctx->x;
  const PassUniforms<MODEL> u = su;

  double acc[kSums];
#pragma unroll
  for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
  double mx = 0.0;
  double *outp = ctx->partials + (size_t)(par ^ 1) * kSlots * kStreamMaxBlocks;

  // Samples are taken U at a time per lane: all loads of a batch are issued before the first
  // transcendental so that one memory round trip is paid per batch, not per sample.
  // FAST: the first pass of a fit (pass 0) reads the raw cosine planes, derives the per-sample
  // invariants (brdf_models.h: Prep) and parks them in two scratch planes; every later pass reads those.
  double *__restrict__ q1p = ctx->prep[0];
  double *__restrict__ q2p = ctx->prep[1];
  const bool first = (pass == 0);
#define SWEEP_BEGIN(U)                                                                        \
  for (int base = begin + tid; base < end; base += (U) * kStreamThreads) {                    \
    int idx[U];                                                                               \
    bool ok[U];                                                                               \
    double s0[U], sx[U];                                                                      \
    Prep pq[U];                                                                               \
    _Pragma("unroll") for (int k = 0; k < (U); ++k) {                                         \
      const int i = base + k * kStreamThreads;                                                \
      ok[k] = i < end;                                                                        \
      idx[k] = ok[k] ? i : begin;                                                             \
      s0[k] = c0[idx[k]];                                                                     \
      sx[k] = x[idx[k]];                                                                      \
    }                                                                                         \
    if (!FAST || first) {                                                                     \
      double r1[U], r2[U];                                                                    \
      _Pragma("unroll") for (int k = 0; k < (U); ++k) {                                       \
        r1[k] = Mdl::uses_c1 ? c1[idx[k]] : 0.0;                                              \
        r2[k] = Mdl::uses_c2 ? c2[idx[k]] : 0.0;                                              \
      }                                                                                       \
      _Pragma("unroll") for (int k = 0; k < (U); ++k) {                                       \
        pq[k] = Mdl::template prepare<FAST>(s0[k], r1[k], r2[k]);                             \
        if (FAST && ok[k]) {                                                                  \
          q1p[idx[k]] = pq[k].q1;                                                             \
          if (Mdl::prep_planes > 1) q2p[idx[k]] = pq[k].q2;                                   \
          if (!Mdl::domain_ok(s0[k], r1[k], r2[k])) ctx->domain_bad = 1;                      \
        }                                                                                     \
      }                                                                                       \
    } else {                                                                                  \
      _Pragma("unroll") for (int k = 0; k < (U); ++k) {                                       \
        pq[k].q1 = q1p[idx[k]];                                                               \
        pq[k].q2 = (Mdl::prep_planes > 1) ? q2p[idx[k]] : 0.0;                                \
      }                                                                                       \
    }
#define SWEEP_END }

  switch (kind) {
  case RQ_EVAL:
    SWEEP_BEGIN(4)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
      const double e = ok[k] ? sx[k] - f : 0.0;
      acc[0] += e * e;
      mx = fmax(mx, fabs(e));
    }
    SWEEP_END
    STAMP();
    block_reduce<1, kStreamThreads>(acc, mx, red, sums);
    break;
  case RQ_EVAL_MULTI:  // several projected-gradient candidates share one sweep (lm_machine.h, BcMachine::Cold::multi)
    SWEEP_BEGIN(4)
    if (u.ncand == kMaxCand) {  // the usual case: no guard between a sample's candidates, four independent exp chains per basic
                                // block (resident_fit_impl.h, RQ_EVAL_MULTI)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j0 = 0; j0 < kMaxCand; j0 += 4) {
          double e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = sx[k] - model_value_k<MODEL, FAST>(u, j0 + j, s0[k], pq[k]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!ok[k]) e[j] = 0.0;
            acc[j0 + j] += e[j] * e[j];
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
          if (j < u.ncand) {
            const double e = ok[k] ? sx[k] - model_value_k<MODEL, FAST>(u, j, s0[k], pq[k]) : 0.0;
            acc[j] += e * e;
          }
      }
    }
    SWEEP_END
    STAMP();
    block_reduce<kMaxCand, kStreamThreads>(acc, mx, red, sums);
    break;
  case RQ_SCALED:
    SWEEP_BEGIN(4)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
      const double t = ok[k] ? (sx[k] - f) / u.scal : 0.0;
      acc[0] += t * t;
    }
    SWEEP_END
    STAMP();
    block_reduce<1, kStreamThreads>(acc, mx, red, sums);
    break;
  case RQ_JAC:
    SWEEP_BEGIN(4)
    {  // (the row's kind is chosen per batch of four samples, outside their bodies: straight-line bodies, see resident_fit_impl.h RQ_JAC)
      auto rows = [&](auto jk) {
        constexpr int JK = decltype(jk)::value;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double f0 = 0.0, j[kM];
          if (JK == 2)  // dlevmar_bc_der: the caller's jacf is the model's own analytic Jacobian (lmbc_core.c:578)
            model_an_row<MODEL, FAST>(u, s0[k], pq[k], f0, j);
          else if (JK == 1)
            model_fd_row_t<MODEL, FAST, true>(u, s0[k], pq[k], true, f0, 0.0, false, j);
          else
            model_fd_row_t<MODEL, FAST, false>(u, s0[k], pq[k], true, f0, 0.0, false, j);
          double e = sx[k] - f0;
          if (!ok[k]) e = j[0] = j[1] = j[2] = 0.0;
          acc_normal_eq(j, e, acc, acc + kNL);
          acc[kNL + kM] += e * e;
        }
      };
      if (u.analytic)
        rows(std::integral_constant<int, 2>{});
      else if (u.central)
        rows(std::integral_constant<int, 1>{});
      else
        rows(std::integral_constant<int, 0>{});
    }
    SWEEP_END
    STAMP();
    block_reduce<SumLayout<kM>::JAC, kStreamThreads>(acc, mx, red, sums);
    break;
  case RQ_DIF_INIT: {
    // f(p) is never stored: wherever dlevmar_dif reads its hx vector (lm_core.c:580, :763, :785) the pass
    // re-evaluates it from the same p with the same code -- identical bits, one exp instead of 16 B of traffic
    SWEEP_BEGIN(4)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
      const double e = ok[k] ? sx[k] - f : 0.0;
      acc[0] += e * e;
    }
    SWEEP_END
    STAMP();
    block_reduce<1, kStreamThreads>(acc, mx, red, sums);
    break;
  }
  case RQ_DIF_JAC: {
    double *__restrict__ jb = ctx->jac[sm.h.req.sel_j];
    SWEEP_BEGIN(4)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double f0 = 0.0, j[kM];
      model_fd_row<MODEL, FAST>(u, s0[k], pq[k], true, f0, 0.0, false, j);
      double e = sx[k] - f0;
      if (ok[k]) {
        jb[idx[k]] = j[0];
        jb[(size_t)n + idx[k]] = j[1];
        jb[2 * (size_t)n + idx[k]] = j[2];
      } else {
        e = j[0] = j[1] = j[2] = 0.0;
      }
      acc_normal_eq(j, e, acc, acc + kNL);
    }
    SWEEP_END
    STAMP();
    block_reduce<SumLayout<kM>::DIF_JAC, kStreamThreads>(acc, mx, red, sums);
    break;
  }
  case RQ_DIF_TRIAL: {
    const double *__restrict__ jo = ctx->jac[sm.h.req.sel_j];
    double *__restrict__ jn = ctx->jac[sm.h.req.sel_j ^ 1];
    SWEEP_BEGIN(4)
    double h[4], jold[4][kM];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      jold[k][0] = jo[idx[k]];
      jold[k][1] = jo[(size_t)n + idx[k]];
      jold[k][2] = jo[2 * (size_t)n + idx[k]];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double w = model_value_q<MODEL, FAST>(u, s0[k], pq[k]);
      h[k] = model_value<MODEL, FAST>(u, s0[k], pq[k]);  // f(p), see RQ_DIF_INIT
      double j[kM];
      broyden_row(jold[k], w, h[k], u.dp, u.dp_l2, j);
      double en = sx[k] - w, eo = sx[k] - h[k];
      if (ok[k]) {
        jn[idx[k]] = j[0];
        jn[(size_t)n + idx[k]] = j[1];
        jn[2 * (size_t)n + idx[k]] = j[2];
      } else {
        en = eo = j[0] = j[1] = j[2] = 0.0;
      }
      acc[0] += en * en;
      acc_normal_eq(j, en, acc + 1, acc + 1 + kNL);
      acc[1 + kNL + kM + 0] += j[0] * eo;
      acc[1 + kNL + kM + 1] += j[1] * eo;
      acc[1 + kNL + kM + 2] += j[2] * eo;
    }
    SWEEP_END
    STAMP();
    block_reduce<SumLayout<kM>::DIF_TRIAL, kStreamThreads>(acc, mx, red, sums);
    break;
  }
  default: return;
  }
#undef SWEEP_BEGIN
#undef SWEEP_END

  if (tid < kSlots) outp[tid * kStreamMaxBlocks + blockIdx.x] = sums[tid];
  STAMP();
#ifdef BRDF_STAMPS
  if (blockIdx.x == 0 && tid == 0 && pass > 0 && sti_ == 8)
    for (int k = 0; k < 7; ++k) ctx->stamps[k] += st_[k + 1] - st_[k];
#endif
}

// K1 alone: hx[i] = f(p; sample i)  (BRDFFunc, brdfdata.cpp:969-989)
template <int MODEL>
__global__ __launch_bounds__(256) void model_eval_kernel(const double *__restrict__ c0, const double *__restrict__ c1,
                                                         const double *__restrict__ c2, int n, Request<kM> r,
                                                         double *__restrict__ hx) {
  using Mdl = BrdfModel<MODEL>;
  PassUniforms<MODEL> u;
  u.build(r);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    hx[i] = model_value<MODEL, false>(u, c0[i], Mdl::template prepare<false>(c0[i], Mdl::uses_c1 ? c1[i] : 0.0, Mdl::uses_c2 ? c2[i] : 0.0));
}

// the models' analytic Jacobian alone, n x 3 row-major like every levmar jacf (BRDFJac_hip)
template <int MODEL>
__global__ __launch_bounds__(256) void model_jac_kernel(const double *__restrict__ c0, const double *__restrict__ c1,
                                                        const double *__restrict__ c2, int n, Request<kM> r,
                                                        double *__restrict__ jac) {
  using Mdl = BrdfModel<MODEL>;
  PassUniforms<MODEL> u;
  u.build(r, true, true);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double f0, j[kM];
    model_an_row<MODEL, false>(u, c0[i], Mdl::template prepare<false>(c0[i], Mdl::uses_c1 ? c1[i] : 0.0, Mdl::uses_c2 ? c2[i] : 0.0), f0, j);
    jac[3 * (size_t)i] = j[0];
    jac[3 * (size_t)i + 1] = j[1];
    jac[3 * (size_t)i + 2] = j[2];
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  fprintf(stderr, "libbrdf_hip: %s\n", g_err);
}
const char *get_error() { return g_err; }

#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return kLmError;                                                                \
    }                                                                                 \
  } while (0)

namespace {

struct Workspace {
  int device = -1;
  StreamCtx *d_ctx = nullptr;
  StreamCtx *h_ctx = nullptr;  // pinned staging
  Mailbox *h_mbox = nullptr;   // pinned + mapped
  Mailbox *d_mbox = nullptr;
  double *d_partials = nullptr;
  double *d_dif = nullptr;  // per-sample scratch: 2n prepared planes + (dif) 6n = two SoA secant Jacobians
  size_t dif_cap = 0;
  hipStream_t last_stream = nullptr;
  bool used = false;
  FitStats stats{};

  int ensure(int dev) {
    if (device == dev && d_ctx) return 0;
    release();
    device = dev;
    HIP_OK(hipMalloc(&d_ctx, sizeof(StreamCtx)));
    HIP_OK(hipHostMalloc(&h_ctx, sizeof(StreamCtx), hipHostMallocDefault));
    HIP_OK(hipHostMalloc(&h_mbox, sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void **)&d_mbox, h_mbox, 0));
    HIP_OK(hipMalloc(&d_partials, sizeof(double) * 2 * kSlots * kStreamMaxBlocks));
    HIP_OK(hipMemset(d_partials, 0, sizeof(double) * 2 * kSlots * kStreamMaxBlocks));
    return 0;
  }
  int ensure_dif(size_t n) {
    if (dif_cap >= n) return 0;
    if (d_dif) (void)hipFree(d_dif);
    d_dif = nullptr;
    dif_cap = 0;
    HIP_OK(hipMalloc(&d_dif, sizeof(double) * 8 * n));
    dif_cap = n;
    return 0;
  }
  void release() {
    if (d_ctx) (void)hipFree(d_ctx);
    if (h_ctx) (void)hipHostFree(h_ctx);
    if (h_mbox) (void)hipHostFree(h_mbox);
    if (d_partials) (void)hipFree(d_partials);
    if (d_dif) (void)hipFree(d_dif);
    d_ctx = nullptr;
    h_ctx = nullptr;
    h_mbox = nullptr;
    d_mbox = nullptr;
    d_partials = nullptr;
    d_dif = nullptr;
    dif_cap = 0;
  }
};
thread_local Workspace g_ws;

using PassFn = void (*)(StreamCtx *, int);
PassFn pass_kernel(int model, int method, bool fast) {
  static const PassFn table[2][MODEL_COUNT][3] = {
      {{stream_pass<0, 0, false>, stream_pass<0, 1, false>, stream_pass<0, 2, false>},
       {stream_pass<1, 0, false>, stream_pass<1, 1, false>, stream_pass<1, 2, false>},
       {stream_pass<2, 0, false>, stream_pass<2, 1, false>, stream_pass<2, 2, false>}},
      {{stream_pass<0, 0, true>, stream_pass<0, 1, true>, stream_pass<0, 2, true>},
       {stream_pass<1, 0, true>, stream_pass<1, 1, true>, stream_pass<1, 2, true>},
       {stream_pass<2, 0, true>, stream_pass<2, 1, true>, stream_pass<2, 2, true>}},
  };
  return table[fast ? 1 : 0][model][method];
}

int blocks_for(int n) {
  // ~4 samples per lane; at most 2 workgroups per CU; a multiple of 8 so that the XCD dealing applies
  long long nb = ((long long)n + kStreamThreads * 4 - 1) / (kStreamThreads * 4);
  if (nb > kStreamGridCap) nb = kStreamGridCap;
  if (nb >= 8) nb &= ~7LL;
  if (nb < 1) nb = 1;
  return (int)nb;
}

}  // namespace

static thread_local bool g_last_was_resident = false;
FitStats stream_fit_last_stats() { return g_last_was_resident ? resident_fit_last_stats() : g_ws.stats; }
bool brdf_fast_path_enabled();
int pg_candidates();

// one attempt on the FAST (prepared-sample) or the exact model path; *retry_exact is set when the FAST path
// met a cosine <= 0 and the result must be discarded
static int stream_fit_attempt(const StreamFitArgs &a, bool fast, bool *retry_exact) {
  *retry_exact = false;
  Workspace &ws = g_ws;
  StreamCtx &h = *ws.h_ctx;
  // the previous call may still have run-ahead launches queued that read d_ctx; the upload below is
  // stream-ordered behind them, but the pinned staging copy must not change under an in-flight copy
  HIP_OK(hipStreamSynchronize(a.stream));
  memset(&h, 0, sizeof h);
  h.c0 = a.d_angles;
  h.c1 = a.d_angles + a.n;
  h.c2 = a.d_angles + 2 * (size_t)a.n;
  h.x = a.d_x;
  h.partials = ws.d_partials;
  h.mbox = ws.d_mbox;
  h.n = a.n;
  h.nb = blocks_for(a.n);
  h.method = a.method;
  h.model = a.model;
  if (ws.ensure_dif((size_t)a.n) != 0) return kLmError;
  h.prep[0] = ws.d_dif;
  h.prep[1] = ws.d_dif + (size_t)a.n;
  h.jac[0] = ws.d_dif + 2 * (size_t)a.n;
  h.jac[1] = ws.d_dif + 5 * (size_t)a.n;

  if (a.method == 0) {
    DifMachine<kM> &m = h.m[0].dif;
    m.start(a.p, a.n, a.itmax, a.opts, a.covar != nullptr, /*speculative=*/1, dif_chain_candidates());
    if (m.h.req.kind == RQ_DONE) {
      set_error("dlevmar_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM);
      return kLmError;
    }
  } else if (a.method == 2) {
    DerMachine<kM> &m = h.m[0].der;
    m.start(a.p, a.n, a.itmax, a.opts, a.covar != nullptr);
    if (m.h.req.kind == RQ_DONE) {
      set_error("dlevmar_der(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM);
      return kLmError;
    }
  } else {
    BcMachine<kM> &m = h.m[0].bc;
    m.start(a.p, a.n, a.lb, a.ub, a.dscl, a.itmax, a.opts, a.covar != nullptr, pg_candidates(), bc_spec_jac_enabled() ? 1 : 0);
    m.c.analytic_jac = a.analytic ? 1 : 0;
    if (m.h.req.kind == RQ_DONE) {
      switch (m.c.bad_input) {
      case 1: set_error("dlevmar_bc_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM); break;
      case 2: set_error("dlevmar_bc_dif(): at least one lower bound exceeds the upper one"); break;
      default: set_error("dlevmar_bc_dif(): scaling constants should be positive"); break;
      }
      return kLmError;
    }
    if (fast || !brdf_fast_path_enabled())  // (an exact re-run must not print the warning twice)
      for (int i = 0; i < kM; ++i)          // same warning as lmbc_core.c:516-520
        if (m.c.infeasible_mask & (1 << i))
          fprintf(stderr, "Warning: component %d of starting point not feasible in dlevmar_bc_dif()! [%g projected to %g]\n",
                  i, m.c.p_start[i], m.h.p[i]);
  }

  Mailbox &mb = *ws.h_mbox;
  memset(&mb, 0, sizeof mb);
  HIP_OK(hipMemcpyAsync(ws.d_ctx, &h, sizeof h, hipMemcpyHostToDevice, a.stream));

  const PassFn fn = pass_kernel(a.model, a.method, fast);
  const dim3 grid(h.nb), block(kStreamThreads);
  constexpr int kRunAhead = 6;
  const long long cap = (long long)(a.itmax > 0 ? a.itmax : 1) * 700 + 64;  // LM + <=150 LS + ~400 PG evals per iteration
  volatile int *done = &mb.done;
  volatile int *progress = &mb.progress;
  long long pass = 0;
  const auto t0 = std::chrono::steady_clock::now();
  long long spins = 0;
  while (!*done) {
    if (pass - (long long)*progress < kRunAhead) {
      if (pass >= cap) {
        set_error("pass budget exhausted (%lld launches) without termination", pass);
        (void)hipStreamSynchronize(a.stream);
        return kLmError;
      }
      hipLaunchKernelGGL(fn, grid, block, 0, a.stream, ws.d_ctx, (int)pass);
      ++pass;
      if ((pass & 63) == 0) HIP_OK(hipGetLastError());
    } else if ((++spins & 0xFFFF) == 0) {
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (sec > 120.0) {
        set_error("device did not finish within 120 s (pass %lld, progress %d)", pass, *progress);
        return kLmError;
      }
      HIP_OK(hipGetLastError());
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  HIP_OK(hipGetLastError());
  if (fast && mb.domain_bad) {
    *retry_exact = true;
    return 0;
  }

  for (int i = 0; i < kM; ++i) a.p[i] = mb.p[i];
  if (a.info)
    for (int i = 0; i < kInfoSz; ++i) a.info[i] = mb.info[i];
  if (a.covar)
    for (int i = 0; i < kM * kM; ++i) a.covar[i] = mb.covar[i];
  ws.stats.passes = mb.passes;
  ws.stats.launches = pass;
  ws.stats.jac_passes = mb.n_jac;
  ws.stats.eval_passes = mb.n_eval;
  ws.stats.device_us = (double)(mb.t_last - mb.t_first) / 100.0;  // s_memrealtime ticks at 100 MHz
  ws.stats.kernel_us = -1.0;  // (a chain of launches with host decisions in between: timed by its callers' events)
  for (int k = 0; k < 8; ++k) ws.stats.stamps[k] = mb.stamps[k];
#ifdef BRDF_STAMPS
  if (const char *path = getenv("BRDF_HIP_STEP_DUMP")) {  // diagnostic: per-pass cost of the LM step by transition
    static StreamCtx tmp;
    (void)hipStreamSynchronize(a.stream);
    (void)hipMemcpy(&tmp, ws.d_ctx, sizeof tmp, hipMemcpyDeviceToHost);
    long long ls[8];
    if (hipMemcpyFromSymbol(ls, HIP_SYMBOL(g_lm_stamps), sizeof ls) == hipSuccess) {
      fprintf(stderr, "lm step sections (cycles, summed over %d passes, cumulative since load): after_trial=%lld top+gradient=%lld lu_solve=%lld rest_of_solve=%lld\n",
              mb.passes, ls[1], ls[2], ls[3], ls[4]);
    }
    if (FILE *f = fopen(path, "a")) {
      for (int i = 1; i < mb.passes && i < 4096; ++i) fprintf(f, "%d %d %d %d\n", a.method, tmp.dbg[i * 4], tmp.dbg[i * 4 + 1], tmp.dbg[i * 4 + 2]);
      fclose(f);
    }
  }
#endif
  return mb.ret;
}

// BRDF_HIP_PG_MULTI=k: candidates per sweep in bc_dif's projected-gradient search (default 8, 1 = one at a time)
int pg_candidates() {
  const char *e = getenv("BRDF_HIP_PG_MULTI");
  const int k = e ? atoi(e) : kMaxCand;
  return k < 1 ? 1 : (k > kMaxCand ? kMaxCand : k);
}

// BRDF_HIP_DIF_CHAIN=k: dlevmar_dif trial points per sweep in a chain of rejections (lm_machine.h: DifMachine::Cold::multi;
// default 8, 1 = one at a time)
static std::atomic<int> g_launch_timing{0};
bool launch_timing_enabled() { return g_launch_timing.load(std::memory_order_relaxed) != 0; }
void set_launch_timing(bool on) { g_launch_timing.store(on ? 1 : 0, std::memory_order_relaxed); }

int dif_chain_candidates() {
  const char *e = getenv("BRDF_HIP_DIF_CHAIN");
  const int k = e ? atoi(e) : kMaxCand;
  return k < 1 ? 1 : (k > kMaxCand ? kMaxCand : k);
}

// BRDF_HIP_SPEC_JAC=0: single dlevmar_bc_dif / bc_der fits evaluate their candidates by plain evaluation passes (default: by
// Jacobian passes, lm_machine.h: BcMachine::Cold::spec_jac)
bool bc_spec_jac_enabled() {
  const char *e = getenv("BRDF_HIP_SPEC_JAC");
  return !(e && e[0] == '0');
}

// BRDF_HIP_EXACT_POW=1 forces the exact model path (reference expression, pow per evaluation)
bool brdf_fast_path_enabled() {
  const char *e = getenv("BRDF_HIP_EXACT_POW");
  return !(e && e[0] == '1');
}

int stream_fit_run(const StreamFitArgs &a) {
  if (a.model < 0 || a.model >= MODEL_COUNT) {
    set_error("unknown BRDF model %d (0 Phong, 1 Blinn-Phong, 2 Ward)", a.model);
    return kLmError;
  }
  if (a.method != 0 && a.method != 1 && a.method != 2) {
    set_error("unknown method %d (0 dlevmar_dif, 1 dlevmar_bc_dif / bc_der, 2 dlevmar_der)", a.method);
    return kLmError;
  }
  if (!a.p || !a.d_angles || a.n <= 0) {
    set_error("null parameter vector / sample planes, or n <= 0");
    return kLmError;
  }
  if (!a.d_x) {
    set_error("x == NULL (zero measurement vector) is not supported on the device path");
    return kLmError;
  }
  (void)hipGetLastError();  // drop any stale sticky error left by unrelated runtime calls on this thread
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  Workspace &ws = g_ws;
  if (ws.ensure(dev) != 0) return kLmError;
  if (ws.used && ws.last_stream != a.stream) HIP_OK(hipStreamSynchronize(ws.last_stream));
  ws.last_stream = a.stream;
  ws.used = true;

  double p_keep[kM];
  for (int i = 0; i < kM; ++i) p_keep[i] = a.p[i];
  {  // fits that fit the chip's registers + LDS run as ONE launch with the samples resident (resident_fit.hip)
    int pret = 0;
    if (resident_fit_try(a, &pret)) {
      g_last_was_resident = true;
      return pret;
    }
    for (int i = 0; i < kM; ++i) a.p[i] = p_keep[i];
    g_last_was_resident = false;
  }
  bool retry = false;
  int ret = stream_fit_attempt(a, brdf_fast_path_enabled(), &retry);
  if (retry) {  // a cosine <= 0 on the cached-log path: redo with the reference's pow expression
    for (int i = 0; i < kM; ++i) a.p[i] = p_keep[i];
    ret = stream_fit_attempt(a, false, &retry);
  }
  return ret;
}

int model_jac_run(int model, const double *d_angles, int n, const double *p, double *d_jac, hipStream_t stream) {
  if (model < 0 || model >= MODEL_COUNT || !d_angles || !d_jac || !p || n <= 0) {
    set_error("model_jac: bad arguments");
    return kLmError;
  }
  (void)hipGetLastError();
  Request<kM> r;
  memset(&r, 0, sizeof r);
  r.kind = RQ_JAC;
  r.scal = 1.0;
  for (int i = 0; i < kM; ++i) r.p[i] = p[i];
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const double *c0 = d_angles, *c1 = d_angles + n, *c2 = d_angles + 2 * (size_t)n;
  switch (model) {
  case 0: hipLaunchKernelGGL(model_jac_kernel<0>, dim3(blocks), dim3(256), 0, stream, c0, c1, c2, n, r, d_jac); break;
  case 1: hipLaunchKernelGGL(model_jac_kernel<1>, dim3(blocks), dim3(256), 0, stream, c0, c1, c2, n, r, d_jac); break;
  default: hipLaunchKernelGGL(model_jac_kernel<2>, dim3(blocks), dim3(256), 0, stream, c0, c1, c2, n, r, d_jac); break;
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int model_eval_run(int model, const double *d_angles, int n, const double *p, double *d_hx, hipStream_t stream) {
  if (model < 0 || model >= MODEL_COUNT || !d_angles || !d_hx || !p || n <= 0) {
    set_error("model_eval: bad arguments");
    return kLmError;
  }
  (void)hipGetLastError();
  Request<kM> r;
  memset(&r, 0, sizeof r);
  r.kind = RQ_EVAL;
  r.scal = 1.0;
  for (int i = 0; i < kM; ++i) r.p[i] = p[i];
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const double *c0 = d_angles, *c1 = d_angles + n, *c2 = d_angles + 2 * (size_t)n;
  switch (model) {
  case 0: hipLaunchKernelGGL(model_eval_kernel<0>, dim3(blocks), dim3(256), 0, stream, c0, c1, c2, n, r, d_hx); break;
  case 1: hipLaunchKernelGGL(model_eval_kernel<1>, dim3(blocks), dim3(256), 0, stream, c0, c1, c2, n, r, d_hx); break;
  default: hipLaunchKernelGGL(model_eval_kernel<2>, dim3(blocks), dim3(256), 0, stream, c0, c1, c2, n, r, d_hx); break;
  }
  HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace brdf
