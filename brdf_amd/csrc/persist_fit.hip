// persist_fit.hip -- streamed regime, persistent variant: ONE launch per fit.
//
// For fits that fit the chip's register file (n <= #CUs * 4096 samples, i.e. 1,048,576 on MI355X) the whole
// fit runs inside one launch of #CUs workgroups (one per CU, all co-resident).  Every workgroup reads its
// tile of the sample planes from HBM exactly once and keeps it -- together with the per-sample invariants,
// dlevmar_dif's f(p) and the secant Jacobian rows (lm_core.c:759-769) -- in REGISTERS for the rest of the fit:
// a pass moves no sample bytes at all.  Passes are separated by an in-launch hand-off instead of a kernel
// boundary:
//
//   every workgroup : sweep -> workgroup reduction -> its 14 partial sums as ONE 128-B line (write-through)
//                     -> drain -> ticket = atomicAdd(arrive)
//   last arriver    : loads all lines + the LM machine, folds them in a fixed order, steps the machine
//                     (lm_machine.h), builds the next pass's uniforms, publishes machine + request record
//                     (write-through, drained) and then the generation flag
//   the others      : one lane polls the generation flag (relaxed, L1-bypassing), workgroup barrier, then the
//                     request record is read with L1-bypassing loads
//
// This is the flag/counter hand-off of MI355X_MICROARCH.md ("Valid forms", first table row): every handed-off
// byte is stored sc1 by the producing wave, which drains (s_waitcnt vmcnt(0)) before ONE lane signals for the
// workgroup behind a workgroup barrier; every consumer load of those bytes is an sc1 load issued after the
// polling lane has seen the signal and the workgroup barrier that lane then joins.  No grid-wide barrier
// object, no fences; results do not depend on dispatch order or XCD placement (the fold order is by
// workgroup index).  Every spin is bounded by a wall-clock budget: if the grid is not co-resident (or anything
// else goes wrong) all workgroups drain, the launch ends with mailbox.error set and the host falls back to the
// launch-chain path of stream_fit.hip.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "stream_fit.h"

namespace brdf {

constexpr int kPThreads = 512;
constexpr int kPSpt = 8;                      // samples per lane held in registers
constexpr int kPTile = kPThreads * kPSpt;     // 4096 samples per workgroup
constexpr int kLine = 16;                     // doubles per 128-B line
constexpr long long kSpinBudgetTicks = 300000000LL;  // 3 s of s_memrealtime (100 MHz) per wait

typedef unsigned long long u64;

template <int MODEL>
struct PersistReq {  // what a pass needs to know; published by the last arriver, 8-byte words
  int kind, aux;
  int sel_hx, sel_j;
  PassUniforms<MODEL> u;
};

struct PersistCtl {  // one 128-B line per word that is polled or added to
  unsigned arrive;
  unsigned pad0[31];
  unsigned gen;
  unsigned pad1[31];
  unsigned abort;
  unsigned domain_bad;
  unsigned pad2[30];
};

struct PersistCtx {
  const double *c0, *c1, *c2, *x;
  double *partials;  // [2][G][kLine]
  PersistCtl *ctl;
  void *req;      // PersistReq<MODEL>[2], each padded to a multiple of 128 B
  void *machine;  // DifMachine<3> or BcMachine<3>
  Mailbox *mbox;
  int n, req_stride;
};

__device__ __forceinline__ void st_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<u64 *>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1_u64(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_sc1_u64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// copy `words` 8-byte words LDS/global <-> global with L1-bypassing accesses, all threads of the workgroup
__device__ __forceinline__ void publish_words(u64 *dst_global, const u64 *src_lds, int words) {
  for (int w = threadIdx.x; w < words; w += kPThreads) st_sc1_u64(dst_global + w, src_lds[w]);
}
__device__ __forceinline__ void fetch_words(u64 *dst_lds, const u64 *src_global, int words) {
  for (int w = threadIdx.x; w < words; w += kPThreads) dst_lds[w] = ld_sc1_u64(src_global + w);
}

template <int METHOD>
using PMachine = typename std::conditional<METHOD == 0, DifMachine<kM>, BcMachine<kM>>::type;

template <int MODEL, int METHOD, bool FAST>
__global__ __launch_bounds__(kPThreads) void persist_fit_kernel(PersistCtx ctx) {
  using Machine = PMachine<METHOD>;
  using Mdl = BrdfModel<MODEL>;
  using Req = PersistReq<MODEL>;
  static_assert(sizeof(Machine) % 8 == 0 && sizeof(Req) % 8 == 0, "published as 8-byte words");
  __shared__ Machine sm;
  __shared__ Req rq;
  __shared__ double red[reduce_buf_doubles<kPThreads>()];
  __shared__ double sums[kSlots];
  __shared__ int s_role;  // 1: this workgroup arrived last, 0: wait for the flag, -1: abort

  const int tid = threadIdx.x;
  const bool first_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < kWave;  // see stream_fit.hip
  const int G = gridDim.x;
  const int blk = blockIdx.x;
  const int n = ctx.n;
  PersistCtl *ctl = ctx.ctl;

  // ---- this workgroup's tile: one HBM read, then registers --------------------------------------------------
  const int tile = (n + G - 1) / G;  // <= kPTile, checked on the host
  const int begin = blk * tile;
  const int end = min(n, begin + tile);
  double s0[kPSpt], sx[kPSpt];
  Prep pq[kPSpt];
  bool ok[kPSpt];
  {
    double r1[kPSpt], r2[kPSpt];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kPSpt; ++k) {
      const int i = begin + tid + k * kPThreads;
      ok[k] = i < end;
      const int ii = ok[k] ? i : 0;
      s0[k] = ctx.c0[ii];
      r1[k] = Mdl::uses_c1 ? ctx.c1[ii] : 0.0;
      r2[k] = Mdl::uses_c2 ? ctx.c2[ii] : 0.0;
      sx[k] = ctx.x[ii];
    }
#pragma unroll
    for (int k = 0; k < kPSpt; ++k) {
      pq[k] = Mdl::template prepare<FAST>(s0[k], r1[k], r2[k]);
      if (FAST && ok[k] && !Mdl::domain_ok(s0[k], r1[k], r2[k])) bad = true;
    }
    if (FAST && bad) __hip_atomic_store(&ctl->domain_bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // dif state of this lane's samples: f(p), f(q), the Broyden coefficient of the last trial and the secant rows
  double hx[kPSpt], wrk[kPSpt], bro[kPSpt], jac[kPSpt][kM];
#pragma unroll
  for (int k = 0; k < kPSpt; ++k) {
    hx[k] = wrk[k] = bro[k] = 0.0;
    jac[k][0] = jac[k][1] = jac[k][2] = 0.0;
  }
  int cur_sel_hx = 0, cur_sel_j = 0;
  double dp_prev[kM] = {0.0, 0.0, 0.0};

  // the first request was written by the host before the launch
  fetch_words(reinterpret_cast<u64 *>(&rq), reinterpret_cast<const u64 *>(ctx.req), (int)(sizeof(Req) / 8));
  __syncthreads();

  for (unsigned epoch = 0;; ++epoch) {
    const int kind = rq.kind;
    if (kind == RQ_DONE) break;
    const PassUniforms<MODEL> &u = rq.u;
    if (METHOD == 0) {  // commit what the machine decided about the previous trial (speculative protocol)
      if (rq.sel_j != cur_sel_j) {  // adopt the Broyden-updated rows: J += coeff * Dp^T  (lm_core.c:764-765)
#pragma unroll
        for (int k = 0; k < kPSpt; ++k)
#pragma unroll
          for (int j = 0; j < kM; ++j) jac[k][j] = jac[k][j] + bro[k] * dp_prev[j];
        cur_sel_j = rq.sel_j;
      }
      if (rq.sel_hx != cur_sel_hx) {  // step accepted: hx <- f(p + Dp)
#pragma unroll
        for (int k = 0; k < kPSpt; ++k) hx[k] = wrk[k];
        cur_sel_hx = rq.sel_hx;
      }
    }

    double acc[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
    double mx = 0.0;
    switch (kind) {
    case RQ_EVAL:
#pragma unroll
      for (int k = 0; k < kPSpt; ++k) {
        const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double e = ok[k] ? sx[k] - f : 0.0;
        acc[0] += e * e;
        mx = fmax(mx, fabs(e));
      }
      block_reduce<1, kPThreads>(acc, mx, red, sums);
      break;
    case RQ_SCALED:
#pragma unroll
      for (int k = 0; k < kPSpt; ++k) {
        const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double t = ok[k] ? (sx[k] - f) / u.scal : 0.0;
        acc[0] += t * t;
      }
      block_reduce<1, kPThreads>(acc, mx, red, sums);
      break;
    case RQ_JAC:
#pragma unroll
      for (int k = 0; k < kPSpt; ++k) {
        double f0 = 0.0, j[kM];
        model_fd_row<MODEL, FAST>(u, s0[k], pq[k], true, f0, 0.0, false, j);
        double e = sx[k] - f0;
        if (!ok[k]) e = j[0] = j[1] = j[2] = 0.0;
        acc_normal_eq(j, e, acc, acc + kNL);
        acc[kNL + kM] += e * e;
      }
      block_reduce<SumLayout<kM>::JAC, kPThreads>(acc, mx, red, sums);
      break;
    case RQ_DIF_INIT:
#pragma unroll
      for (int k = 0; k < kPSpt; ++k) {
        hx[k] = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double e = ok[k] ? sx[k] - hx[k] : 0.0;
        acc[0] += e * e;
      }
      block_reduce<1, kPThreads>(acc, mx, red, sums);
      break;
    case RQ_DIF_JAC:
#pragma unroll
      for (int k = 0; k < kPSpt; ++k) {
        double f0 = 0.0;
        model_fd_row<MODEL, FAST>(u, s0[k], pq[k], false, f0, hx[k], true, jac[k]);
        double e = sx[k] - hx[k];
        if (!ok[k]) e = jac[k][0] = jac[k][1] = jac[k][2] = 0.0;
        acc_normal_eq(jac[k], e, acc, acc + kNL);
      }
      block_reduce<SumLayout<kM>::DIF_JAC, kPThreads>(acc, mx, red, sums);
      break;
    case RQ_DIF_TRIAL: {  // speculative protocol: the Broyden update is formed but only its coefficient is kept
#pragma unroll
      for (int j = 0; j < kM; ++j) dp_prev[j] = u.dp[j];
#pragma unroll
      for (int k = 0; k < kPSpt; ++k) {
        const double w = model_value_q<MODEL, FAST>(u, s0[k], pq[k]);
        double t = 0.0;  // (f(p+Dp) - f(p) - J Dp) / ||Dp||^2, lm_core.c:761-763
#pragma unroll
        for (int l = 0; l < kM; ++l) t += jac[k][l] * u.dp[l];
        t = (w - hx[k] - t) / u.dp_l2;
        double jn[kM];
#pragma unroll
        for (int j = 0; j < kM; ++j) jn[j] = jac[k][j] + t * u.dp[j];
        double en = sx[k] - w, eo = sx[k] - hx[k];
        if (!ok[k]) en = eo = t = jn[0] = jn[1] = jn[2] = 0.0;
        wrk[k] = w;
        bro[k] = t;
        acc[0] += en * en;
        acc_normal_eq(jn, en, acc + 1, acc + 1 + kNL);
        acc[1 + kNL + kM + 0] += jn[0] * eo;
        acc[1 + kNL + kM + 1] += jn[1] * eo;
        acc[1 + kNL + kM + 2] += jn[2] * eo;
      }
      block_reduce<SumLayout<kM>::DIF_TRIAL, kPThreads>(acc, mx, red, sums);
      break;
    }
    default: break;
    }

    // ---- publish this workgroup's partial line, then take a ticket ------------------------------------------
    double *line = ctx.partials + ((size_t)(epoch & 1u) * G + blk) * kLine;
    if (tid < kLine) {
      st_sc1(line + tid, tid < kSlots ? sums[tid] : 0.0);  // lanes 0..15 of wave 0: one 128-B line, one instruction
      drain_stores();
    }
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_role = (old == (epoch + 1u) * (unsigned)G - 1u) ? 1 : 0;
    }
    __syncthreads();

    Req *req_next = reinterpret_cast<Req *>(reinterpret_cast<char *>(ctx.req) + (size_t)((epoch + 1u) & 1u) * ctx.req_stride);
    if (s_role == 1) {
      // ---- last arriver: fold, step, publish ------------------------------------------------------------------
      fetch_words(reinterpret_cast<u64 *>(&sm), reinterpret_cast<const u64 *>(ctx.machine), (int)(sizeof(Machine) / 8));
      double pv[kSlots];
      {
        const double *lines = ctx.partials + (size_t)(epoch & 1u) * G * kLine;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) pv[k] = (tid < G) ? ld_sc1(lines + (size_t)tid * kLine + k) : 0.0;
      }
      __syncthreads();
      switch (kind) {
      case RQ_JAC: block_reduce<SumLayout<kM>::JAC, kPThreads>(pv, pv[kSums], red, sums); break;
      case RQ_DIF_JAC: block_reduce<SumLayout<kM>::DIF_JAC, kPThreads>(pv, pv[kSums], red, sums); break;
      case RQ_DIF_TRIAL: block_reduce<SumLayout<kM>::DIF_TRIAL, kPThreads>(pv, pv[kSums], red, sums); break;
      default: block_reduce<1, kPThreads>(pv, pv[kSums], red, sums); break;
      }
      if (first_wave) {
        sm.template step<true>(sums, sums[kSums]);
        rq.kind = sm.h.req.kind;
        rq.aux = sm.h.req.aux;
        rq.sel_hx = sm.h.req.sel_hx;
        rq.sel_j = sm.h.req.sel_j;
        if (rq.kind != RQ_DONE) rq.u.build(sm.h.req);
      }
      __syncthreads();
      publish_words(reinterpret_cast<u64 *>(ctx.machine), reinterpret_cast<const u64 *>(&sm), (int)(sizeof(Machine) / 8));
      publish_words(reinterpret_cast<u64 *>(req_next), reinterpret_cast<const u64 *>(&rq), (int)(sizeof(Req) / 8));
      drain_stores();
      __syncthreads();
      if (tid == 0) {
        if (rq.kind == RQ_DONE) {  // the fit is over: results to the pinned mailbox before anybody leaves
          Mailbox *mb = ctx.mbox;
          mb->ret = sm.c.ret;
          mb->passes = (int)epoch + 1;
          if constexpr (METHOD == 1)
            mb->infeasible_mask = sm.c.infeasible_mask;
          else
            mb->infeasible_mask = 0;
          mb->domain_bad = (int)__hip_atomic_load(&ctl->domain_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int i = 0; i < kM; ++i) mb->p[i] = sm.h.p[i];
          for (int i = 0; i < kInfoSz; ++i) mb->info[i] = sm.c.info[i];
          for (int i = 0; i < kM * kM; ++i) mb->covar[i] = sm.c.covar[i];
          mb->t_last = (long long)wall_clock64();
          __threadfence_system();
          __hip_atomic_store(&mb->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&ctl->gen, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // ---- everybody else: one lane polls the generation flag, bounded -----------------------------------------
      if (tid == 0) {
        const long long t0 = (long long)wall_clock64();
        int role = 0;
        for (unsigned spins = 0;; ++spins) {
          if (__hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch + 1u) break;
          if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                (long long)wall_clock64() - t0 > kSpinBudgetTicks) {
              __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              role = -1;
              break;
            }
          }
          __builtin_amdgcn_s_sleep(2);
        }
        s_role = role;
      }
      __syncthreads();
      if (s_role < 0) return;  // give up: the host sees no `done`, reads ctl->abort and falls back
      fetch_words(reinterpret_cast<u64 *>(&rq), reinterpret_cast<const u64 *>(req_next), (int)(sizeof(Req) / 8));
    }
    __syncthreads();
  }
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

namespace {

struct PWorkspace {
  int device = -1, cus = 0;
  char *d_block = nullptr;  // ctl | req[2] | machine | partials
  char *h_block = nullptr;  // pinned staging of the same prefix (ctl, req[0], machine)
  Mailbox *h_mbox = nullptr, *d_mbox = nullptr;
  size_t off_req = 0, off_machine = 0, off_partials = 0, total = 0, req_stride = 0;
  FitStats stats{};

  int ensure(int dev) {
    if (device == dev && d_block) return 0;
    device = dev;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
    req_stride = 1024;  // >= sizeof(PersistReq<MODEL>) for every model, multiple of 128
    off_req = sizeof(PersistCtl);
    off_machine = off_req + 2 * req_stride;
    off_partials = off_machine + 2048;
    total = off_partials + sizeof(double) * 2 * (size_t)cus * kLine;
    HIP_OK(hipMalloc(&d_block, total));
    HIP_OK(hipHostMalloc(&h_block, off_partials, hipHostMallocDefault));
    HIP_OK(hipHostMalloc(&h_mbox, sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void **)&d_mbox, h_mbox, 0));
    return 0;
  }
};
thread_local PWorkspace g_pws;

template <int MODEL, int METHOD, bool FAST>
int persist_attempt(const StreamFitArgs &a, PWorkspace &ws, bool *retry_exact, bool *unavailable) {
  using Machine = PMachine<METHOD>;
  using Req = PersistReq<MODEL>;
  static_assert(sizeof(Req) <= 1024 && sizeof(Machine) <= 2048, "persistent workspace layout");
  *retry_exact = *unavailable = false;
  // workgroups: enough that a tile fits the registers (<= 4096 samples), at most one per CU (co-residency),
  // and not more than the work justifies (the hand-off cost grows with the number of participants)
  long long want = ((long long)a.n + 1023) / 1024;
  if (want < 1) want = 1;
  const int G = (int)std::min<long long>(ws.cus, want);
  HIP_OK(hipStreamSynchronize(a.stream));
  memset(ws.h_block, 0, ws.off_partials);
  Machine &m = *reinterpret_cast<Machine *>(ws.h_block + ws.off_machine);
  if constexpr (METHOD == 0) {
    m.start(a.p, a.n, a.itmax, a.opts, a.covar != nullptr, /*speculative=*/1);
    if (m.h.req.kind == RQ_DONE) {
      set_error("dlevmar_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM);
      return kLmError;
    }
  } else {
    m.start(a.p, a.n, a.lb, a.ub, a.dscl, a.itmax, a.opts, a.covar != nullptr);
    if (m.h.req.kind == RQ_DONE) {
      switch (m.c.bad_input) {
      case 1: set_error("dlevmar_bc_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM); break;
      case 2: set_error("dlevmar_bc_dif(): at least one lower bound exceeds the upper one"); break;
      default: set_error("dlevmar_bc_dif(): scaling constants should be positive"); break;
      }
      return kLmError;
    }
    if (FAST || !brdf_fast_path_enabled())
      for (int i = 0; i < kM; ++i)
        if (m.c.infeasible_mask & (1 << i))
          fprintf(stderr, "Warning: component %d of starting point not feasible in dlevmar_bc_dif()! [%g projected to %g]\n",
                  i, m.c.p_start[i], m.h.p[i]);
  }
  Req &r0 = *reinterpret_cast<Req *>(ws.h_block + ws.off_req);
  r0.kind = m.h.req.kind;
  r0.aux = m.h.req.aux;
  r0.sel_hx = m.h.req.sel_hx;
  r0.sel_j = m.h.req.sel_j;
  r0.u.build(m.h.req);

  Mailbox &mb = *ws.h_mbox;
  memset(&mb, 0, sizeof mb);
  HIP_OK(hipMemcpyAsync(ws.d_block, ws.h_block, ws.off_partials, hipMemcpyHostToDevice, a.stream));

  PersistCtx c;
  c.c0 = a.d_angles;
  c.c1 = a.d_angles + a.n;
  c.c2 = a.d_angles + 2 * (size_t)a.n;
  c.x = a.d_x;
  c.partials = reinterpret_cast<double *>(ws.d_block + ws.off_partials);
  c.ctl = reinterpret_cast<PersistCtl *>(ws.d_block);
  c.req = ws.d_block + ws.off_req;
  c.machine = ws.d_block + ws.off_machine;
  c.mbox = ws.d_mbox;
  c.n = a.n;
  c.req_stride = (int)ws.req_stride;

  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, a.stream));
  hipLaunchKernelGGL((persist_fit_kernel<MODEL, METHOD, FAST>), dim3(G), dim3(kPThreads), 0, a.stream, c);
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(e1, a.stream));
  HIP_OK(hipStreamSynchronize(a.stream));  // one launch: the kernel always terminates (bounded spins)
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (!mb.done) {  // aborted: not co-resident / spin budget exhausted
    *unavailable = true;
    return 0;
  }
  if (FAST && mb.domain_bad) {
    *retry_exact = true;
    return 0;
  }
  for (int i = 0; i < kM; ++i) a.p[i] = mb.p[i];
  if (a.info)
    for (int i = 0; i < kInfoSz; ++i) a.info[i] = mb.info[i];
  if (a.covar)
    for (int i = 0; i < kM * kM; ++i) a.covar[i] = mb.covar[i];
  ws.stats.passes = mb.passes;
  ws.stats.launches = 1;
  ws.stats.jac_passes = (long long)mb.info[8];
  ws.stats.eval_passes = mb.passes - ws.stats.jac_passes;
  ws.stats.device_us = 1e3 * ms;
  return mb.ret;
}

template <int MODEL, int METHOD>
int persist_run_mm(const StreamFitArgs &a, PWorkspace &ws, bool *unavailable) {
  bool retry = false;
  double keep[kM];
  for (int i = 0; i < kM; ++i) keep[i] = a.p[i];
  int ret;
  if (brdf_fast_path_enabled() || MODEL == MODEL_WARD) {
    ret = persist_attempt<MODEL, METHOD, true>(a, ws, &retry, unavailable);
    if (!retry || *unavailable) return ret;
    for (int i = 0; i < kM; ++i) a.p[i] = keep[i];
  }
  if constexpr (MODEL != MODEL_WARD)
    return persist_attempt<MODEL, METHOD, false>(a, ws, &retry, unavailable);
  else
    return kLmError;
}

}  // namespace

FitStats persist_fit_last_stats() { return g_pws.stats; }

// returns true if the persistent path handled the fit (*ret is then the solver's return value)
bool persist_fit_try(const StreamFitArgs &a, int *ret) {
  // Opt-in (BRDF_HIP_PERSISTENT=1).  Measured on MI355X (1M-sample Ward fit): 34 us per dlevmar_dif pass and 17 us
  // per dlevmar_bc_dif pass against 22 / 11 us for the launch chain -- each pass needs four cross-XCD
  // visibility hops (partials -> ticket, ticket -> last arriver, publish -> flag, flag -> request fetch) at
  // ~2 us each, which costs more than the ~1.5 us kernel boundary + one round trip they replace, even though
  // the sweep itself no longer touches HBM.  Kept as a correct, tested alternative; the chain is the default.
  const char *e = getenv("BRDF_HIP_PERSISTENT");
  if (!(e && e[0] == '1')) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  PWorkspace &ws = g_pws;
  if (ws.ensure(dev) != 0) return false;
  if ((long long)a.n > (long long)ws.cus * kPTile) return false;  // does not fit the register file: launch chain
  bool unavailable = false;
  int r;
  switch (a.model * 2 + a.method) {
  case 0: r = persist_run_mm<0, 0>(a, ws, &unavailable); break;
  case 1: r = persist_run_mm<0, 1>(a, ws, &unavailable); break;
  case 2: r = persist_run_mm<1, 0>(a, ws, &unavailable); break;
  case 3: r = persist_run_mm<1, 1>(a, ws, &unavailable); break;
  case 4: r = persist_run_mm<2, 0>(a, ws, &unavailable); break;
  default: r = persist_run_mm<2, 1>(a, ws, &unavailable); break;
  }
  if (unavailable) {
    static bool warned = false;
    if (!warned) fprintf(stderr, "libbrdf_hip: persistent single-launch path unavailable (grid not co-resident?); using the launch chain\n");
    warned = true;
    return false;
  }
  *ret = r;
  return true;
}

}  // namespace brdf
