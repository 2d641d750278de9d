// channels_fit_impl.h -- K <= 3 box-constrained fits over ONE set of cosine planes in ONE resident launch.
//
// The reference fits the three colour channels of a capture over the same planes, one dlevmar_bc_dif call after the other
// (brdfdata.cpp:1159-1181 for the single-BRDF mode, :1202-1219 per pixel).  A resident single fit (resident_fit_impl.h)
// is a chain of latencies -- sweep, reduction, two visibility hops, the serial LM step -- during most of which seven of a
// workgroup's eight waves have nothing to do.  Here the chains of K fits are interleaved:
//
//   waves 0..K-1   control wave of channel c: its own BcMachine, its own request / uniforms, its own exchange tables.  It
//                  never sweeps: it finishes the reduction of its channel's sums, publishes, gathers (two levels, as in the
//                  single-fit kernel), steps its machine, posts the next request.
//   waves 4..7     the sweeping waves.  Each lane stands for TWO lanes of the single-fit geometry (virtual threads t and
//                  t + 256 of that kernel's 512): 2 x 8 samples per lane, planes and the prepared invariants in registers
//                  (shared by all channels: loaded and prepared once), the K measurement vectors in LDS (private words of
//                  the lane that reads them).  They serve whichever channel has a request out: while channel A's sums are in
//                  flight and its machine is stepped, they sweep channels B and C.
//
// No workgroup barrier after start-up (a control wave that polls the exchange tables could not take part in one): requests
// and arrivals are LDS words -- req_epoch[c] posted by control wave c behind its uniforms, arrive[c] counted up by every
// sweeping wave behind its partial sums; a wave's LDS operations execute in order, so a word is behind the data it covers.
//
// BIT-IDENTICAL to the single-fit resident path, channel by channel (tests/test_gpu_parity.py): the same sample -> virtual
// thread -> slot mapping, the same per-(virtual-)lane accumulation order, the same two-stage reduction (two DPP steps, the
// 128 columns, column c + column c + 64, one tree per slot), the same exchange and fold order, the same machines.
// Only who executes an operation differs.  dlevmar_bc_dif and dlevmar_bc_der (METHOD 1 of the resident kernels); the other
// entry points keep per-sample state per channel (dlevmar_dif: f(p), f(p + Dp), the secant Jacobian -- 96 KB of LDS per
// channel at 4,096 samples per workgroup) and run their channels one after the other.
#pragma once

#include "resident_fit_impl.h"

namespace brdf {

constexpr int kMaxChannels = 3;
constexpr int kCFirstWorker = 4;                      // waves 4..7 sweep: one per SIMD, next to (at most) one control wave
constexpr int kCWorkers = kRThreads / kWave - kCFirstWorker;
constexpr int kCVirt = 2;                             // virtual threads (of the single-fit kernel's 512) per sweeping lane
constexpr int kCRedSlots = SumLayout<kM>::JAC;        // most values a pass of these entry points reduces: 10 (EVAL: 1 + max; MULTI: 8)
constexpr unsigned kCDone = 0xFFFFFFFFu;              // req_epoch: the channel has finished
static_assert(kCWorkers * kCVirt * kWave == kRThreads, "the sweeping lanes stand for all 512 threads of the single-fit geometry");

struct ChannelsCtx {
  const double *c0, *c1, *c2;
  const double *x[kMaxChannels];
  u64 *rows, *groups;  // channel c uses rows + c * kRowsGranules, groups + c * kGroupsGranules (same layouts as ResidentCtx)
  ResidentCtl *ctl;
  unsigned launch_id;
  double p0[kMaxChannels][kM], opts[5], lb[kM], ub[kM], dscl[kM];
  int itmax, has_opts, has_lb, has_ub, has_dscl, want_covar, multi, analytic, spec_jac;
  Mailbox *mbox;  // [K]
  int n, K;
  unsigned tag_base[kMaxChannels];
  long long spin_ticks;
  int replicas;
  int sabotage_epoch;  // test hook (BRDF_HIP_RESIDENT_SABOTAGE): the last workgroup withholds its rows at this epoch; -1 = never
};

// what control_exchange() / gather_block() read of a context, for one channel
struct ChannelView {
  u64 *rows, *groups;
  ResidentCtl *ctl;
  unsigned launch_id, tag_base;
  long long spin_ticks;
  int sabotage_epoch, replicas;
  long long *trace;
  int trace_epoch;
};

__device__ __forceinline__ unsigned lds_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }


// stage 1 of a channel's reduction by a sweeping wave, both virtual threads of its lanes: worker_reduce()'s first stage (two DPP
// steps inside each row of 16 lanes; lanes 3, 7, 11, .. park the sum of four as buf[slot][virtual thread / 4]; with WITH_MAX slot NS is
// the max), all values' steps side by side
template <int NS, bool WITH_MAX>
__device__ __forceinline__ void channel_stage1(const double (&acc)[kCVirt][kCRedSlots], const double (&mx)[kCVirt], double *buf, const int (&vt)[kCVirt],
                                               int lane) {
  double t[kCVirt][NS + 1];
#pragma unroll
  for (int v = 0; v < kCVirt; ++v) {
#pragma unroll
    for (int k = 0; k < NS; ++k) t[v][k] = acc[v][k];
    t[v][NS] = mx[v];
  }
#pragma unroll
  for (int v = 0; v < kCVirt; ++v) {
#pragma unroll
    for (int k = 0; k < NS; ++k) t[v][k] = t[v][k] + dpp_move<0x111, 0xf, 0xf>(t[v][k], 0.0);  // row_shr:1
    if (WITH_MAX) t[v][NS] = fmax(t[v][NS], dpp_move<0x111, 0xf, 0xf>(t[v][NS], 0.0));
  }
#pragma unroll
  for (int v = 0; v < kCVirt; ++v) {
#pragma unroll
    for (int k = 0; k < NS; ++k) t[v][k] = t[v][k] + dpp_move<0x112, 0xf, 0xf>(t[v][k], 0.0);  // row_shr:2
    if (WITH_MAX) t[v][NS] = fmax(t[v][NS], dpp_move<0x112, 0xf, 0xf>(t[v][NS], 0.0));
  }
  if ((lane & 3) == 3) {
#pragma unroll
    for (int v = 0; v < kCVirt; ++v)
#pragma unroll
      for (int k = 0; k < NS + (WITH_MAX ? 1 : 0); ++k) buf[k * kRedCols + (vt[v] >> 2)] = t[v][k];
  }
}

// stage 2, by the channel's control wave: worker_reduce()'s second stage for every slot (column l + column l + 64, one DPP tree; with
// WITH_MAX the last of the NV slots is the max and goes to out[kSums]), the NV trees side by side
template <int NV, bool WITH_MAX>
__device__ __forceinline__ void channel_stage2(const double *buf, int lane, double *out) {
  double s[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double a = buf[k * kRedCols + lane], b = buf[k * kRedCols + lane + kWave];
    s[k] = (WITH_MAX && k == NV - 1) ? fmax(a, b) : a + b;
  }
#define CH_STEP(CTRL, ROWS)                                                                                     \
  _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                                              \
    if (WITH_MAX && k == NV - 1)                                                                                \
      s[k] = fmax(s[k], dpp_move<CTRL, ROWS, 0xf>(s[k], 0.0));                                                  \
    else                                                                                                        \
      s[k] = s[k] + dpp_move<CTRL, ROWS, 0xf>(s[k], 0.0);                                                       \
  }
  CH_STEP(0x111, 0xf)  // row_shr:1   (wave_reduce_to_last's six steps, device_common.h)
  CH_STEP(0x112, 0xf)  // row_shr:2
  CH_STEP(0x114, 0xf)  // row_shr:4
  CH_STEP(0x118, 0xf)  // row_shr:8
  CH_STEP(0x142, 0xa)  // row_bcast:15
  CH_STEP(0x143, 0xc)  // row_bcast:31
#undef CH_STEP
  if (lane == kWave - 1) {
#pragma unroll
    for (int k = 0; k < NV; ++k) out[(WITH_MAX && k == NV - 1) ? kSums : k] = s[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int MODEL, bool FAST>
__global__ __launch_bounds__(kRThreads) void channels_fit_kernel(ChannelsCtx ctx) {
  using Machine = BcMachine<kM>;
  using Mdl = BrdfModel<MODEL>;
  static_assert(sizeof(Machine) % 4 == 0, "machine copied as dwords");
  __shared__ Machine sm[kMaxChannels];
  __shared__ PassUniforms<MODEL> su[kMaxChannels];
  __shared__ double red[kMaxChannels][kCRedSlots * kRedCols];  // stage 1 of a channel's reduction: [slot][virtual thread / 4]
  __shared__ double sums[kMaxChannels][kSlots];
  __shared__ double xl[kMaxChannels * kRCap];                  // the channels' measurements: [c][k * 512 + virtual thread]
  __shared__ unsigned req_epoch[kMaxChannels], arrive[kMaxChannels];
  __shared__ int s_abort;

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = ctx.K;
  const int G = (int)gridDim.x;
  const int n = ctx.n;
  if (tid < kMaxChannels) {
    req_epoch[tid] = 0u;
    arrive[tid] = 0u;
  }
  if (tid == 0) s_abort = 0;
  __syncthreads();  // the only workgroup barrier of the launch

  // the resident tile: the single-fit kernel's dealing of tiles (resident_fit_kernel), so that every sum is formed over the
  // same samples in the same order
  int vb = (int)blockIdx.x;
  if ((G & 7) == 0) vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int tile = (n + G - 1) / G;  // <= kRTile, checked on the host
  const int begin = vb * tile;
  const int end = min(n, begin + tile);
  const int nk = (tile + kRThreads - 1) / kRThreads;

  if (wave < K) {
    // ======================================= control wave of channel `wave` =======================================
    const int c = wave;
    Machine &m = sm[c];
    PassUniforms<MODEL> &u = su[c];
    double *mysums = sums[c];
    {
      double p0[kM], opts[5], lb[kM], ub[kM], dscl[kM];  // (locals first: see resident_fit_kernel)
#pragma unroll
      for (int i = 0; i < kM; ++i) {
        p0[i] = c == 0 ? ctx.p0[0][i] : (c == 1 ? ctx.p0[1][i] : ctx.p0[2][i]);  // (constant indices into the argument struct)
        lb[i] = ctx.lb[i];
        ub[i] = ctx.ub[i];
        dscl[i] = ctx.dscl[i];
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) opts[i] = ctx.opts[i];
      m.start(p0, n, ctx.has_lb ? lb : nullptr, ctx.has_ub ? ub : nullptr, ctx.has_dscl ? dscl : nullptr, ctx.itmax,
              ctx.has_opts ? opts : nullptr, ctx.want_covar, ctx.multi, ctx.spec_jac);
      m.c.analytic_jac = ctx.analytic;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const ChannelView view{ctx.rows + (size_t)c * kRowsGranules, ctx.groups + (size_t)c * kGroupsGranules, ctx.ctl, ctx.launch_id,
                           c == 0 ? ctx.tag_base[0] : (c == 1 ? ctx.tag_base[1] : ctx.tag_base[2]), ctx.spin_ticks, ctx.sabotage_epoch,
                           ctx.replicas, nullptr, -1};
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long last_ = 0;
    long long n_jac = 0;
    const long long t_first = (long long)wall_clock64();
#ifdef BRDF_STAMPS  // cycles per section of this control wave's loop, summed over the passes (mailbox stamps[1..5])
    long long cst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clast_ = clock64();
#define CSTAMP(i) do { const long long now_ = clock64(); cst_[i] += now_ - clast_; clast_ = now_; } while (0)
#else
#define CSTAMP(i) do {} while (0)
#endif
    unsigned epoch = 0;  // passes of this channel so far = the exchange epoch of the pass in flight
    for (;; ++epoch) {
      const int kind = m.h.req.kind;
      if (kind == RQ_DONE) break;
      if (kind == RQ_JAC) ++n_jac;
      CSTAMP(4);  // (the step)
      u.build(m.h.req, /*need_base=*/epoch == 0, m.c.analytic_jac != 0);
      CSTAMP(5);  // (the uniforms)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the request and its uniforms are in LDS ...
      __builtin_amdgcn_wave_barrier();
      lds_store(&req_epoch[c], epoch + 1u);                    // ... behind this word
      {  // the sweeping waves' partial sums
        const long long t0 = (long long)wall_clock64();
        for (unsigned spins = 0; lds_load(&arrive[c]) != (unsigned)kCWorkers * (epoch + 1u); ++spins) {
          if ((spins & 255u) == 255u) {
            if (lds_load(reinterpret_cast<unsigned *>(&s_abort)) || (long long)wall_clock64() - t0 > ctx.spin_ticks) {
              s_abort = 1;
              __hip_atomic_store(&ctx.ctl->abort, ctx.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              return;
            }
          }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      CSTAMP(1);  // (waiting for the sweeping waves: their sweep + stage 1)
      // stage 2 of the reduction, every slot exactly as worker_reduce() finishes it: column l + column l + 64, one tree -- all
      // the slots' trees side by side (each is a chain of six dependent DPP steps)
      switch (kind) {
      case RQ_JAC: channel_stage2<SumLayout<kM>::JAC, false>(red[c], lane, mysums); break;
      case RQ_EVAL_MULTI: channel_stage2<kMaxCand, false>(red[c], lane, mysums); break;
      default: channel_stage2<2, true>(red[c], lane, mysums); break;  // (evaluation passes: the sum and the max)
      }
      CSTAMP(2);  // (stage 2)
      bool alive = true;
      if (G > 1) {
        switch (kind) {
        case RQ_JAC: alive = control_exchange<SumLayout<kM>::JAC, false>(view, epoch, mysums, &s_abort, st_, last_); break;
        case RQ_EVAL_MULTI: alive = control_exchange<kMaxCand, false>(view, epoch, mysums, &s_abort, st_, last_); break;
        default: alive = control_exchange<1, true>(view, epoch, mysums, &s_abort, st_, last_); break;
        }
      }
      if (!alive) return;  // (s_abort is set: the other waves of the workgroup leave too)
      CSTAMP(3);  // (the exchange)
      m.template step<true, true, false, true>(mysums, mysums[kSums]);
    }
    lds_store(&req_epoch[c], kCDone);
    if (blockIdx.x == 0 && lane == 0) {  // every workgroup holds the same finished machine; workgroup 0 reports
      Mailbox *mb = ctx.mbox + c;
      mb->ret = m.c.ret;
      mb->passes = (int)epoch;
      mb->infeasible_mask = m.c.infeasible_mask;
      mb->domain_bad = __hip_atomic_load(&ctx.ctl->domain_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ctx.launch_id ? 1 : 0;
      mb->n_jac = n_jac;
      mb->n_eval = (long long)epoch - n_jac;
      mb->t_first = t_first;
      mb->t_last = (long long)wall_clock64();
#ifdef BRDF_STAMPS
      for (int i = 0; i < 6; ++i) mb->stamps[i] = cst_[i];  // ([6], [7]: the sweeping waves' view, below)
#endif
      for (int i = 0; i < kM; ++i) mb->p[i] = m.h.p[i];
      for (int i = 0; i < kInfoSz; ++i) mb->info[i] = m.c.info[i];
      for (int i = 0; i < kM * kM; ++i) mb->covar[i] = m.c.covar[i];
      __threadfence_system();
      __hip_atomic_store(&mb->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (wave < kCFirstWorker) return;  // (no role: fewer than four channels)

  // =============================================== sweeping waves ================================================
  const int ww = wave - kCFirstWorker;
  double sc0[kCVirt][kRSpt], sq1[kCVirt][kRSpt], sq2[kCVirt][kRSpt];
  unsigned okm = 0;  // bit v * 8 + k: virtual thread v of this lane has a sample in slot k
  int vt[kCVirt];
  {
    bool bad = false;
#pragma unroll
    for (int v = 0; v < kCVirt; ++v) {
      vt[v] = (v * kCWorkers + ww) * kWave + lane;  // virtual waves 0..3 are the lanes' first halves, 4..7 their second
#pragma unroll
      for (int k = 0; k < kRSpt; ++k) {
        const int i = begin + vt[v] + k * kRThreads;
        const bool ok = i < end;
        okm |= ok ? (1u << (v * kRSpt + k)) : 0u;
        const int ii = ok ? i : begin;
        const double r0 = ctx.c0[ii];
        const double r1 = Mdl::uses_c1 ? ctx.c1[ii] : 0.0;
        const double r2 = Mdl::uses_c2 ? ctx.c2[ii] : 0.0;
        const Prep q = Mdl::template prepare<FAST>(r0, r1, r2);
        sc0[v][k] = r0;
        sq1[v][k] = q.q1;
        sq2[v][k] = q.q2;
        xl[k * kRThreads + vt[v]] = ctx.x[0][ii];
        if (K > 1) xl[kRCap + k * kRThreads + vt[v]] = ctx.x[1][ii];
        if (K > 2) xl[2 * kRCap + k * kRThreads + vt[v]] = ctx.x[2][ii];
        if (FAST && ok && !Mdl::domain_ok(r0, r1, r2)) bad = true;
      }
    }
    if (FAST && bad) __hip_atomic_store(&ctx.ctl->domain_bad, ctx.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  constexpr bool W2 = Mdl::prep_planes == 2;
  unsigned served0 = 0u, served1 = 0u, served2 = 0u;  // the last request of each channel this wave has swept
#ifdef BRDF_STAMPS  // wave 4 of workgroup 0: cycles in sweeps / in stage 1 of the reduction / requests served / cycles in all
  long long w_sweep = 0, w_stage1 = 0, w_served = 0, w_t0 = clock64();
#endif
  for (unsigned idle = 0;;) {
    bool any = false, all_done = true;
    for (int c = 0; c < K; ++c) {
      const unsigned e = lds_load(&req_epoch[c]);
      if (e == kCDone) continue;
      all_done = false;
      if (e == (c == 0 ? served0 : (c == 1 ? served1 : served2))) continue;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (c == 0) served0 = e; else if (c == 1) served1 = e; else served2 = e;
      any = true;
      const PassUniforms<MODEL> &ul = su[c];  // (in LDS)
      const int kind = sm[c].h.req.kind;
      const double *xc = xl + c * kRCap;
      double acc[kCVirt][kCRedSlots], mx[kCVirt] = {0.0, 0.0};
#pragma unroll
      for (int v = 0; v < kCVirt; ++v)
#pragma unroll
        for (int k = 0; k < kCRedSlots; ++k) acc[v][k] = 0.0;
#ifdef BRDF_STAMPS
      const long long w_a = clock64();
#endif
      // The sample bodies of resident_fit_impl.h's sweep_pass(), the two virtual threads of a slot side by side in one basic block
      // (a lone sample's dependent fp64 chains issue one instruction in four; two samples' chains interleave).  The request's
      // uniforms live in LDS: every slot loads the ones its request kind reads afresh, all at once behind ONE wait (left to the
      // compiler, every sample re-read them one by one -- ten ds_reads and nine waits per Jacobian row; held in registers for the
      // whole sweep, they push the sixteen resident samples' registers into scratch: 229-425 spilled VGPRs in every variant tried).
      auto sweep = [&](auto &&load, auto &&body) {
#ifdef BRDF_EXP_CHANNEL_SCALAR
        const auto u = scalar_copy(load());
#pragma unroll
        for (int k = 0; k < kRSpt; ++k)
          if (k < nk) {
            body(u, 0, k);
            body(u, 1, k);
          }
#else
#pragma unroll
        for (int k = 0; k < kRSpt; ++k)
          if (k < nk) {
            asm volatile("" ::: "memory");  // (the loads of this slot are this slot's: not merged with the previous slot's)
            const auto u = load();
            body(u, 0, k);
            body(u, 1, k);
          }
#endif
      };
      switch (kind) {
      case RQ_EVAL:
        sweep([&] { return EvalUniforms{ul.l0, ul.n0, ul.scal}; },
              [&](const EvalUniforms &u, int v, int k) {
                const Prep q{sq1[v][k], W2 ? sq2[v][k] : 0.0};
                const double f = model_value<MODEL, FAST>(u, sc0[v][k], q);
                double e2 = xc[k * kRThreads + vt[v]] - f;
                if (!(okm >> (v * kRSpt + k) & 1u)) e2 = 0.0;
                acc[v][0] = fma(e2, e2, acc[v][0]);
                mx[v] = fmax(mx[v], fabs(e2));
              });
        break;
      case RQ_SCALED:
        sweep([&] { return EvalUniforms{ul.l0, ul.n0, ul.scal}; },
              [&](const EvalUniforms &u, int v, int k) {
                const Prep q{sq1[v][k], W2 ? sq2[v][k] : 0.0};
                const double f = model_value<MODEL, FAST>(u, sc0[v][k], q);
                double t = (xc[k * kRThreads + vt[v]] - f) / u.scal;
                if (!(okm >> (v * kRSpt + k) & 1u)) t = 0.0;
                acc[v][0] = fma(t, t, acc[v][0]);
              });
        break;
      case RQ_EVAL_MULTI: {  // candidate by candidate (four uniforms at a time; every sum still takes its samples in slot order)
        const int ncand = __builtin_amdgcn_readfirstlane(ul.ncand);
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
          if (j < ncand) {
            sweep([&] { return EvalUniforms{ul.lk[j], ul.nk[j], 1.0}; },
                  [&](const EvalUniforms &u, int v, int k) {
                    const Prep q{sq1[v][k], W2 ? sq2[v][k] : 0.0};
                    double e2 = xc[k * kRThreads + vt[v]] - model_value<MODEL, FAST>(u, sc0[v][k], q);  // = model_value_k(.., j, ..)
                    if (!(okm >> (v * kRSpt + k) & 1u)) e2 = 0.0;
                    acc[v][j] = fma(e2, e2, acc[v][j]);
                  });
          }
        break;
      }
      case RQ_JAC: {
        const int analytic = __builtin_amdgcn_readfirstlane(ul.analytic), central = __builtin_amdgcn_readfirstlane(ul.central);
        auto jac_sweep = [&](auto jk) {
          constexpr int JK = decltype(jk)::value;  // 0 forward differences, 1 central differences, 2 the model's analytic row
          sweep(
            [&] {
              JacUniforms u;
              u.l0 = ul.l0;
              u.n0 = ul.n0;
              u.analytic = analytic;
              u.central = central;
              if (analytic) {
                u.an[0] = ul.an[0];
                u.an[1] = ul.an[1];
              } else {
#pragma unroll
                for (int j = 0; j < kM; ++j) {
                  u.lp[j] = ul.lp[j];
                  u.dinv[j] = ul.dinv[j];
                }
                u.np2 = ul.np2;
                if (central) {
#pragma unroll
                  for (int j = 0; j < kM; ++j) u.lm[j] = ul.lm[j];
                  u.nm2 = ul.nm2;
                }
              }
              return u;
            },
            [&](const JacUniforms &u, int v, int k) {  // (the request-uniform choice of the row's kind is made per slot, outside the sample bodies)
              const Prep q{sq1[v][k], W2 ? sq2[v][k] : 0.0};
              double f0 = 0.0, j[kM];
              if (JK == 2)
                model_an_row<MODEL, FAST>(u, sc0[v][k], q, f0, j);
              else if (JK == 1)
                model_fd_row_t<MODEL, FAST, true>(u, sc0[v][k], q, true, f0, 0.0, false, j);
              else
                model_fd_row_t<MODEL, FAST, false>(u, sc0[v][k], q, true, f0, 0.0, false, j);
              double e2 = xc[k * kRThreads + vt[v]] - f0;
              if (!(okm >> (v * kRSpt + k) & 1u)) e2 = j[0] = j[1] = j[2] = 0.0;
              acc_normal_eq_fma(j, e2, acc[v], acc[v] + kNL);
              acc[v][kNL + kM] = fma(e2, e2, acc[v][kNL + kM]);
            });
        };
        if (analytic)
          jac_sweep(std::integral_constant<int, 2>{});
        else if (central)
          jac_sweep(std::integral_constant<int, 1>{});
        else
          jac_sweep(std::integral_constant<int, 0>{});
        break;
      }
      default: break;
      }
#ifdef BRDF_STAMPS
      const long long w_b = clock64();
      w_sweep += w_b - w_a;
#endif
      // stage 1 of worker_reduce(): two in-row DPP steps, lanes 3, 7, 11, .. park the sums of four as red[slot][virtual thread / 4]
      switch (kind) {
      case RQ_JAC: channel_stage1<SumLayout<kM>::JAC, false>(acc, mx, red[c], vt, lane); break;
      case RQ_EVAL_MULTI: channel_stage1<kMaxCand, false>(acc, mx, red[c], vt, lane); break;
      default: channel_stage1<1, true>(acc, mx, red[c], vt, lane); break;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the partial sums are in LDS ...
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_fetch_add(&arrive[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ... behind this count
#ifdef BRDF_STAMPS
      w_stage1 += clock64() - w_b;
      ++w_served;
#endif
    }
    if (all_done) break;
    if (any) {
      idle = 0;
      continue;
    }
    if ((++idle & 255u) == 0u && lds_load(reinterpret_cast<unsigned *>(&s_abort))) break;  // a control wave gave up
    __builtin_amdgcn_s_sleep(1);
  }
#ifdef BRDF_STAMPS
  if (blockIdx.x == 0 && wave == kCFirstWorker && lane == 0) {
    ctx.mbox[0].stamps[6] = w_sweep;
    ctx.mbox[0].stamps[7] = w_stage1;
    ctx.mbox[1].stamps[6] = w_served;
    ctx.mbox[1].stamps[7] = clock64() - w_t0;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct ChannelsArgs {
  int method, model;  // method: 1 dlevmar_bc_dif, 2 dlevmar_bc_der with the model's analytic Jacobian (BRDF_METHOD_*)
  const double *d_angles;
  const double *d_x[kMaxChannels];
  int n, K;
  double *p;  // [K][3] in/out
  const double *lb, *ub, *dscl;
  int itmax;
  const double *opts;
  double *info, *covar;  // [K][10], [K][9] or null
  hipStream_t stream;
};

struct CWorkspace {
  int device = -1, cus = 0;
  char *d_block = nullptr;  // ctl | per channel: rows | group rows
  Mailbox *h_mbox = nullptr, *d_mbox = nullptr;  // [kMaxChannels]
  static constexpr size_t off_rows = sizeof(ResidentCtl);
  static constexpr size_t chan_bytes = sizeof(u64) * (kRowsGranules + kGroupsGranules);
  static constexpr size_t block_bytes = off_rows + kMaxChannels * chan_bytes;
  unsigned tag_base[kMaxChannels] = {0, 0, 0};
  unsigned launches = 0;
  FitStats stats[kMaxChannels] = {};
  double launch_us = 0.0;
  LaunchTimer timer;
  int backoff = 0, skip = 0;
  void release() {
    if (!d_block && !h_mbox) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (device >= 0 && cur != device) (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    if (d_block) (void)hipFree(d_block);
    if (h_mbox) (void)hipHostFree(h_mbox);
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    d_block = nullptr;
    h_mbox = d_mbox = nullptr;
  }
  ~CWorkspace() { release(); }
  int ensure(int dev) {
    if (device == dev && d_block) return 0;
    release();
    device = dev;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
    HIP_OK(hipMalloc(&d_block, block_bytes));
    HIP_OK(hipMemset(d_block, 0, block_bytes));
    for (int c = 0; c < kMaxChannels; ++c) tag_base[c] = 0;
    launches = 0;
    HIP_OK(hipHostMalloc(&h_mbox, sizeof(Mailbox) * kMaxChannels, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void **)&d_mbox, h_mbox, 0));
    return 0;
  }
};

template <int MODEL, bool FAST>
int channels_attempt(const ChannelsArgs &a, CWorkspace &ws, bool *retry_exact, bool *unavailable) {
  using Machine = BcMachine<kM>;
  *retry_exact = *unavailable = false;
  const int G = (int)std::min<long long>(ws.cus, std::max<long long>(1, ((long long)a.n + 1023) / 1024));  // (the single fit's grid)
  for (int c = 0; c < a.K; ++c) {  // the entry point's argument checks and warnings, per channel (the kernel starts its own machines)
    Machine m;
    memset(&m, 0, sizeof m);
    m.start(a.p + c * kM, a.n, a.lb, a.ub, a.dscl, a.itmax, a.opts, a.covar != nullptr, pg_candidates());
    if (m.h.req.kind == RQ_DONE) {
      switch (m.c.bad_input) {
      case 1: set_error("dlevmar_bc_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM); break;
      case 2: set_error("dlevmar_bc_dif(): at least one lower bound exceeds the upper one"); break;
      default: set_error("dlevmar_bc_dif(): scaling constants should be positive"); break;
      }
      return kLmError;
    }
    if (FAST || !brdf_fast_path_enabled())
      for (int i = 0; i < kM; ++i)  // same warning as lmbc_core.c:516-520
        if (m.c.infeasible_mask & (1 << i))
          fprintf(stderr, "Warning: component %d of starting point not feasible in dlevmar_bc_dif()! [%g projected to %g]\n", i,
                  m.c.p_start[i], m.h.p[i]);
  }
  memset(ws.h_mbox, 0, sizeof(Mailbox) * kMaxChannels);
  unsigned top = 0;
  for (int c = 0; c < kMaxChannels; ++c) top = std::max(top, ws.tag_base[c]);
  if (top > 0xF0000000u || ws.launches > 0xF0000000u) {  // tag / launch-id space nearly used up: start over from zeroed tables
    HIP_OK(hipMemsetAsync(ws.d_block, 0, CWorkspace::block_bytes, a.stream));
    for (int c = 0; c < kMaxChannels; ++c) ws.tag_base[c] = 0;
    ws.launches = 0;
  }
  ChannelsCtx c;
  c.c0 = a.d_angles;
  c.c1 = a.d_angles + a.n;
  c.c2 = a.d_angles + 2 * (size_t)a.n;
  for (int k = 0; k < kMaxChannels; ++k) c.x[k] = a.d_x[k < a.K ? k : 0];
  c.ctl = reinterpret_cast<ResidentCtl *>(ws.d_block);
  c.rows = reinterpret_cast<u64 *>(ws.d_block + CWorkspace::off_rows);
  c.groups = c.rows + kMaxChannels * kRowsGranules;
  c.launch_id = ++ws.launches;  // nonzero, different for every launch on this workspace
  for (int k = 0; k < kMaxChannels; ++k)
    for (int i = 0; i < kM; ++i) c.p0[k][i] = a.p[(k < a.K ? k : 0) * kM + i];
  for (int i = 0; i < kM; ++i) {
    c.lb[i] = a.lb ? a.lb[i] : 0.0;
    c.ub[i] = a.ub ? a.ub[i] : 0.0;
    c.dscl[i] = a.dscl ? a.dscl[i] : 1.0;
  }
  for (int i = 0; i < 5; ++i) c.opts[i] = a.opts ? a.opts[i] : 0.0;
  c.itmax = a.itmax;
  c.has_opts = a.opts != nullptr;
  c.has_lb = a.lb != nullptr;
  c.has_ub = a.ub != nullptr;
  c.has_dscl = a.dscl != nullptr;
  c.want_covar = a.covar != nullptr;
  c.multi = pg_candidates();
  c.analytic = a.method == 2 ? 1 : 0;
  c.spec_jac = bc_spec_jac_enabled() ? 1 : 0;
  c.mbox = ws.d_mbox;
  c.n = a.n;
  c.K = a.K;
  for (int k = 0; k < kMaxChannels; ++k) c.tag_base[k] = ws.tag_base[k];
  c.spin_ticks = kSpinBudgetTicks;
  c.replicas = kReplicas;
  if (const char *e = getenv("BRDF_HIP_RESIDENT_REPLICAS")) c.replicas = std::min(kReplicas, std::max(1, atoi(e)));
  if (const char *e = getenv("BRDF_HIP_RESIDENT_SPIN_MS")) c.spin_ticks = std::max(1LL, atoll(e)) * 100000LL;
  c.sabotage_epoch = -1;
  if (const char *e = getenv("BRDF_HIP_RESIDENT_SABOTAGE")) c.sabotage_epoch = atoi(e);  // tests only: forces the fallback
  {  // one workgroup per CU must be able to live there at all (registers, LDS): checked once per kernel
    static int per_cu = -1;
    if (per_cu < 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, channels_fit_kernel<MODEL, FAST>, kRThreads, 0) != hipSuccess) per_cu = 0;
    if (per_cu < 1) {
      *unavailable = true;
      return 0;
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  ws.timer.before(a.stream);
  hipLaunchKernelGGL((channels_fit_kernel<MODEL, FAST>), dim3(G), dim3(kRThreads), 0, a.stream, c);
  HIP_OK(hipGetLastError());
  ws.timer.after(a.stream);
  {  // wait on the pinned mailboxes; the launch always terminates (bounded spins), which hipStreamQuery reports
    auto all_done = [&] {
      for (int k = 0; k < a.K; ++k)
        if (!*(volatile int *)&ws.h_mbox[k].done) return false;
      return true;
    };
    for (unsigned spins = 0; !all_done(); ++spins)
      if ((spins & 0x3FFu) == 0x3FFu && hipStreamQuery(a.stream) != hipErrorNotReady) break;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (!all_done()) {
      HIP_OK(hipStreamSynchronize(a.stream));
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (!all_done()) {  // aborted: not co-resident / spin budget exhausted.  Tags of unknown epochs were stored: start over
      (void)hipMemsetAsync(ws.d_block, 0, CWorkspace::block_bytes, a.stream);
      for (int k = 0; k < kMaxChannels; ++k) ws.tag_base[k] = 0;
      *unavailable = true;
      return 0;
    }
  }
  ws.launch_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
#ifdef BRDF_STAMPS
  HIP_OK(hipStreamSynchronize(a.stream));  // (the sweeping waves write their stamps when they leave, after the last channel's result)
#endif
  int worst = 0;
  bool bad_domain = false;
  for (int k = 0; k < a.K; ++k) {
    const Mailbox &mb = ws.h_mbox[k];
    ws.tag_base[k] += (unsigned)mb.passes + 2u;
    bad_domain = bad_domain || (FAST && mb.domain_bad);
  }
  if (bad_domain) {
    *retry_exact = true;
    return 0;
  }
  for (int k = 0; k < a.K; ++k) {
    const Mailbox &mb = ws.h_mbox[k];
    for (int i = 0; i < kM; ++i) a.p[k * kM + i] = mb.p[i];
    if (a.info)
      for (int i = 0; i < kInfoSz; ++i) a.info[k * kInfoSz + i] = mb.info[i];
    if (a.covar)
      for (int i = 0; i < kM * kM; ++i) a.covar[k * kM * kM + i] = mb.covar[i];
    ws.stats[k].passes = mb.passes;
    ws.stats[k].launches = 1;
    ws.stats[k].jac_passes = mb.n_jac;
    ws.stats[k].eval_passes = mb.n_eval;
    ws.stats[k].device_us = (double)(mb.t_last - mb.t_first) / 100.0;
    ws.stats[k].kernel_us = k == 0 ? ws.timer.elapsed_us() : ws.stats[0].kernel_us;  // (the shared launch's)
    for (int i = 0; i < 8; ++i) ws.stats[k].stamps[i] = mb.stamps[i];
    if (mb.ret < 0) worst = kLmError;
  }
  return worst;
}

template <int MODEL>
int channels_run_m(const ChannelsArgs &a, CWorkspace &ws, bool *unavailable) {
  bool retry = false;
  double keep[kMaxChannels * kM];
  for (int i = 0; i < a.K * kM; ++i) keep[i] = a.p[i];
  if (brdf_fast_path_enabled() || MODEL == MODEL_WARD) {
    const int ret = channels_attempt<MODEL, true>(a, ws, &retry, unavailable);
    if (!retry || *unavailable) return ret;
    for (int i = 0; i < a.K * kM; ++i) a.p[i] = keep[i];
  }
  if constexpr (MODEL != MODEL_WARD)
    return channels_attempt<MODEL, false>(a, ws, &retry, unavailable);
  else
    return kLmError;
}

// one translation unit per MODEL: channels_inst.hip
#define BRDF_CHANNELS_INSTANCE(MODEL_, NAME_) \
  int channels_run_##NAME_(const ChannelsArgs &a, CWorkspace &ws, bool *unavailable) { return channels_run_m<MODEL_>(a, ws, unavailable); }

}  // namespace brdf
