// resident_fit.hip -- "resident" regime: ONE launch per fit, the samples never leave the chip.
//
// For fits that fit the chip's register file + LDS (n <= #CUs * 4096 samples, i.e. 1,048,576 on MI355X) the whole
// fit runs inside one launch of #CUs workgroups (one 512-thread workgroup per CU, all co-resident).  Every workgroup
// reads its tile of the sample planes from HBM exactly once and keeps it for the rest of the fit:
//
//   waves 1..7, "sample waves"         per lane 10 samples in registers: c0, x, the two per-sample invariants
//                                      (brdf_models.h: Prep) and, for dlevmar_dif, f(p) and f(p+Dp) (lm_core.c:551, :742)
//   LDS (dlevmar_dif only, 105 KiB)    the secant Jacobian rows, three SoA planes            (lm_core.c:759-769)
//   wave 0, "control wave"             no samples: it runs the exchange and the serial LM step, so the step's ~150
//                                      live registers never compete with the resident samples (a symmetric
//                                      eight-wave version spilled 40-120 VGPRs and lost to the launch chain)
//
// A pass (= one LM evaluation: e=x-hx / ||e||^2, FD Jacobian, J^T J / J^T e, Broyden update, brdfdata.cpp:975-988 +
// misc_core.c:153-171 + lm_core.c:617-653) therefore moves no sample bytes at all.  Passes are separated by an
// in-launch exchange of the per-workgroup partial sums instead of a kernel boundary:
//
//   sample waves : sweep -> reduction over the seven waves -> <= 14 partial sums in LDS            (barriers X1, X2)
//   control wave : publishes them as tagged 8-byte granules {tag : 32, half of a double : 32}, each ONE write-through
//                  (sc1) store; gathers everybody's in two levels (control_exchange below); folds in a fixed order;
//                  steps ITS OWN copy of the LM state machine (lm_machine.h) -- the same redundant execution as in
//                  the launch chain of stream_fit.hip, so there is no broadcast hop; builds the next pass's uniforms
//                                                                                                       (barrier B)
//
// The granules are recipe R2 of the CDNA guide (cdna_hip_programming.md, Guideline 16: "the data IS the flag"): a
// granule is one naturally aligned 8-byte word written by one store, so it cannot tear; no flag, no fence, no ordering
// between granules is needed.  Tags are (launch base + epoch + 1): the tables are never zeroed between fits.  They
// are double-buffered by epoch parity: a workgroup can be at most one epoch ahead of the slowest one (it needs
// everybody's row of epoch e+1 before it can publish epoch e+2), so two buffers suffice.  Results do not depend on
// dispatch order or XCD placement (fold order = workgroup index).  Every spin is bounded by a wall-clock budget: if
// the grid is not co-resident (or anything else goes wrong) all workgroups drain, the launch ends with ctl->abort set
// and the host falls back to the launch chain (tests/test_gpu_parity.py exercises that path by sabotage).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "batch_fit.h"
#include "stream_fit.h"

namespace brdf {

constexpr int kRThreads = 512;               // wave 0 = control wave, waves 1..7 = sample waves
constexpr int kRWorkers = kRThreads - kWave;  // 448 lanes hold samples
constexpr int kRSpt = 10;                     // samples per worker lane (448 * 10 = 4480 >= 4096)
constexpr int kRTile = 4096;                  // samples per workgroup (so that #CUs * kRTile >= 2^20 on MI355X)
constexpr int kRCap = kRWorkers * kRSpt;      // sample slots per workgroup
constexpr int kRowWords = 2 * kSlots;         // 8-byte granules per partial row: 2 per slot
constexpr int kRowStride = 256;               // workgroups per granule word (>= #CUs), a whole number of lines
constexpr int kRedCols = kRWorkers / 4;       // reduction buffer columns (after two in-row DPP steps)
constexpr long long kSpinBudgetTicks = 200000000LL;  // default budget per wait: 2 s of s_memrealtime (100 MHz)

typedef unsigned long long u64;

struct ResidentCtl {  // zeroed before every launch (uploaded together with the machine)
  unsigned abort;
  unsigned domain_bad;
  unsigned pad[30];
};

struct ResidentCtx {
  const double *c0, *c1, *c2, *x;
  u64 *rows;            // [2][kRowWords][kRowStride] tagged granules; every tag stored so far is <= tag_base
  u64 *groups;          // [2][kRowWords][kGroupStride]: sums over groups of 16 workgroups, same granule format
  ResidentCtl *ctl;
  const void *machine0;  // DifMachine<3> / BcMachine<3> as started by the host
  Mailbox *mbox;
  int n;
  unsigned tag_base;  // tags of this launch are tag_base + epoch + 1: the rows need no zeroing between launches
  long long spin_ticks;  // budget of one wait (s_memrealtime ticks)
  int sabotage_epoch;    // test hook (BRDF_HIP_RESIDENT_SABOTAGE): the last workgroup withholds its row at this epoch; -1 = never
};

// METHOD 0 dlevmar_dif, 1 dlevmar_bc_dif / bc_der, 2 dlevmar_der (analytic Jacobian, lm_core.c:64-432)
template <int METHOD>
using RMachine = typename std::conditional<METHOD == 0, DifMachine<kM>, typename std::conditional<METHOD == 1, BcMachine<kM>, DerMachine<kM>>::type>::type;

__device__ __forceinline__ void put_value(u64 *rows, int stride, int word, int col, unsigned tag, double v) {
  const u64 bits = (u64)__double_as_longlong(v);
  __hip_atomic_store(rows + (size_t)word * stride + col, ((u64)tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(rows + (size_t)(word + 1) * stride + col, ((u64)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 get_granule(const u64 *g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double join_halves(u64 lo, u64 hi) { return __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32))); }

// The sweeps of this kernel are bound by fp64 issue, not by memory, so the accumulations are written as fused
// multiply-adds (one instruction and one rounding instead of two; the library is otherwise built with
// -ffp-contract=off so that lm_machine.h performs the reference's operations one by one).
__device__ __forceinline__ void acc_normal_eq_fma(const double *j, double e, double *jtj6, double *jte3) {
  jtj6[0] = fma(j[0], j[0], jtj6[0]);
  jtj6[1] = fma(j[0], j[1], jtj6[1]);
  jtj6[2] = fma(j[1], j[1], jtj6[2]);
  jtj6[3] = fma(j[0], j[2], jtj6[3]);
  jtj6[4] = fma(j[1], j[2], jtj6[4]);
  jtj6[5] = fma(j[2], j[2], jtj6[5]);
  jte3[0] = fma(j[0], e, jte3[0]);
  jte3[1] = fma(j[1], e, jte3[1]);
  jte3[2] = fma(j[2], e, jte3[2]);
}
// x / d given r = RN(1/d): RN(x*r) followed by one correction step is the correctly rounded quotient (Markstein's
// theorem; the operands here are normal numbers) -- 3 instructions instead of the ~15 of a full fp64 division, for a
// divisor that is the same for every sample of a pass (||Dp||^2, lm_core.c:763)
__device__ __forceinline__ double div_by(double x, double d, double r) {
  const double q = x * r;
  return fma(fma(-q, d, x), r, q);
}

#ifdef BRDF_STAMPS
#define RSTAMP(i) do { const long long now_ = clock64(); st_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define RSTAMP(i) do {} while (0)
#endif

// Reduction of NS sums and one max over the seven sample waves (executed by those waves only; the control wave
// joins the two barriers): two DPP steps inside each row of 16 lanes leave the sum of every 4 consecutive lanes in
// lanes 3,7,11,..; those park their values as buf[slot][worker/4]; sample wave w (1..7) then owns slots {w-1, w+6}:
// each lane adds its entries and one DPP tree per slot finishes it.  A pure function of NS: reproducible.
template <int NS>
__device__ __forceinline__ void worker_reduce(const double *acc, double mx, double *buf, double *out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = (threadIdx.x >> 6) - 1;       // 0..6
  const int wt = threadIdx.x - kWave;            // 0..447
  double v[NS + 1];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double t = acc[k];
    t = t + dpp_move<0x111, 0xf, 0xf>(t, 0.0);  // row_shr:1
    t = t + dpp_move<0x112, 0xf, 0xf>(t, 0.0);  // row_shr:2
    v[k] = t;
  }
  {
    double t = mx;
    t = fmax(t, dpp_move<0x111, 0xf, 0xf>(t, 0.0));
    t = fmax(t, dpp_move<0x112, 0xf, 0xf>(t, 0.0));
    v[NS] = t;
  }
  if ((wt & 3) == 3) {
#pragma unroll
    for (int k = 0; k <= NS; ++k) buf[k * kRedCols + (wt >> 2)] = v[k];
  }
  __syncthreads();  // X1
  for (int k = wave; k <= NS; k += kRWorkers / kWave) {  // wave-uniform loop
    const double *src = buf + k * kRedCols;
    const bool two = lane + kWave < kRedCols;
    double s = src[lane];
    if (k < NS) {
      s += two ? src[lane + kWave] : 0.0;
      s = wave_reduce_to_last<OpSum>(s);
      if (lane == kWave - 1) out[k] = s;
    } else {
      s = fmax(s, two ? src[lane + kWave] : 0.0);
      s = wave_reduce_to_last<OpMax>(s);
      if (lane == kWave - 1) out[kSums] = s;
    }
  }
  __syncthreads();  // X2
}

// The exchange, executed by the control wave: a two-level gather.  (A flat all-gather -- every workgroup reading all
// 256 rows -- was measured at 6.2 us: 256 readers per line make the few hundred lines of the row table a hot spot
// of the memory side; more loads in flight per reader made it slower, not faster.)
//
//   level 1  rows [parity][word][workgroup], word = 2*slot + half: every workgroup publishes its NS sums + max
//            (lanes 0..NS-1 and lane 13, two granules each).  Workgroups are grouped 16 by 16; the first one of a group
//            is its leader: lanes 0..15 of its control wave each gather one member's row (a wave load instruction
//            reads 16 consecutive words = ONE line), fold the 16 rows with a DPP row reduction (fixed order) and
//            lane 15 publishes the group's sums into
//   level 2  groups [parity][word][group]: 16 groups x 8 B = one line per word.  Every workgroup (leaders too)
//            gathers the <=16 group rows the same way and folds them: identical bits everywhere.
//
// On level 2 (256 readers per line) a lane first probes ONE word of its row (the last one its producer stores) and
// only then loads the row, and waits between probes: the lines being polled are the ones the producers have to
// write.  On level 1 (one reader per line) the row is simply re-read until all its tags match.  false = wait abandoned.
constexpr int kGroup = 16;
constexpr int kGroupStride = 16;  // group rows per granule word (>= ceil(#CUs / kGroup)), one 128-B line

template <int NS>
__device__ __forceinline__ bool gather_row(const ResidentCtx &ctx, const u64 *g, int stride, unsigned tag, bool probe, bool active,
                                           double (&pv)[NS], double &pmx) {
#pragma unroll
  for (int k = 0; k < NS; ++k) pv[k] = 0.0;
  pmx = 0.0;
  bool failed = false;
  if (active) {
    const long long t0 = (long long)wall_clock64();
    for (unsigned spins = 0;; ++spins) {
      if (!probe || (unsigned)(get_granule(g + (size_t)(2 * kSums + 1) * stride) >> 32) == tag) {
        u64 w[2 * NS + 2];
#pragma unroll
        for (int k = 0; k < 2 * NS; ++k) w[k] = get_granule(g + (size_t)k * stride);
        w[2 * NS] = get_granule(g + (size_t)(2 * kSums) * stride);
        w[2 * NS + 1] = get_granule(g + (size_t)(2 * kSums + 1) * stride);
        bool ready = true;
#pragma unroll
        for (int k = 0; k < 2 * NS + 2; ++k) ready = ready && ((unsigned)(w[k] >> 32) == tag);
        if (ready) {
#pragma unroll
          for (int k = 0; k < NS; ++k) pv[k] = join_halves(w[2 * k], w[2 * k + 1]);
          pmx = join_halves(w[2 * NS], w[2 * NS + 1]);
          break;
        }
      }
      if ((spins & 63u) == 63u) {
        if (__hip_atomic_load(&ctx.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
            (long long)wall_clock64() - t0 > ctx.spin_ticks) {
          __hip_atomic_store(&ctx.ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          failed = true;
          break;
        }
      }
      if (probe)
        __builtin_amdgcn_s_sleep(2);  // many readers of one line: leave the line to its producers between probes
      else
        __builtin_amdgcn_s_sleep(1);
    }
  }
  return !__any(failed);
}

template <int NS>
__device__ __forceinline__ bool control_exchange(const ResidentCtx &ctx, unsigned epoch, double *sums, int *s_abort,
                                                 long long *st_, long long &last_) {
  const int lane = threadIdx.x;  // control wave = wave 0
  const int G = gridDim.x;
  const int grp = blockIdx.x / kGroup, ngrp = (G + kGroup - 1) / kGroup;
  const unsigned tag = ctx.tag_base + epoch + 1u;
  u64 *rows = ctx.rows + (size_t)(epoch & 1u) * kRowWords * kRowStride;
  u64 *groups = ctx.groups + (size_t)(epoch & 1u) * kRowWords * kGroupStride;
  const bool withhold = (int)epoch == ctx.sabotage_epoch && blockIdx.x == gridDim.x - 1;  // test hook, see ResidentCtx
  if (lane < NS && !withhold)
    put_value(rows, kRowStride, 2 * lane, blockIdx.x, tag, sums[lane]);
  else if (lane == kSums && !withhold)
    put_value(rows, kRowStride, 2 * kSums, blockIdx.x, tag, sums[kSums]);

  double pv[NS], pmx;
  if (blockIdx.x % kGroup == 0) {  // group leader (workgroup-uniform branch)
    const int m = grp * kGroup + lane;
    if (!gather_row<NS>(ctx, rows + m, kRowStride, tag, /*probe=*/false, lane < kGroup && m < G, pv, pmx)) {
      *s_abort = 1;
      return false;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double t = row_reduce_to_last<OpSum>(pv[k]);
      if (lane == kGroup - 1) put_value(groups, kGroupStride, 2 * k, grp, tag, t);
    }
    const double t = row_reduce_to_last<OpMax>(pmx);
    if (lane == kGroup - 1) put_value(groups, kGroupStride, 2 * kSums, grp, tag, t);
  }
  RSTAMP(2);
  if (!gather_row<NS>(ctx, groups + lane, kGroupStride, tag, /*probe=*/true, lane < ngrp, pv, pmx)) {
    *s_abort = 1;
    return false;
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double t = row_reduce_to_last<OpSum>(pv[k]);
    if (lane == kGroup - 1) sums[k] = t;
  }
  {
    const double t = row_reduce_to_last<OpMax>(pmx);
    if (lane == kGroup - 1) sums[kSums] = t;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  RSTAMP(3);
  return true;
}

// BATCHED = false: one fit spread over the grid (ctx).  BATCHED = true: one workgroup per fit of 1024 < n <= 4096 samples
// (bctx, batch_fit.h) -- the same control wave / sample waves / LDS Jacobian, no exchange between workgroups, the
// speculative dlevmar_dif protocol (one pass per LM iteration), the machine started on the device.
template <int MODEL, int METHOD, bool FAST, bool BATCHED>
__global__ __launch_bounds__(kRThreads) void resident_fit_kernel(ResidentCtx ctx, BatchCtx bctx) {
  using Machine = RMachine<METHOD>;
  using Mdl = BrdfModel<MODEL>;
  static_assert(sizeof(Machine) % 4 == 0, "machine copied as dwords");
  __shared__ Machine sm;
  __shared__ PassUniforms<MODEL> su;
  __shared__ double red[kSlots * kRedCols];
  __shared__ double sums[kSlots];
  __shared__ double dp_prev[kM + 1];  // Dp and ||Dp||^2 of the last trial (dif)
  __shared__ int s_abort, s_bad;
  constexpr int kJl = (METHOD == 0) ? 3 * kRCap : 2;
  __shared__ double jl[kJl];  // dif: the secant Jacobian, SoA planes

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int G = BATCHED ? 1 : (int)gridDim.x;
  const int n = BATCHED ? bctx.n : ctx.n;
  const int fit = blockIdx.x;  // BATCHED only
  if constexpr (BATCHED && !FAST) {
    if (bctx.flags[fit] != kNeedsExact) return;  // exact kernel: only the fits the fast kernel declined
  }

  if constexpr (BATCHED) {  // the machine is started here (all 64 lanes of the control wave, identical values)
    if (wave == 0) {
      const double *p0 = bctx.p + (size_t)fit * kM;
      const double *opts = bctx.has_opts ? bctx.opts : nullptr;
      if constexpr (METHOD == 0)
        sm.start(p0, n, bctx.itmax, opts, 0, /*speculative=*/1);
      else
        sm.start(p0, n, bctx.has_lb ? bctx.lb : nullptr, bctx.has_ub ? bctx.ub : nullptr, nullptr, bctx.itmax, opts, 0, bctx.multi);
    }
  } else {  // the started machine, written by the host before the launch
    const unsigned *src = reinterpret_cast<const unsigned *>(ctx.machine0);
    unsigned *dst = reinterpret_cast<unsigned *>(&sm);
    for (int w = tid; w < (int)(sizeof(Machine) / 4); w += kRThreads) dst[w] = src[w];
  }
  if (tid == 0) s_abort = s_bad = 0;
  if (tid <= kM) dp_prev[tid] = 0.0;
  __syncthreads();
  if (wave == 0) {
    if constexpr (METHOD == 1)
      su.build(sm.h.req, true, sm.c.analytic_jac != 0);
    else
      su.build(sm.h.req, true, METHOD == 2);
  }
  __syncthreads();

  if (wave == 0) {
    // =========================== control wave: exchange, fold, LM step ===========================================
    // All 64 lanes execute the scalar step with identical values (stream_fit.hip explains why that beats one lane).
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long last_ = clock64();
    long long n_jac = 0;
    const long long t_first = (long long)wall_clock64();
    unsigned epoch = 0;
    for (;; ++epoch) {
      const int kind = sm.h.req.kind;
      if (kind == RQ_DONE) break;
      if (kind == RQ_JAC || kind == RQ_DIF_JAC) ++n_jac;
      __syncthreads();  // X1 (worker_reduce)
      __syncthreads();  // X2: sums[] hold this workgroup's partial sums
      RSTAMP(1);
      bool alive = true;
      if constexpr (BATCHED && FAST) {  // the sample waves looked at the cosines while loading the tile (before X1)
        if (epoch == 0) {
          if (s_bad) {  // log of a non-positive cosine: leave this fit to the exact kernel
            if (tid == 0) bctx.flags[fit] = kNeedsExact;
            s_abort = 1;
            alive = false;
          } else if (tid == 0) {
            bctx.flags[fit] = 0;
          }
        }
      }
      if constexpr (!BATCHED) {  // (a batched fit is one workgroup: sums[] already hold everything)
        if constexpr (METHOD == 0) {
          switch (kind) {
          case RQ_DIF_JAC: alive = control_exchange<SumLayout<kM>::DIF_JAC>(ctx, epoch, sums, &s_abort, st_, last_); break;
          case RQ_DIF_TRIAL: alive = control_exchange<SumLayout<kM>::DIF_TRIAL>(ctx, epoch, sums, &s_abort, st_, last_); break;
          default: alive = control_exchange<1>(ctx, epoch, sums, &s_abort, st_, last_); break;
          }
        } else {
          switch (kind) {
          case RQ_JAC: alive = control_exchange<SumLayout<kM>::JAC>(ctx, epoch, sums, &s_abort, st_, last_); break;
          case RQ_EVAL_MULTI: alive = control_exchange<kMaxCand>(ctx, epoch, sums, &s_abort, st_, last_); break;
          default: alive = control_exchange<1>(ctx, epoch, sums, &s_abort, st_, last_); break;
          }
        }
      }
      if (!alive) {  // give up: the host sees no `done`, reads ctl->abort and falls back
        __syncthreads();  // B (the sample waves read s_abort behind it)
        return;
      }
      if (kind == RQ_DIF_TRIAL) {
#pragma unroll
        for (int j = 0; j < kM; ++j) dp_prev[j] = su.dp[j];
        dp_prev[kM] = su.dp_l2;
      }
      sm.template step<true>(sums, sums[kSums]);
      if (sm.h.req.kind != RQ_DONE) {
        if constexpr (METHOD == 1)
          su.build(sm.h.req, /*need_base=*/false, sm.c.analytic_jac != 0);
        else
          su.build(sm.h.req, /*need_base=*/false, METHOD == 2);
      }
      __syncthreads();  // B: the next request and its uniforms are in LDS
      RSTAMP(4);
    }
    if constexpr (BATCHED) {
      if (tid == 0) {
        double *po = bctx.p + (size_t)fit * kM;
        for (int i = 0; i < kM; ++i) po[i] = sm.h.p[i];
        if (bctx.info)
          for (int i = 0; i < kInfoSz; ++i) bctx.info[(size_t)fit * kInfoSz + i] = sm.c.info[i];
        if (bctx.ret) bctx.ret[fit] = sm.c.ret;
      }
      return;
    }
    if (blockIdx.x == 0 && tid == 0) {  // every workgroup holds the same finished machine; workgroup 0 reports
      Mailbox *mb = ctx.mbox;
      mb->ret = sm.c.ret;
      mb->passes = (int)epoch;
      if constexpr (METHOD == 1)
        mb->infeasible_mask = sm.c.infeasible_mask;
      else
        mb->infeasible_mask = 0;
      mb->domain_bad = (int)__hip_atomic_load(&ctx.ctl->domain_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mb->n_jac = n_jac;
      mb->n_eval = (long long)epoch - n_jac;
      mb->t_first = t_first;
      mb->t_last = (long long)wall_clock64();
      for (int k = 0; k < 8; ++k) mb->stamps[k] = st_[k];
      for (int i = 0; i < kM; ++i) mb->p[i] = sm.h.p[i];
      for (int i = 0; i < kInfoSz; ++i) mb->info[i] = sm.c.info[i];
      for (int i = 0; i < kM * kM; ++i) mb->covar[i] = sm.c.covar[i];
      __threadfence_system();
      __hip_atomic_store(&mb->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }

  // ============================= sample waves: the resident tile and the sweeps ================================
  const int wt = tid - kWave;  // 0..447
  int vb = BATCHED ? 0 : (int)blockIdx.x;  // same XCD-contiguous dealing of tiles as the launch chain
  if (!BATCHED && (G & 7) == 0) vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int tile = (n + G - 1) / G;  // <= kRTile, checked on the host
  const int begin = vb * tile;
  const int end = min(n, begin + tile);
  const double *pc0 = BATCHED ? bctx.angles + (size_t)fit * 3 * n : ctx.c0;
  const double *pc1 = BATCHED ? pc0 + n : ctx.c1;
  const double *pc2 = BATCHED ? pc0 + 2 * (size_t)n : ctx.c2;
  const double *px = BATCHED ? bctx.x + (size_t)fit * n : ctx.x;
  const int nk = (tile + kRWorkers - 1) / kRWorkers;  // occupied sample slots of a lane (workgroup-uniform)
  double s0[kRSpt], sx[kRSpt];
  Prep pq[kRSpt];
  unsigned okm = 0;
  {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kRSpt; ++k) {
      const int i = begin + wt + k * kRWorkers;
      const bool ok = i < end;
      okm |= ok ? (1u << k) : 0u;
      const int ii = ok ? i : begin;
      s0[k] = pc0[ii];
      const double r1 = Mdl::uses_c1 ? pc1[ii] : 0.0;
      const double r2 = Mdl::uses_c2 ? pc2[ii] : 0.0;
      sx[k] = px[ii];
      pq[k] = Mdl::template prepare<FAST>(s0[k], r1, r2);
      if (FAST && ok && !Mdl::domain_ok(s0[k], r1, r2)) bad = true;
    }
    if constexpr (BATCHED) {
      if (FAST && bad) s_bad = 1;  // benign race: every writer stores 1
    } else {
      if (FAST && bad) __hip_atomic_store(&ctx.ctl->domain_bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  double hx[METHOD == 0 ? kRSpt : 1], wrk[METHOD == 0 ? kRSpt : 1];
  if constexpr (METHOD == 0) {
#pragma unroll
    for (int k = 0; k < kRSpt; ++k) hx[k] = wrk[k] = 0.0;
  }
  int cur_sel_hx = 0, cur_sel_j = 0;

  for (;;) {
    const int kind = sm.h.req.kind;
    if (kind == RQ_DONE) break;
    const PassUniforms<MODEL> &u = su;

    if constexpr (METHOD == 0) {  // commit what the machine decided about the previous trial (speculative protocol)
      if (sm.h.req.sel_j != cur_sel_j) {  // adopt the Broyden update J += ((wrk - hx - J Dp)/||Dp||^2) Dp^T, lm_core.c:760-766
        const double rinv = 1.0 / dp_prev[kM];
#pragma unroll
        for (int k = 0; k < kRSpt; ++k) if (k < nk) {
          const int s = k * kRWorkers + wt;
          const double jo[kM] = {jl[s], jl[kRCap + s], jl[2 * kRCap + s]};
          double t = 0.0;
#pragma unroll
          for (int l = 0; l < kM; ++l) t += jo[l] * dp_prev[l];
          t = div_by(wrk[k] - hx[k] - t, dp_prev[kM], rinv);
#pragma unroll
          for (int j = 0; j < kM; ++j) jl[j * kRCap + s] = jo[j] + t * dp_prev[j];
        }
        cur_sel_j = sm.h.req.sel_j;
      }
      if (sm.h.req.sel_hx != cur_sel_hx) {  // step accepted: hx <- f(p + Dp)
#pragma unroll
        for (int k = 0; k < kRSpt; ++k) hx[k] = wrk[k];
        cur_sel_hx = sm.h.req.sel_hx;
      }
    }

    double acc[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
    double mx = 0.0;
    switch (kind) {
    case RQ_EVAL:  // (the four kinds only dlevmar_bc_dif / bc_der issue are compiled into those kernels only)
      if constexpr (METHOD != 0) {
#pragma unroll
      for (int k = 0; k < kRSpt; ++k) if (k < nk) {
        const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double e = (okm >> k & 1u) ? sx[k] - f : 0.0;
        acc[0] = fma(e, e, acc[0]);
        mx = fmax(mx, fabs(e));
      }
      worker_reduce<1>(acc, mx, red, sums);
      }
      break;
    case RQ_SCALED:
      if constexpr (METHOD != 0) {
#pragma unroll
      for (int k = 0; k < kRSpt; ++k) if (k < nk) {
        const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double t = (okm >> k & 1u) ? (sx[k] - f) / u.scal : 0.0;
        acc[0] = fma(t, t, acc[0]);
      }
      worker_reduce<1>(acc, mx, red, sums);
      }
      break;
    case RQ_EVAL_MULTI:
      if constexpr (METHOD != 0) {
#pragma unroll
      for (int k = 0; k < kRSpt; ++k) if (k < nk) {
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
          if (j < u.ncand) {
            const double e = (okm >> k & 1u) ? sx[k] - model_value_k<MODEL, FAST>(u, j, s0[k], pq[k]) : 0.0;
            acc[j] = fma(e, e, acc[j]);
          }
      }
      worker_reduce<kMaxCand>(acc, mx, red, sums);
      }
      break;
    case RQ_JAC:
      if constexpr (METHOD != 0) {
#pragma unroll
      for (int k = 0; k < kRSpt; ++k) if (k < nk) {
        double f0 = 0.0, j[kM];
        if (u.analytic)  // dlevmar_bc_der with the model's analytic Jacobian
          model_an_row<MODEL, FAST>(u, s0[k], pq[k], f0, j);
        else
          model_fd_row<MODEL, FAST>(u, s0[k], pq[k], true, f0, 0.0, false, j);
        double e = sx[k] - f0;
        if (!(okm >> k & 1u)) e = j[0] = j[1] = j[2] = 0.0;
        acc_normal_eq_fma(j, e, acc, acc + kNL);
        acc[kNL + kM] = fma(e, e, acc[kNL + kM]);
      }
      worker_reduce<SumLayout<kM>::JAC>(acc, mx, red, sums);
      }
      break;
    case RQ_DIF_INIT:
      if constexpr (METHOD == 0) {
#pragma unroll
        for (int k = 0; k < kRSpt; ++k) if (k < nk) {
          hx[k] = model_value<MODEL, FAST>(u, s0[k], pq[k]);
          const double e = (okm >> k & 1u) ? sx[k] - hx[k] : 0.0;
          acc[0] = fma(e, e, acc[0]);
        }
        worker_reduce<1>(acc, mx, red, sums);
      }
      break;
    case RQ_DIF_JAC:
      if constexpr (METHOD == 0) {
#pragma unroll
        for (int k = 0; k < kRSpt; ++k) if (k < nk) {
          const int s = k * kRWorkers + wt;
          double f0 = 0.0, j[kM];
          model_fd_row<MODEL, FAST>(u, s0[k], pq[k], false, f0, hx[k], true, j);
          double e = sx[k] - hx[k];
          if (!(okm >> k & 1u)) e = j[0] = j[1] = j[2] = 0.0;
          jl[s] = j[0];
          jl[kRCap + s] = j[1];
          jl[2 * kRCap + s] = j[2];
          acc_normal_eq_fma(j, e, acc, acc + kNL);
        }
        worker_reduce<SumLayout<kM>::DIF_JAC>(acc, mx, red, sums);
      }
      break;
    case RQ_DIF_TRIAL:  // speculative protocol: the Broyden-updated row is formed for the sums only; J itself is
                        // updated (from wrk, hx) at the top of the next pass if the machine adopts it
      if constexpr (METHOD == 0) {
        const double rinv = 1.0 / u.dp_l2;
#pragma unroll
        for (int k = 0; k < kRSpt; ++k) if (k < nk) {
          const int s = k * kRWorkers + wt;
          const double w = model_value_q<MODEL, FAST>(u, s0[k], pq[k]);
          const double jo[kM] = {jl[s], jl[kRCap + s], jl[2 * kRCap + s]};
          double t = 0.0, jn[kM];  // broyden_row() with the division by ||Dp||^2 done by div_by()
#pragma unroll
          for (int l = 0; l < kM; ++l) t += jo[l] * u.dp[l];
          t = div_by(w - hx[k] - t, u.dp_l2, rinv);
#pragma unroll
          for (int j = 0; j < kM; ++j) jn[j] = jo[j] + t * u.dp[j];
          double en = sx[k] - w, eo = sx[k] - hx[k];
          if (!(okm >> k & 1u)) en = eo = jn[0] = jn[1] = jn[2] = 0.0;
          wrk[k] = w;
          acc[0] = fma(en, en, acc[0]);
          acc_normal_eq_fma(jn, en, acc + 1, acc + 1 + kNL);
          acc[1 + kNL + kM + 0] = fma(jn[0], eo, acc[1 + kNL + kM + 0]);
          acc[1 + kNL + kM + 1] = fma(jn[1], eo, acc[1 + kNL + kM + 1]);
          acc[1 + kNL + kM + 2] = fma(jn[2], eo, acc[1 + kNL + kM + 2]);
        }
        worker_reduce<SumLayout<kM>::DIF_TRIAL>(acc, mx, red, sums);
      }
      break;
    default:  // unknown request: keep the barrier protocol, the control wave will not survive it either
      worker_reduce<1>(acc, mx, red, sums);
      break;
    }
    __syncthreads();  // B: the control wave has stepped the machine
    if (s_abort) return;
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

struct RWorkspace {
  int device = -1, cus = 0;
  char *d_block = nullptr;  // ctl | machine | rows[2][kRowWords][kRowStride]
  char *h_block = nullptr;  // pinned staging of ctl (zeros) | started machine: ONE upload per fit
  Mailbox *h_mbox = nullptr, *d_mbox = nullptr;
  static constexpr size_t kMachineBytes = 4096;
  static constexpr size_t off_machine = sizeof(ResidentCtl);
  static constexpr size_t off_rows = off_machine + kMachineBytes;
  static constexpr size_t rows_bytes = sizeof(u64) * 2 * (size_t)kRowWords * (kRowStride + 16);  // rows + group rows
  unsigned tag_base = 0;
  FitStats stats{};

  int ensure(int dev) {
    if (device == dev && d_block) return 0;
    device = dev;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
    HIP_OK(hipMalloc(&d_block, off_rows + rows_bytes));
    HIP_OK(hipMemset(d_block, 0, off_rows + rows_bytes));
    tag_base = 0;
    HIP_OK(hipHostMalloc(&h_block, off_rows, hipHostMallocDefault));
    HIP_OK(hipHostMalloc(&h_mbox, sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void **)&d_mbox, h_mbox, 0));
    return 0;
  }
};
thread_local RWorkspace g_rws;

template <int MODEL, int METHOD, bool FAST>
int resident_attempt(const StreamFitArgs &a, RWorkspace &ws, bool *retry_exact, bool *unavailable) {
  using Machine = RMachine<METHOD>;
  static_assert(sizeof(Machine) <= 4096, "resident workspace layout");
  *retry_exact = *unavailable = false;
  const int G = (int)std::min<long long>(ws.cus, std::max<long long>(1, ((long long)a.n + 1023) / 1024));
  HIP_OK(hipStreamSynchronize(a.stream));  // (no-op on an idle stream) the pinned staging block is about to be rewritten
  memset(ws.h_block, 0, RWorkspace::off_rows);
  Machine &m = *reinterpret_cast<Machine *>(ws.h_block + RWorkspace::off_machine);
  if constexpr (METHOD == 0) {
    m.start(a.p, a.n, a.itmax, a.opts, a.covar != nullptr, /*speculative=*/1);
    if (m.h.req.kind == RQ_DONE) {
      set_error("dlevmar_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM);
      return kLmError;
    }
  } else if constexpr (METHOD == 2) {
    m.start(a.p, a.n, a.itmax, a.opts, a.covar != nullptr);
    if (m.h.req.kind == RQ_DONE) {
      set_error("dlevmar_der(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM);
      return kLmError;
    }
  } else {
    m.start(a.p, a.n, a.lb, a.ub, a.dscl, a.itmax, a.opts, a.covar != nullptr, pg_candidates());
    m.c.analytic_jac = a.analytic ? 1 : 0;
    if (m.h.req.kind == RQ_DONE) {
      switch (m.c.bad_input) {
      case 1: set_error("dlevmar_bc_dif(): cannot solve a problem with fewer measurements [%d] than unknowns [%d]", a.n, kM); break;
      case 2: set_error("dlevmar_bc_dif(): at least one lower bound exceeds the upper one"); break;
      default: set_error("dlevmar_bc_dif(): scaling constants should be positive"); break;
      }
      return kLmError;
    }
    if (FAST || !brdf_fast_path_enabled())  // (an exact re-run must not print the warning twice)
      for (int i = 0; i < kM; ++i)          // same warning as lmbc_core.c:516-520
        if (m.c.infeasible_mask & (1 << i))
          fprintf(stderr, "Warning: component %d of starting point not feasible in dlevmar_bc_dif()! [%g projected to %g]\n",
                  i, m.c.p_start[i], m.h.p[i]);
  }
  Mailbox &mb = *ws.h_mbox;
  memset(&mb, 0, sizeof mb);
  if (ws.tag_base > 0xF0000000u) {  // tag space nearly used up: start over from zeroed rows
    HIP_OK(hipMemsetAsync(ws.d_block + RWorkspace::off_rows, 0, RWorkspace::rows_bytes, a.stream));
    ws.tag_base = 0;
  }
  HIP_OK(hipMemcpyAsync(ws.d_block, ws.h_block, RWorkspace::off_machine + sizeof(Machine), hipMemcpyHostToDevice, a.stream));

  ResidentCtx c;
  c.c0 = a.d_angles;
  c.c1 = a.d_angles + a.n;
  c.c2 = a.d_angles + 2 * (size_t)a.n;
  c.x = a.d_x;
  c.ctl = reinterpret_cast<ResidentCtl *>(ws.d_block);
  c.rows = reinterpret_cast<u64 *>(ws.d_block + RWorkspace::off_rows);
  c.groups = c.rows + 2 * (size_t)kRowWords * kRowStride;
  c.machine0 = ws.d_block + RWorkspace::off_machine;
  c.mbox = ws.d_mbox;
  c.n = a.n;
  c.tag_base = ws.tag_base;
  c.spin_ticks = kSpinBudgetTicks;
  c.sabotage_epoch = -1;
  if (const char *e = getenv("BRDF_HIP_RESIDENT_SPIN_MS")) c.spin_ticks = std::max(1LL, atoll(e)) * 100000LL;
  if (const char *e = getenv("BRDF_HIP_RESIDENT_SABOTAGE")) c.sabotage_epoch = atoi(e);  // tests only: forces the fallback

  hipLaunchKernelGGL((resident_fit_kernel<MODEL, METHOD, FAST, false>), dim3(G), dim3(kRThreads), 0, a.stream, c, BatchCtx{});
  HIP_OK(hipGetLastError());
  {  // wait on the pinned mailbox (a stream synchronise sleeps and wakes up tens of microseconds late); the launch
     // always terminates (bounded spins), which hipStreamQuery reports even if `done` never comes
    volatile int *done = &mb.done;
    for (unsigned spins = 0; !*done; ++spins)
      if ((spins & 0x3FFu) == 0x3FFu && hipStreamQuery(a.stream) != hipErrorNotReady) break;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  if (!mb.done) {
    HIP_OK(hipStreamSynchronize(a.stream));
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  if (!mb.done) {  // aborted: not co-resident / spin budget exhausted.  Tags of unknown epochs were stored: start over
    (void)hipMemsetAsync(ws.d_block + RWorkspace::off_rows, 0, RWorkspace::rows_bytes, a.stream);
    ws.tag_base = 0;
    *unavailable = true;
    return 0;
  }
  ws.tag_base += (unsigned)mb.passes + 2u;
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
  ws.stats.jac_passes = mb.n_jac;
  ws.stats.eval_passes = mb.n_eval;
  ws.stats.device_us = (double)(mb.t_last - mb.t_first) / 100.0;  // first pass start -> result (s_memrealtime, 100 MHz)
  for (int k = 0; k < 8; ++k) ws.stats.stamps[k] = mb.stamps[k];
  return mb.ret;
}

template <int MODEL, int METHOD>
int resident_run_mm(const StreamFitArgs &a, RWorkspace &ws, bool *unavailable) {
  bool retry = false;
  double keep[kM];
  for (int i = 0; i < kM; ++i) keep[i] = a.p[i];
  int ret;
  if (brdf_fast_path_enabled() || MODEL == MODEL_WARD) {
    ret = resident_attempt<MODEL, METHOD, true>(a, ws, &retry, unavailable);
    if (!retry || *unavailable) return ret;
    for (int i = 0; i < kM; ++i) a.p[i] = keep[i];
  }
  if constexpr (MODEL != MODEL_WARD)
    return resident_attempt<MODEL, METHOD, false>(a, ws, &retry, unavailable);
  else
    return kLmError;
}

}  // namespace

FitStats resident_fit_last_stats() { return g_rws.stats; }

namespace {
template <int MODEL, int METHOD>
int resident_batch_mm(bool fast, const BatchCtx &c, hipStream_t stream) {
  const dim3 grid(c.S), block(kRThreads);
  const ResidentCtx none{};
  if (fast) {
    hipLaunchKernelGGL((resident_fit_kernel<MODEL, METHOD, true, true>), grid, block, 0, stream, none, c);
    HIP_OK(hipGetLastError());
  }
  if constexpr (MODEL != MODEL_WARD) {  // fits with a cosine <= 0 marked themselves (or all are marked: exact mode)
    hipLaunchKernelGGL((resident_fit_kernel<MODEL, METHOD, false, true>), grid, block, 0, stream, none, c);
    HIP_OK(hipGetLastError());
  }
  return 0;
}
}  // namespace

int resident_batch_enqueue(int model, int method, bool fast, const BatchCtx &c, hipStream_t stream) {
  switch (model * 2 + method) {
  case 0: return resident_batch_mm<0, 0>(fast, c, stream);
  case 1: return resident_batch_mm<0, 1>(fast, c, stream);
  case 2: return resident_batch_mm<1, 0>(fast, c, stream);
  case 3: return resident_batch_mm<1, 1>(fast, c, stream);
  case 4: return resident_batch_mm<2, 0>(fast, c, stream);
  default: return resident_batch_mm<2, 1>(fast, c, stream);
  }
}

// returns true if the resident path handled the fit (*ret is then the solver's return value)
bool resident_fit_try(const StreamFitArgs &a, int *ret) {
  // Default for every single fit that fits the chip.  Measured on MI355X, 1M-sample Ward fit: 14.3 us per dlevmar_dif
  // pass against 19.7 us for the launch chain (the secant Jacobian no longer travels through HBM) and 10.7 us per
  // dlevmar_bc_dif pass against 11.3 us.  BRDF_HIP_RESIDENT=0: always the launch chain.
  const char *e = getenv("BRDF_HIP_RESIDENT");
  if (e && e[0] == '0') return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  RWorkspace &ws = g_rws;
  if (ws.ensure(dev) != 0) return false;
  if ((long long)a.n > (long long)ws.cus * kRTile || ws.cus > kRowStride) return false;  // does not fit the chip: launch chain
  bool unavailable = false;
  int r;
  switch (a.model * 3 + a.method) {
  case 0: r = resident_run_mm<0, 0>(a, ws, &unavailable); break;
  case 1: r = resident_run_mm<0, 1>(a, ws, &unavailable); break;
  case 2: r = resident_run_mm<0, 2>(a, ws, &unavailable); break;
  case 3: r = resident_run_mm<1, 0>(a, ws, &unavailable); break;
  case 4: r = resident_run_mm<1, 1>(a, ws, &unavailable); break;
  case 5: r = resident_run_mm<1, 2>(a, ws, &unavailable); break;
  case 6: r = resident_run_mm<2, 0>(a, ws, &unavailable); break;
  case 7: r = resident_run_mm<2, 1>(a, ws, &unavailable); break;
  default: r = resident_run_mm<2, 2>(a, ws, &unavailable); break;
  }
  if (unavailable) {
    static bool warned = false;
    if (!warned) fprintf(stderr, "libbrdf_hip: resident single-launch path unavailable (grid not co-resident?); using the launch chain\n");
    warned = true;
    return false;
  }
  *ret = r;
  return true;
}

}  // namespace brdf
