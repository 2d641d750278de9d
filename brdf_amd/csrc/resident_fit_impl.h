#pragma once
// resident_fit_impl.h -- kernels and per-instance host code of the resident regime; compiled once per (MODEL, METHOD) by
// resident_inst.hip (nine translation units: one 6.5-minute compile became nine of under a minute, in parallel).
//
// "resident" regime: ONE launch per fit, the samples never leave the chip.
//
// For fits that fit the chip's register file + LDS (n <= #CUs * 4096 samples, i.e. 1,048,576 on MI355X) the whole
// fit runs inside one launch of #CUs workgroups (one 512-thread workgroup per CU, all co-resident).  Every workgroup
// reads its tile of the sample planes from HBM exactly once and keeps it for the rest of the fit:
//
//   waves 1..7                         per lane 8 samples in registers: c0, x, the two per-sample invariants
//                                      (brdf_models.h: Prep) and, for dlevmar_dif, f(p), f(p+Dp) (lm_core.c:551, :742)
//                                      and the Broyden scalar of the last trial (lm_core.c:763)
//   wave 0, "control wave"             the same 8 samples per lane, but parked in LDS (28 KiB): it sweeps them from
//                                      there with a rolled loop, so that the ~250 live registers of the serial LM step it
//                                      also runs never compete with resident samples -- and all four SIMDs carry the
//                                      same sweep load (2 waves x 8 samples; the first version had seven sample waves
//                                      x 10 samples and an idle control wave: 20 sample-iterations on three SIMDs, 10 on
//                                      the fourth)
//   LDS (dlevmar_dif only, 96 KiB)     the secant Jacobian rows, three SoA planes            (lm_core.c:759-769)
//
// A pass (= one LM evaluation: e=x-hx / ||e||^2, FD Jacobian, J^T J / J^T e, Broyden update, brdfdata.cpp:975-988 +
// misc_core.c:153-171 + lm_core.c:617-653) therefore moves no sample bytes at all.  Passes are separated by an
// in-launch exchange of the per-workgroup partial sums instead of a kernel boundary:
//
//   all waves    : sweep -> reduction over the eight waves -> <= 14 partial sums in LDS            (barriers X1, X2)
//   control wave : publishes them as tagged 16-byte cells {lo32, tag, hi32, tag}, each ONE write-through (sc1)
//                  store; gathers everybody's in two levels (control_exchange below); folds in a fixed order;
//                  steps ITS OWN copy of the LM state machine (lm_machine.h) -- the same redundant execution as in
//                  the launch chain of stream_fit.hip, so there is no broadcast hop -- in LDS (a register copy of
//                  the machine's busy half was measured and lost, see the control wave's comment below); builds the
//                  next pass's uniforms                                                             (barrier B)
//
// The speculative dlevmar_dif protocol (lm_machine.h): a trial pass forms the Broyden-updated Jacobian row for the sums
// only and keeps the update's scalar t = (f(p+Dp) - f(p) - J Dp)/||Dp||^2 per sample; if the machine adopts the update,
// the NEXT trial pass applies J += t Dp^T on the fly while it reads the row anyway (the first version re-derived t in
// a separate commit sweep: ~1 us per pass).
//
// The cells are recipe R2 of the CDNA guide (cdna_hip_programming.md, Guideline 16: "the data IS the flag"): each
// 8-byte half {32 bits of the double, tag} is naturally aligned and carries its own tag, so a torn cell reads as "not
// ready yet"; no flag, no fence, no ordering between cells is needed.  Tags are (launch base + epoch + 1): the tables are never zeroed between fits.  They
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

#if defined(BRDF_STAMPS)
#include <hip/hip_runtime.h>
static __device__ long long g_rlm_stamps[8];  // diagnostic: cycles per section of the machines' run(), workgroup 0 only, summed over passes
static __device__ long long g_rlm_last;
#endif
#if defined(BRDF_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define LM_STAMP(i) do { if (blockIdx.x == 0) { const long long now_ = clock64(); if ((i) != 0) g_rlm_stamps[i] += now_ - g_rlm_last; g_rlm_last = now_; } } while (0)
#endif
#include "batch_fit.h"
#include "stream_fit.h"

namespace brdf {

constexpr int kRThreads = 512;               // eight waves, all of them sweep; wave 0 is also the control wave
constexpr int kRSpt = 8;                      // samples per lane
constexpr int kRTile = kRThreads * kRSpt;     // 4096 samples per workgroup (so that #CUs * kRTile >= 2^20 on MI355X)
constexpr int kRCap = kRTile;                 // sample slots per workgroup
constexpr int kRowWords = 2 * kSlots;         // 8-byte granules per partial row: 2 per slot
constexpr int kRowStride = 256;               // most workgroups of a launch (>= #CUs)
constexpr int kRedCols = kRThreads / 4;       // reduction buffer columns (after two in-row DPP steps)
constexpr long long kSpinBudgetTicks = 200000000LL;  // default budget per wait: 2 s of s_memrealtime (100 MHz)

typedef unsigned long long u64;

struct ResidentCtl {  // never zeroed between launches: a word is "set" when it equals the launch's id (ResidentCtx::launch_id)
  unsigned abort;
  unsigned domain_bad;
  unsigned pad[30];
};

struct ResidentCtx {
  const double *c0, *c1, *c2, *x;
  u64 *rows;            // [2][kMaxGroups][kRowWords][kGroup] tagged granules; every tag stored so far is <= tag_base
  u64 *groups;          // [2][kReplicas][kRowWords][kMaxGroups]: sums over groups of 16 workgroups, same granule format
  ResidentCtl *ctl;
  unsigned launch_id;    // nonzero, different for every launch on this workspace
  // what the machine is started from (single fits: every workgroup starts its own copy, as the batched kernels do; the
  // host starts one too, for the argument checks and warnings of the entry point, but nothing is uploaded)
  double p0[kM], opts[5], lb[kM], ub[kM], dscl[kM];
  int itmax, has_opts, has_lb, has_ub, has_dscl, want_covar, multi, analytic;
  int chain;     // dlevmar_dif: trial points per sweep in a chain of rejections (DifMachine::Cold::multi)
  int spec_jac;  // dlevmar_bc_dif / bc_der: candidates evaluated by Jacobian passes (BcMachine::Cold::spec_jac)
  Mailbox *mbox;
  int n;
  unsigned tag_base;  // tags of this launch are tag_base + epoch + 1: the rows need no zeroing between launches
  long long spin_ticks;  // budget of one wait (s_memrealtime ticks)
  int sabotage_epoch;    // test hook (BRDF_HIP_RESIDENT_SABOTAGE): the last workgroup withholds its row at this epoch; -1 = never
  int replicas;          // copies of the group rows in use (1..kReplicas): readers pick copy blockIdx % replicas
  long long *trace;      // diagnostic builds (-DBRDF_STAMPS): [workgroup][8] s_memrealtime stamps of epoch trace_epoch
  int trace_epoch;
};

// METHOD 0 dlevmar_dif, 1 dlevmar_bc_dif / bc_der, 2 dlevmar_der (analytic Jacobian, lm_core.c:64-432)
template <int METHOD>
using RMachine = typename std::conditional<METHOD == 0, DifMachine<kM>, typename std::conditional<METHOD == 1, BcMachine<kM>, DerMachine<kM>>::type>::type;

__device__ __forceinline__ double read_lane(double v, int lane) {  // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

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
// one epoch's timeline of EVERY workgroup (arrival skew, hop latencies): slot <- s_memrealtime (100 MHz) or a counter
#define RTRACE(ctx_, epoch_, slot_, val_) do { if ((ctx_).trace && (int)(epoch_) == (ctx_).trace_epoch && (threadIdx.x & 63) == 0) (ctx_).trace[blockIdx.x * 8 + (slot_)] = (long long)(val_); } while (0)
#else
#define RSTAMP(i) do {} while (0)
#define RTRACE(ctx_, epoch_, slot_, val_) do {} while (0)
#endif

// ---- where a lane's samples live ---------------------------------------------------------------------------------
// Fields: 0 c0, 1 x, 2 q1, 3 q2 (Ward only), and for dlevmar_dif 4 hx = f(p), 5 wrk = f(p+Dp), 6 tb = Broyden scalar.
// RegSamples: registers (every index a compile-time constant: the loops over k are fully unrolled).
// LdsSamples: the control wave's copy, [field][k][lane] in LDS, swept by a rolled loop.
constexpr int kFc0 = 0, kFx = 1, kFq1 = 2, kFq2 = 3, kFhx = 4, kFwrk = 5, kFtb = 6;
template <int METHOD>
constexpr int sample_fields() { return METHOD == 0 ? 7 : 4; }

template <int METHOD>
struct RegSamples {
  static constexpr bool kUnrolled = true;
  double v[sample_fields<METHOD>()][kRSpt];
  __device__ __forceinline__ double get(int f, int k) const { return v[f][k]; }
  __device__ __forceinline__ void set(int f, int k, double x) { v[f][k] = x; }
};
template <int METHOD>
struct LdsSamples {
  static constexpr bool kUnrolled = false;
  double *base;  // + lane
  __device__ __forceinline__ double get(int f, int k) const { return base[(f * kRSpt + k) * kWave]; }
  __device__ __forceinline__ void set(int f, int k, double x) { base[(f * kRSpt + k) * kWave] = x; }
};

template <bool UNROLLED, class F>
__device__ __forceinline__ void for_samples(int nk, F &&f) {
  if constexpr (UNROLLED) {
    // (one sample per guarded block.  Two per block -- so that the scheduler can interleave two samples' exp / Broyden chains
    // inside one basic block -- was measured: 28-36 more spilled VGPRs, 539 against 527 us per 10^6-sample dlevmar_dif fit)
#pragma unroll
    for (int k = 0; k < kRSpt; ++k)
      if (k < nk) f(k);
  } else {  // two samples per trip: one sample's dependent chain (exp, Broyden, 13 accumulations) alone leaves the pipe idle
    // (the eight slots unrolled for full tiles -- constant LDS offsets, the next sample's loads free to move up -- was measured: 33
    // spilled VGPRs, 469 against 428 us per 10^6-sample dlevmar_dif fit: this kernel has no register to spare)
#pragma unroll 2
    for (int k = 0; k < nk; ++k) f(k);
  }
}

// two samples per guarded block where both exist: one sample's dependent chains (exp, the row differences, ten accumulations)
// leave the fp64 pipe idle a third of the time with two waves per SIMD, two samples in one basic block let the scheduler
// interleave them.  For the kernels whose register budget has the room (everything but the dlevmar_dif trial sweep).
template <bool UNROLLED, class F>
__device__ __forceinline__ void for_sample_pairs(int nk, F &&f) {
  if constexpr (UNROLLED) {
#pragma unroll
    for (int k = 0; k < kRSpt; k += 2) {
      if (k + 1 < nk) {
        f(k);
        f(k + 1);
      } else if (k < nk) {
        f(k);
      }
    }
  } else {
#pragma unroll 2
    for (int k = 0; k < nk; ++k) f(k);
  }
}

__device__ __forceinline__ constexpr bool pend_flag(std::true_type) { return true; }
__device__ __forceinline__ constexpr bool pend_flag(std::false_type) { return false; }
__device__ __forceinline__ constexpr bool pend_flag(bool b) { return b; }

// Reduction of NS sums and one max over the eight waves: two DPP steps inside each row of 16 lanes leave the sum of
// every 4 consecutive lanes in lanes 3,7,11,..; those park their values as buf[slot][thread/4]; wave w then owns slots
// {w, w+8}: each lane adds its two entries and one DPP tree per slot finishes it.  A pure function of NS: reproducible.
// MAX = false (dlevmar_dif: its machine never looks at max |e|): no max slot -- one value less per wave, one 16-byte cell less
// per row and, with nine sums, one of the four gather instructions of an exchange
template <int NS, bool MAX = true>
__device__ __forceinline__ void worker_reduce(const double *acc, double mx, double *buf, double *out, long long *st_, long long &last_) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;  // 0..7
  double v[NS + 1];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double t = acc[k];
    t = t + dpp_move<0x111, 0xf, 0xf>(t, 0.0);  // row_shr:1
    t = t + dpp_move<0x112, 0xf, 0xf>(t, 0.0);  // row_shr:2
    v[k] = t;
  }
  if constexpr (MAX) {
    double t = mx;
    t = fmax(t, dpp_move<0x111, 0xf, 0xf>(t, 0.0));
    t = fmax(t, dpp_move<0x112, 0xf, 0xf>(t, 0.0));
    v[NS] = t;
  }
  constexpr int NV = NS + (MAX ? 1 : 0);  // values per lane
  if ((threadIdx.x & 3) == 3) {
#pragma unroll
    for (int k = 0; k < NV; ++k) buf[k * kRedCols + (threadIdx.x >> 2)] = v[k];
  }
  __syncthreads();  // X1
  RSTAMP(6);
  {  // wave w finishes slots w and w + 8 (wave-uniform conditions), the two dependent DPP trees interleaved
    constexpr int NW = kRThreads / kWave;
    const int k0 = wave, k1 = wave + NW;
    const bool has0 = k0 < NV, has1 = k1 < NV;
    double s0 = 0.0, s1 = 0.0;
    if (has0) {
      const double *src = buf + k0 * kRedCols;
      s0 = (k0 < NS || !MAX) ? src[lane] + src[lane + kWave] : fmax(src[lane], src[lane + kWave]);
    }
    if (has1) {
      const double *src = buf + k1 * kRedCols;
      s1 = (k1 < NS || !MAX) ? src[lane] + src[lane + kWave] : fmax(src[lane], src[lane + kWave]);
    }
    // (the max slot is slot NS: it is the LAST slot of whichever wave owns it, so at most one of the two is a max)
    const bool max0 = MAX && has0 && k0 == NS, max1 = MAX && has1 && k1 == NS;
    if (!max0 && !max1) {
      wave_reduce2_to_last<OpSum, OpSum>(s0, s1);
    } else if (max1) {
      wave_reduce2_to_last<OpSum, OpMax>(s0, s1);
    } else {
      wave_reduce2_to_last<OpMax, OpSum>(s0, s1);
    }
    if (lane == kWave - 1) {
      if (has0) out[max0 ? kSums : k0] = s0;
      if (has1) out[max1 ? kSums : k1] = s1;
    }
  }
  __syncthreads();  // X2
}

// The exchange, executed by the control wave: a two-level gather.  (A flat all-gather -- every workgroup reading all
// 256 rows -- was measured at 6.2 us: 256 readers per line make the few hundred lines of the row table a hot spot
// of the memory side; more loads in flight per reader made it slower, not faster.)
//
//   level 1  rows [parity][group][slot][member]: workgroups are grouped 16 by 16 and a group's 14 slots x 16 members
//            are 3.5 KB of CONSECUTIVE 16-byte cells.  Every workgroup publishes its NS sums + max (lanes 0..NS-1 and
//            lane 13, one cell each).  One workgroup of a group is its leader (a different
//            position in every group, so that the leaders -- blockIdx % 8 tells which workgroups share an XCD -- are spread
//            over the XCDs).  ALL 64 lanes of its control wave gather: lane l takes member l % 16 of slot 4j + l / 16,
//            j = 0..3 (a wave load instruction reads 1 KB = 8 whole lines; 4 instructions fetch a TRIAL row set), so the 16
//            members of a slot sit in one DPP row and four row reductions (fixed order) fold everything.  (The first
//            version kept ONE table [word][256 workgroups]: a leader's 16 active lanes issued 28 loads, 2 KB apart, per
//            attempt, and an attempt took ~1.5 us -- the in-kernel trace showed level 1 alone at 3.4 us.)
//   level 2  groups [parity][replica][slot][group]: the leader broadcasts each folded value inside its DPP row
//            (row_newbcast) and lanes 0..replicas-1 of the row store one copy each, so that a line is polled by #CUs /
//            replicas workgroups instead of all of them (one copy: +0.9 us).  Every workgroup (leaders too) gathers the
//            <= 16 group rows of copy blockIdx % replicas the same way and folds them: identical bits everywhere.
//
// A gather re-reads its block until every tag matches; false = wait abandoned (spin budget, or somebody else gave up).
constexpr int kGroup = 16;
constexpr int kMaxGroups = 16;  // >= ceil(#CUs / kGroup); also the width of a level-2 row
constexpr int kReplicas = 8;
constexpr size_t kBlockGranules = (size_t)kRowWords * kGroup;                    // one group's rows / one copy of the group rows
constexpr size_t kRowsGranules = 2 * (size_t)kMaxGroups * kBlockGranules;         // [parity][group]
constexpr size_t kGroupsGranules = 2 * (size_t)kReplicas * kBlockGranules;        // [parity][replica]

// A value travels as ONE 16-byte cell = two granules {lo32, tag | hi32, tag}, written by one write-through (sc1) store and
// read by one sc1 load: a scalar sc1 store is one fabric write whatever its size, so 8-byte stores cost 2.7x the time per
// byte of 16-byte ones (MI355X_MICROARCH.md, inter-workgroup visibility table) -- a leader publishes up to 14 x 8 cells
// per pass.  Each half carries its own tag, so a cell torn between its halves is simply "not ready yet".
// A block is [slot][16 columns] cells.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kCacheSc1 = 16;  // aux bits of the gfx940+ buffer instructions: sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const u64 *p, size_t granules) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<u64 *>(p), 0, (int)(granules * sizeof(u64)), 0x00027000);
}
__device__ __forceinline__ void put_cell(__amdgpu_buffer_rsrc_t t, unsigned cell, unsigned tag, double v) {
  const u32x4 d = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
  __builtin_amdgcn_raw_buffer_store_b128(d, t, (int)(cell * 16u), 0, kCacheSc1);
}

template <int NS, bool MAX = true>
__device__ __forceinline__ constexpr bool slot_used(int v) { return v < NS || (MAX && v == kSums); }

// val[j] <- slot 4j + lane/16 of column lane%16 of the block starting at cell `first` (0 where the slot is not used or
// the column >= ncols)
// (Ctx: ResidentCtx, or the per-channel view of channels_fit_impl.h -- anything with ctl, launch_id, spin_ticks)
template <int NS, bool MAX = true, class Ctx>
__device__ __forceinline__ bool gather_block(const Ctx &ctx, __amdgpu_buffer_rsrc_t t, unsigned first, unsigned tag, int ncols,
                                             double (&val)[4], unsigned *polls) {
  const int lane = threadIdx.x & (kWave - 1);  // a control wave's lane (the single-fit kernel's control wave is wave 0)
  const int col = lane & 15, r = lane >> 4;
  const long long t0 = (long long)wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    u32x4 d[4];
    bool ready = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d[j] = u32x4{0u, tag, 0u, tag};
      if ((4 * j < NS) || (MAX && 4 * j <= kSums && kSums < 4 * j + 4)) {  // (compile time) some row of this instruction is used
        const int v = 4 * j + r;
        if (slot_used<NS, MAX>(v) && col < ncols) d[j] = __builtin_amdgcn_raw_buffer_load_b128(t, (int)((first + v * kGroup + col) * 16u), 0, kCacheSc1);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) ready = ready && d[j].y == tag && d[j].w == tag;
    if (__all(ready)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) val[j] = __hiloint2double((int)d[j].z, (int)d[j].x);
      *polls = spins;
      return true;
    }
    if ((spins & 63u) == 63u) {  // (wave-uniform)
      if (__hip_atomic_load(&ctx.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ctx.launch_id ||
          (long long)wall_clock64() - t0 > ctx.spin_ticks) {
        __hip_atomic_store(&ctx.ctl->abort, ctx.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// folds the 16 columns of every slot: the total of slot 4j + r ends in lane 16r + 15 of val[j] (the max slot by max)
template <int NS, bool MAX>
__device__ __forceinline__ void fold_block(double (&val)[4]) {
  const int r = (int)(threadIdx.x & (kWave - 1)) >> 4;
  double mx = 0.0;
  if constexpr (MAX) mx = row_reduce_to_last<OpMax>(val[kSums / 4]);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if ((4 * j < NS) || (MAX && j == kSums / 4)) val[j] = row_reduce_to_last<OpSum>(val[j]);  // (compile time: rows in use)
  if (MAX && r == (kSums & 3)) val[kSums / 4] = mx;
}

template <int NS, bool MAX = true, class Ctx>
__device__ __forceinline__ bool control_exchange(const Ctx &ctx, unsigned epoch, double *sums, int *s_abort,
                                                 long long *st_, long long &last_) {
  const int lane = threadIdx.x & (kWave - 1);  // a control wave's lane
  const int col = lane & 15, r = lane >> 4;
  const int G = gridDim.x;
  const int grp = blockIdx.x / kGroup, ngrp = (G + kGroup - 1) / kGroup;
  const int members = min(kGroup, G - grp * kGroup);
  const int leader = grp * kGroup + (grp & 7) % members;
  const unsigned tag = ctx.tag_base + epoch + 1u;
  constexpr unsigned kBlockCells = kSlots * kGroup;
  const __amdgpu_buffer_rsrc_t rows = table_rsrc(ctx.rows, kRowsGranules), groups = table_rsrc(ctx.groups, kGroupsGranules);
  const unsigned my_rows = ((epoch & 1u) * kMaxGroups + grp) * kBlockCells;   // first cell of this group's block
  const unsigned my_groups = (epoch & 1u) * kReplicas * kBlockCells;           // first cell of copy 0 of the group rows
  const bool withhold = (int)epoch == ctx.sabotage_epoch && blockIdx.x == gridDim.x - 1;  // test hook, see ResidentCtx
  if (slot_used<NS, MAX>(lane) && lane <= kSums && !withhold) put_cell(rows, my_rows + lane * kGroup + blockIdx.x % kGroup, tag, sums[lane]);

  double val[4];
  unsigned polls = 0;
  if ((int)blockIdx.x == leader) {  // group leader (workgroup-uniform branch)
    if (!gather_block<NS, MAX>(ctx, rows, my_rows, tag, members, val, &polls)) {
      *s_abort = 1;
      return false;
    }
#ifndef BRDF_TRACE_WORKERS
    RTRACE(ctx, epoch, 6, polls + 1);
#endif
    fold_block<NS, MAX>(val);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double t = dpp_move<0x15F, 0xf, 0xf>(val[j], 0.0);  // row_newbcast:15: the row's total in all 16 lanes of the row
      const int v = 4 * j + r;
      if (slot_used<NS, MAX>(v) && col < ctx.replicas) put_cell(groups, my_groups + col * kBlockCells + v * kGroup + grp, tag, t);
    }
  }
  RSTAMP(2);
  RTRACE(ctx, epoch, 3, wall_clock64());
  if (!gather_block<NS, MAX>(ctx, groups, my_groups + (blockIdx.x % ctx.replicas) * kBlockCells, tag, ngrp, val, &polls)) {
    *s_abort = 1;
    return false;
  }
  fold_block<NS, MAX>(val);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = 4 * j + r;
    if (slot_used<NS, MAX>(v) && col == kGroup - 1) sums[v] = val[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  RSTAMP(3);
  RTRACE(ctx, epoch, 4, wall_clock64());
#ifndef BRDF_TRACE_WORKERS
  RTRACE(ctx, epoch, 7, polls);
#endif
  return true;
}

// ---- the uniforms of a request, built by the control wave --------------------------------------------------------------------
// PassUniforms::build() forms them one after the other on one lane's worth of values (every lane the same); each lin() / nl()
// hides an fp64 division or two, and a finite-difference Jacobian request needs seven parameter sets (p, p + d_j e_j, p - d_j e_j),
// a multi-candidate request up to eight: 2,500 shader-clock ticks of the 12 us a dlevmar_bc_dif pass takes.  Here lane l of the
// wave forms parameter set l -- the same operations on the same operands as build(), hence the same bits -- and stores its
// results where build() would.  Everything else goes to build().
template <int MODEL>
__device__ __forceinline__ void build_uniforms_wave(PassUniforms<MODEL> &u, const Request<kM> &r, bool need_base, bool analytic_jac) {
  using Mdl = BrdfModel<MODEL>;
  const int lane = (int)(threadIdx.x & (kWave - 1));
  if (r.kind == RQ_EVAL_MULTI) {
    u.ncand = r.nk;
    u.analytic = analytic_jac ? 1 : 0;
    if (lane < kMaxCand && lane < r.nk) {
      const double pk[kM] = {r.pk[lane][0], r.pk[lane][1], r.pk[lane][2]};
      u.lk[lane] = Mdl::lin(pk);
      u.nk[lane] = Mdl::nl(pk);
    }
    return;
  }
  if ((r.kind == RQ_JAC && !analytic_jac) || r.kind == RQ_DIF_JAC) {
    u.ncand = 0;
    u.analytic = 0;
    u.central = r.central;
    u.scal = r.scal;
    u.dp_l2 = r.dp_l2;
#pragma unroll
    for (int j = 0; j < kM; ++j) u.dp[j] = r.dp[j];
    if (lane < 1 + 2 * kM) {  // 0: p;  1..3: p + d_j e_j;  4..6: p - d_j e_j
      const int j = lane == 0 ? 0 : (lane - 1) % kM;
      const bool plus = lane >= 1 && lane <= kM, minus = lane > kM;
      double pp[kM];
#pragma unroll
      for (int i = 0; i < kM; ++i) {
        pp[i] = r.p[i];
        if (i == j && plus) pp[i] = r.p[i] + r.d[i];   // "p[j]+=d", misc_core.c:161 / "tmp+d", :202
        if (i == j && minus) pp[i] = r.p[i] - r.d[i];  // "p[j]-=d", misc_core.c:199
      }
      const Lin l = Mdl::lin(pp);
      const Nl nn = Mdl::nl(pp);
      if (lane == 0) {
        u.l0 = l;
        u.n0 = nn;
      } else if (plus) {
        u.lp[j] = l;
        if (j == kM - 1) u.np2 = nn;
        double dj = r.d[0];
#pragma unroll
        for (int i = 1; i < kM; ++i)
          if (i == j) dj = r.d[i];
        u.dinv[j] = (r.central ? 0.5 : 1.0) / dj;
      } else {
        u.lm[j] = l;
        if (j == kM - 1) u.nm2 = nn;
      }
    }
    return;
  }
  u.build(r, need_base, analytic_jac);
}

// ---- one pass over a lane's samples ------------------------------------------------------------------------------
// Executed by all eight waves: waves 1..7 on RegSamples (unrolled), the control wave on LdsSamples (rolled).  Leaves the
// lane's partial sums in acc[] / mx and returns the number of sum slots of the request kind (a wave-uniform value).
// `pend` (dlevmar_dif): the machine adopted the Broyden update of the previous trial; tb[] holds its scalar and
// dpp[] its Dp: J += tb Dp^T is applied to the row while it is being read (lm_core.c:760-766).
// PAIRS: two samples per guarded block in the evaluation / Jacobian sweeps (for_sample_pairs)
template <int MODEL, int METHOD, bool FAST, bool PAIRS, class Store>
__device__ __forceinline__ void sweep_pass(int kind, const PassUniforms<MODEL> &u, Store &st, double *jl, int tid, int nk, int nfull, unsigned okm,
                                           bool pend, const double *dpp, double *acc, double &mx) {
  constexpr bool U = Store::kUnrolled;
  constexpr bool W2 = BrdfModel<MODEL>::prep_planes == 2;
  auto prep = [&](int k) { return Prep{st.get(kFq1, k), W2 ? st.get(kFq2, k) : 0.0}; };
  // slots below nfull = (this workgroup's sample count) / 512 are full in every lane: the masking code sits behind a
  // workgroup-uniform branch and is skipped for them.  (nfull is the WORKGROUP's own count, not the tile's: the last
  // workgroup of a single fit holds up to #workgroups - 1 samples fewer than a tile, so its slot nk - 2 can be partly
  // empty and its slot nk - 1 entirely -- e.g. n = 262,145 on 256 CUs: tile 1025, last workgroup 770 samples.)
  auto dead = [&](int k) { return k >= nfull && !(okm >> k & 1u); };
  // (measured, production builds, 10^6-sample Ward dlevmar_bc_dif with every candidate a Jacobian pass: 178 us per fit with
  // pairs against 172 without -- the register-resident Jacobian body already interleaves its two exp chains; off by default)
#ifdef BRDF_BC_PAIRS
  constexpr bool kPairs = PAIRS;
#else
  constexpr bool kPairs = false;
#endif
  auto for_bc = [&](auto &&f) {
    if constexpr (kPairs)
      for_sample_pairs<U>(nk, f);
    else
      for_samples<U>(nk, f);
  };
  switch (kind) {
  case RQ_EVAL:  // (the four kinds only dlevmar_bc_dif / bc_der / der issue are compiled into those kernels only)
    if constexpr (METHOD != 0) {
      const EvalUniforms eu{scalar_copy(u.l0), scalar_copy(u.n0), 1.0};  // (scalar registers: see RQ_DIF_TRIAL)
      for_bc([&](int k) {
        const double f = model_value<MODEL, FAST>(eu, st.get(kFc0, k), prep(k));
        double e = st.get(kFx, k) - f;
        if (dead(k)) e = 0.0;
        acc[0] = fma(e, e, acc[0]);
        mx = fmax(mx, fabs(e));
      });
    }
    break;
  case RQ_SCALED:
    if constexpr (METHOD != 0) {
      const EvalUniforms eu{scalar_copy(u.l0), scalar_copy(u.n0), scalar_copy(u.scal)};
      for_bc([&](int k) {
        const double f = model_value<MODEL, FAST>(eu, st.get(kFc0, k), prep(k));
        double t = (st.get(kFx, k) - f) / eu.scal;
        if (dead(k)) t = 0.0;
        acc[0] = fma(t, t, acc[0]);
      });
    }
    break;
  case RQ_EVAL_MULTI:  // bc: candidates of a projected-gradient search; dif: the trial points of a chain of rejections
    if constexpr (METHOD != 2) {
#ifndef BRDF_EXP_MULTI_GUARDED
      if (u.ncand == kMaxCand) {
        // the usual case, a full set of candidates: no `j < ncand` guard between the candidates of a sample, so that four
        // independent exp chains share a basic block and interleave (guarded, every candidate is a block of its own and its
        // chain runs alone: scripts/micro/exp_ilp.hip measures 76 cycles per exp and SIMD that way, 51 with four side by side)
        if constexpr (METHOD == 0) {  // (the dlevmar_dif kernels have no register to spare: sample by sample, uniforms read from LDS;
                                      // the group-major form below cost them 24 us per 10^6-sample fit)
          for_samples<U>(nk, [&](int k) {
            const double c0 = st.get(kFc0, k), x = st.get(kFx, k);
            const Prep q = prep(k);
            const bool d = dead(k);
#pragma unroll
            for (int j0 = 0; j0 < kMaxCand; j0 += 4) {
              double e[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) e[j] = x - model_value_k<MODEL, FAST>(u, j0 + j, c0, q);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (d) e[j] = 0.0;
                acc[j0 + j] = fma(e[j], e[j], acc[j0 + j]);
              }
            }
          });
          break;
        }
        // ... four candidates at a time over ALL samples, their uniforms (sixteen doubles) in scalar registers for that sweep (see
        // RQ_DIF_TRIAL); every candidate's sum still takes its samples in slot order: configs[3] dlevmar_bc_dif 0.93 -> 0.84 s
#pragma unroll
        for (int j0 = 0; j0 < kMaxCand; j0 += 4) {
          Lin l4[4];
          Nl n4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            l4[j] = scalar_copy(u.lk[j0 + j]);
            n4[j] = scalar_copy(u.nk[j0 + j]);
          }
          for_samples<U>(nk, [&](int k) {
            const double c0 = st.get(kFc0, k), x = st.get(kFx, k);
            const Prep q = prep(k);
            const bool d = dead(k);
            double e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = x - BrdfModel<MODEL>::combine(l4[j], c0, BrdfModel<MODEL>::template shape<FAST>(n4[j], c0, q));  // model_value_k()
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (d) e[j] = 0.0;
              acc[j0 + j] = fma(e[j], e[j], acc[j0 + j]);
            }
          });
        }
        break;
      }
#endif
      for_samples<U>(nk, [&](int k) {
        const double c0 = st.get(kFc0, k), x = st.get(kFx, k);
        const Prep q = prep(k);
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
          if (j < u.ncand) {
            double e = x - model_value_k<MODEL, FAST>(u, j, c0, q);
            if (dead(k)) e = 0.0;
            acc[j] = fma(e, e, acc[j]);
          }
      });
    }
    break;
  case RQ_JAC:
    if constexpr (METHOD != 0) {
      // The kind of row -- forward differences, central differences, the model's analytic Jacobian (dlevmar_bc_der / dlevmar_der)
      // -- is the same for every sample of a pass: chosen HERE, not inside the sample body.  A branch in the body, uniform or not,
      // splits its basic block between the row's two exp chains; in one block the scheduler interleaves them (measured with
      // scripts/micro/exp_ilp.hip: a lone dependent exp chain takes 237 cycles per exp on a wave, two interleaved 110 each).
      auto jac = [&](auto jk) {
        constexpr int JK = decltype(jk)::value;
        JacUniforms ju;  // the fields this kind of row reads, in scalar registers for the sweep (see RQ_DIF_TRIAL)
        ju.l0 = scalar_copy(u.l0);
        ju.n0 = scalar_copy(u.n0);
        ju.analytic = JK == 2;
        ju.central = JK == 1;
        if (JK == 2) {
          ju.an[0] = scalar_copy(u.an[0]);
          ju.an[1] = scalar_copy(u.an[1]);
        } else {
#pragma unroll
          for (int j = 0; j < kM; ++j) {
            ju.lp[j] = scalar_copy(u.lp[j]);
            ju.dinv[j] = scalar_copy(u.dinv[j]);
            if (JK == 1) ju.lm[j] = scalar_copy(u.lm[j]);
          }
          ju.np2 = scalar_copy(u.np2);
          if (JK == 1) ju.nm2 = scalar_copy(u.nm2);
        }
        for_bc([&](int k) {
          double f0 = 0.0, j[kM];
          if (JK == 2)
            model_an_row<MODEL, FAST>(ju, st.get(kFc0, k), prep(k), f0, j);
          else if (JK == 1)
            model_fd_row_t<MODEL, FAST, true>(ju, st.get(kFc0, k), prep(k), true, f0, 0.0, false, j);
          else
            model_fd_row_t<MODEL, FAST, false>(ju, st.get(kFc0, k), prep(k), true, f0, 0.0, false, j);
          double e = st.get(kFx, k) - f0;
          if (dead(k)) e = j[0] = j[1] = j[2] = 0.0;
          acc_normal_eq_fma(j, e, acc, acc + kNL);
          acc[kNL + kM] = fma(e, e, acc[kNL + kM]);
        });
      };
#ifdef BRDF_EXP_OLD_JAC
      for_bc([&](int k) {
        double f0 = 0.0, j[kM];
        if (u.analytic)
          model_an_row<MODEL, FAST>(u, st.get(kFc0, k), prep(k), f0, j);
        else
          model_fd_row<MODEL, FAST>(u, st.get(kFc0, k), prep(k), true, f0, 0.0, false, j);
        double e = st.get(kFx, k) - f0;
        if (dead(k)) e = j[0] = j[1] = j[2] = 0.0;
        acc_normal_eq_fma(j, e, acc, acc + kNL);
        acc[kNL + kM] = fma(e, e, acc[kNL + kM]);
      });
#else
      if (u.analytic)
        jac(std::integral_constant<int, 2>{});
      else if (u.central)
        jac(std::integral_constant<int, 1>{});
      else
        jac(std::integral_constant<int, 0>{});
#endif
    }
    break;
  case RQ_DIF_INIT:
    if constexpr (METHOD == 0) {
      for_samples<U>(nk, [&](int k) {
        const double h = model_value<MODEL, FAST>(u, st.get(kFc0, k), prep(k));
        st.set(kFhx, k, h);
        double e = st.get(kFx, k) - h;
        if (dead(k)) e = 0.0;
        acc[0] = fma(e, e, acc[0]);
      });
    }
    break;
  case RQ_DIF_JAC:
    if constexpr (METHOD == 0) {
      for_samples<U>(nk, [&](int k) {
        const int s = k * kRThreads + tid;
        const double h = st.get(kFhx, k);
        double f0 = 0.0, j[kM];
        model_fd_row<MODEL, FAST>(u, st.get(kFc0, k), prep(k), false, f0, h, true, j);
        double e = st.get(kFx, k) - h;
        if (dead(k)) e = j[0] = j[1] = j[2] = 0.0;
        jl[s] = j[0];
        jl[kRCap + s] = j[1];
        jl[2 * kRCap + s] = j[2];
        acc_normal_eq_fma(j, e, acc, acc + kNL);
      });
    }
    break;
  case RQ_DIF_TRIAL:  // speculative protocol: the Broyden-updated row is formed for the sums only (see the file comment)
    if constexpr (METHOD == 0) {
#ifndef BRDF_EXP_LDS_UNIFORMS
      // The pass's uniforms live in LDS (the control wave builds them there); read through `su`, the sweep keeps them in VECTOR
      // registers, and every use of one in an fp64 instruction then costs a register, a copy or an operand slot.  Copied into
      // SCALAR registers once per sweep (readfirstlane: they are wave-uniform) the same loop runs 2.4x faster in isolation
      // (scripts/micro/trial_sweep.hip: 2.0k against 4.9k ticks per sweep of eight samples at two waves per SIMD).
      struct TrialUniforms {
        Lin lq;
        Nl nq;
        double dp[kM], dp_l2;
      };
      TrialUniforms tu;
      tu.lq = scalar_copy(u.lq);
      tu.nq = scalar_copy(u.nq);
#pragma unroll
      for (int j = 0; j < kM; ++j) tu.dp[j] = scalar_copy(u.dp[j]);
      tu.dp_l2 = scalar_copy(u.dp_l2);
      const double dpp_s[kM] = {scalar_copy(dpp[0]), scalar_copy(dpp[1]), scalar_copy(dpp[2])};
      const TrialUniforms &u = tu;  // (shadows the LDS uniforms for the rest of this case)
      const double *dpp = dpp_s;
#endif
      const double rinv = 1.0 / u.dp_l2;
      auto value_q = [&](int k) { return model_value_q<MODEL, FAST>(u, st.get(kFc0, k), prep(k)); };
      auto body = [&](auto pend_c, int k, const double w) {
        const int s = k * kRThreads + tid;
        const double h = st.get(kFhx, k), x = st.get(kFx, k);
        double jo[kM] = {jl[s], jl[kRCap + s], jl[2 * kRCap + s]};
        if (pend_flag(pend_c)) {  // adopt the previous trial's update: the same operation that formed its jn[] below
          const double tp = st.get(kFtb, k);
#pragma unroll
          for (int j = 0; j < kM; ++j) jo[j] = fma(tp, dpp[j], jo[j]);
          jl[s] = jo[0];
          jl[kRCap + s] = jo[1];
          jl[2 * kRCap + s] = jo[2];
        }
        // broyden_row() (lm_core.c:760-766) with fused multiply-adds (the sweeps are bound by fp64 issue) and the division
        // by ||Dp||^2 done by div_by()
        double t = jo[0] * u.dp[0], jn[kM];
        t = fma(jo[1], u.dp[1], t);
        t = fma(jo[2], u.dp[2], t);
        t = div_by(w - h - t, u.dp_l2, rinv);
#pragma unroll
        for (int j = 0; j < kM; ++j) jn[j] = fma(t, u.dp[j], jo[j]);
        double en = x - w, eo = x - h;
        // (compiling this select into the last slot's copy of the body only -- it is predicated, not branched around, 5 of ~85
        // instructions -- was measured: 43 spilled VGPRs, 605 against 520 us per fit)
        if (dead(k)) en = eo = jn[0] = jn[1] = jn[2] = t = 0.0;
        st.set(kFwrk, k, w);
        st.set(kFtb, k, t);
        // NINE sums, not the machine's thirteen: with J' = J + t Dp^T the products of the updated Jacobian follow from those
        // of the current one, which the machine holds:  J'^T J' = J^T J + a Dp^T + Dp a^T + (t^T t) Dp Dp^T with a = J^T t, and
        // J'^T e = J^T e + Dp (t^T e); only J'^T e' (against the NEW residual) is accumulated directly.  The control wave
        // expands them after the exchange (expand_trial_sums).  Layout: [e'^2, a (3), t^T t, J'^T e' (3), t^T e].  Four
        // accumulators, four reduction slots and four exchange cells fewer per pass: 521 -> 501 us per 10^6-sample fit.
        acc[0] = fma(en, en, acc[0]);
#pragma unroll
        for (int j = 0; j < kM; ++j) acc[1 + j] = fma(jo[j], t, acc[1 + j]);
        acc[1 + kM] = fma(t, t, acc[1 + kM]);
#pragma unroll
        for (int j = 0; j < kM; ++j) acc[2 + kM + j] = fma(jn[j], en, acc[2 + kM + j]);
        acc[2 + 2 * kM] = fma(t, eo, acc[2 + 2 * kM]);
      };
      // (the exp chains of two samples side by side and the Broyden / accumulation halves one after the other was measured too:
      // 455 against 446 us per 10^6-sample fit; whole bodies in pairs spill 28-36 VGPRs)
#ifdef BRDF_TRIAL_HOIST_PEND
      // (whether the previous trial's update is pending is the same for every sample: chosen outside the body, which then is one
      // basic block in which the exp chain, the LDS reads of the row and the Broyden arithmetic can interleave)
      if (pend)
        for_samples<U>(nk, [&](int k) { body(std::true_type{}, k, value_q(k)); });
      else
        for_samples<U>(nk, [&](int k) { body(std::false_type{}, k, value_q(k)); });
#else
      for_samples<U>(nk, [&](int k) { body(pend, k, value_q(k)); });  // (the branch stays in the body)
#endif
    }
    break;
  default: break;  // unknown request: the control wave will not survive it either
  }
}

// the matching reduction (the number of sums depends on the request kind only: wave-uniform; a trial sweep leaves kTrialSums:
// device_common.h)
template <int METHOD>
__device__ __forceinline__ void reduce_pass(int kind, const double *acc, double mx, double *red, double *sums, long long *st_, long long &last_) {
  if constexpr (METHOD == 0) {
    switch (kind) {
    case RQ_DIF_JAC: worker_reduce<SumLayout<kM>::DIF_JAC, false>(acc, mx, red, sums, st_, last_); break;
    case RQ_DIF_TRIAL: worker_reduce<kTrialSums, false>(acc, mx, red, sums, st_, last_); break;
    case RQ_EVAL_MULTI: worker_reduce<kMaxCand, false>(acc, mx, red, sums, st_, last_); break;
    default: worker_reduce<1, false>(acc, mx, red, sums, st_, last_); break;
    }
  } else {
    switch (kind) {
    // (max |e| is read after plain evaluations only -- the overflow guards of lmbc_core.c:748, :915 -- so the Jacobian and
    // multi-candidate passes carry no max slot: one reduction value, one exchange cell and one gather instruction less)
#ifdef BRDF_EXP_KEEP_MAX
    case RQ_JAC: worker_reduce<SumLayout<kM>::JAC, true>(acc, mx, red, sums, st_, last_); break;
    case RQ_EVAL_MULTI: worker_reduce<kMaxCand, true>(acc, mx, red, sums, st_, last_); break;
#else
    case RQ_JAC: worker_reduce<SumLayout<kM>::JAC, false>(acc, mx, red, sums, st_, last_); break;
    case RQ_EVAL_MULTI: worker_reduce<kMaxCand, false>(acc, mx, red, sums, st_, last_); break;
#endif
    default: worker_reduce<1, METHOD == 1>(acc, mx, red, sums, st_, last_); break;  // (dlevmar_der never reads max |e|)
    }
  }
}

// BATCHED = false: one fit spread over the grid (ctx).  BATCHED = true: one workgroup per fit of 1024 < n <= 4096 samples
// (bctx, batch_fit.h) -- the same eight waves / LDS Jacobian, no exchange between workgroups, the speculative
// dlevmar_dif protocol (one pass per LM iteration), the machine started on the device.
template <int MODEL, int METHOD, bool FAST, bool BATCHED>
__global__ __launch_bounds__(kRThreads) void resident_fit_kernel(ResidentCtx ctx, BatchCtx bctx) {
  using Machine = RMachine<METHOD>;
  using Mdl = BrdfModel<MODEL>;
  constexpr int NF = sample_fields<METHOD>();
  // dlevmar_dif: the control wave's 7 x 8 doubles of sample state do not fit next to the LM step's registers: they are
  // parked in LDS.  The other entry points keep 4 x 8 doubles per lane: registers, like every other wave -- except
  // dlevmar_bc_dif spread over the grid, where the exchange code on top of the machine's step spilled 56 VGPRs.
  constexpr bool kControlFromLds = (METHOD == 0) || (METHOD == 1 && !BATCHED);
  static_assert(sizeof(Machine) % 4 == 0, "machine copied as dwords");
  __shared__ Machine sm;
  __shared__ PassUniforms<MODEL> su;
  __shared__ double red[kSlots * kRedCols];
  __shared__ double sums[kSlots];
  __shared__ double dp_prev[kM + 1];  // Dp and ||Dp||^2 of the last trial (dif)
  __shared__ int s_abort, s_bad;
  __shared__ double cst[kControlFromLds ? NF * kRSpt * kWave : 2];  // the control wave's samples
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

  // The machine is started here, by all 64 lanes of the control wave (identical values): a fit costs ONE launch and no
  // upload.  (The first version uploaded a host-started machine + two zeroed control words through a pinned staging block
  // before every launch: ~25 us of a 550 us fit; passing the 1.4 KB machine as a kernel ARGUMENT was worse still -- the
  // argument segment is host memory and 256 workgroups read it over the host link.)
  if (wave == 0) {
    // (the arguments are copied into locals first: handing start() pointers INTO the by-value argument structs makes hipcc
    // spill the whole struct to scratch and serve every later ctx.field access from there -- measured +1.3 us per pass)
    double p0[kM], opts[5], lb[kM], ub[kM], dscl[kM];
    int itmax, has_opts, has_lb, has_ub, has_dscl = 0, want_covar = 0, multi, analytic, chain, spec_jac;
    if constexpr (BATCHED) {
#pragma unroll
      for (int i = 0; i < kM; ++i) {
        p0[i] = bctx.p[(size_t)fit * kM + i];
        lb[i] = bctx.lb[i];
        ub[i] = bctx.ub[i];
        dscl[i] = 1.0;
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) opts[i] = bctx.opts[i];
      itmax = bctx.itmax, has_opts = bctx.has_opts, has_lb = bctx.has_lb, has_ub = bctx.has_ub, multi = bctx.multi, analytic = bctx.analytic;
      chain = bctx.chain, spec_jac = bctx.spec_jac;
    } else {
#pragma unroll
      for (int i = 0; i < kM; ++i) {
        p0[i] = ctx.p0[i];
        lb[i] = ctx.lb[i];
        ub[i] = ctx.ub[i];
        dscl[i] = ctx.dscl[i];
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) opts[i] = ctx.opts[i];
      itmax = ctx.itmax, has_opts = ctx.has_opts, has_lb = ctx.has_lb, has_ub = ctx.has_ub, has_dscl = ctx.has_dscl;
      want_covar = ctx.want_covar, multi = ctx.multi, analytic = ctx.analytic, chain = ctx.chain, spec_jac = ctx.spec_jac;
    }
    const double *po = has_opts ? opts : nullptr;
    if constexpr (METHOD == 0) {
      sm.start(p0, n, itmax, po, want_covar, /*speculative=*/1, chain);
    } else if constexpr (METHOD == 1) {
      sm.start(p0, n, has_lb ? lb : nullptr, has_ub ? ub : nullptr, has_dscl ? dscl : nullptr, itmax, po, want_covar, multi, BATCHED ? 0 : spec_jac);
      sm.c.analytic_jac = analytic;
    } else {
      sm.start(p0, n, itmax, po, want_covar);
    }
    (void)analytic, (void)chain, (void)spec_jac;
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

  // ---- the resident tile: one HBM read ---------------------------------------------------------------------------
  int vb = BATCHED ? 0 : (int)blockIdx.x;  // same XCD-contiguous dealing of tiles as the launch chain
  if (!BATCHED && (G & 7) == 0) vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int tile = (n + G - 1) / G;  // <= kRTile, checked on the host
  const int begin = vb * tile;
  const int end = min(n, begin + tile);
  const double *pc0 = BATCHED ? bctx.angles + (size_t)fit * 3 * n : ctx.c0;
  const double *pc1 = BATCHED ? pc0 + n : ctx.c1;
  const double *pc2 = BATCHED ? pc0 + 2 * (size_t)n : ctx.c2;
  const double *px = BATCHED ? bctx.x + (size_t)fit * n : ctx.x;
  const int nk = (tile + kRThreads - 1) / kRThreads;  // occupied sample slots of a lane (workgroup-uniform)
  const int nfull = max(end - begin, 0) / kRThreads;   // slots in which every lane of this workgroup holds a sample
  RegSamples<METHOD> rs;
  LdsSamples<METHOD> ls{cst + (tid & (kWave - 1))};
  unsigned okm = 0;
  {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kRSpt; ++k) {
      const int i = begin + tid + k * kRThreads;
      const bool ok = i < end;
      okm |= ok ? (1u << k) : 0u;
      const int ii = ok ? i : begin;
      const double r0 = pc0[ii];
      const double r1 = Mdl::uses_c1 ? pc1[ii] : 0.0;
      const double r2 = Mdl::uses_c2 ? pc2[ii] : 0.0;
      const Prep q = Mdl::template prepare<FAST>(r0, r1, r2);
      rs.v[kFc0][k] = r0;
      rs.v[kFx][k] = px[ii];
      rs.v[kFq1][k] = q.q1;
      rs.v[kFq2][k] = q.q2;
      if constexpr (METHOD == 0) rs.v[kFhx][k] = rs.v[kFwrk][k] = rs.v[kFtb][k] = 0.0;
      if (FAST && ok && !Mdl::domain_ok(r0, r1, r2)) bad = true;
    }
    if constexpr (BATCHED) {
      if (FAST && bad) s_bad = 1;  // benign race: every writer stores 1
    } else {
      if (FAST && bad) __hip_atomic_store(&ctx.ctl->domain_bad, ctx.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (kControlFromLds && wave == 0) {  // park the control wave's samples in LDS
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int k = 0; k < kRSpt; ++k) ls.set(f, k, rs.v[f][k]);
    }
  }
  __syncthreads();  // machine, uniforms, s_bad and the parked samples are visible

  int cur_sel_hx = 0, cur_sel_j = 0;
  // what every wave does at the top of a pass (dlevmar_dif): learn what the machine decided about the previous trial
  auto decisions = [&](auto &st, bool &pend) {
    pend = false;
    if constexpr (METHOD == 0) {
      pend = sm.h.req.sel_j != cur_sel_j;  // the Broyden update was adopted: J += tb Dp^T, applied by the next trial sweep
      if (pend && sm.h.req.kind != RQ_DIF_TRIAL) pend = false;  // (a fresh FD Jacobian overwrites J anyway)
      cur_sel_j = sm.h.req.sel_j;
      if (sm.h.req.sel_hx != cur_sel_hx) {  // step accepted: hx <- f(p + Dp)
        for_samples<std::remove_reference<decltype(st)>::type::kUnrolled>(nk, [&](int k) { st.set(kFhx, k, st.get(kFwrk, k)); });
        cur_sel_hx = sm.h.req.sel_hx;
      }
    }
  };

  if (wave == 0) {
    // =========================== control wave: sweep from LDS, exchange, fold, LM step ============================
    // All 64 lanes execute the scalar step with identical values (stream_fit.hip explains why that beats one lane).
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long last_ = clock64();
    long long n_jac = 0;
    // The machine is stepped where it lives, in LDS.  Register variants were built and measured (production builds, 10^6-sample
    // Ward dlevmar_dif, us per pass): Machine::Core (the busy half, ~45 doubles + the counters) kept in this wave's registers
    // for the whole fit 11.29 against 10.53 in LDS -- although the step itself got shorter in the stamped build (6970 against
    // 7950 cycles: every `if (h.k < c.itmax ...)` in LDS is a dependent ds_read -> s_waitcnt -> compare -> branch); a register
    // copy made for every step (170 LDS operations to copy in and out) and all of the machine in registers (105 VGPRs
    // spilled) were slower still.  -DBRDF_CORE_IN_REGS builds the first variant (never for dlevmar_bc_dif: its machine --
    // line search, 8 projected-gradient candidates -- spills 160 VGPRs next to its own step code).
#ifdef BRDF_CORE_IN_REGS
    constexpr bool kCoreInRegs = METHOD != 1;
#else
    constexpr bool kCoreInRegs = false;
#endif
    typename Machine::Core hcore = sm.h;
    if constexpr (kCoreInRegs) Machine::uniform_ints(hcore);
    // dlevmar_dif: what the step reads in every pass but never (constants: options, limits) or only itself (its counters and
    // flags) changes lives in SCALAR registers for the whole fit; the reals stay in LDS with the rest of the machine.  On an
    // LDS-resident machine every `if (h.k < c.itmax ...)` is a dependent ds_read -> s_waitcnt -> compare -> branch.  Same box,
    // 10^6-sample fits: Ward 403.2 -> 401.9 us (constants) -> 398.8 us (+ counters), Blinn-Phong 342.9 -> 338.2 us; all of
    // Machine::Core in (vector) registers was slower (BRDF_CORE_IN_REGS above), the integers alone cost no vector register.
    typename Machine::Cold cold0;
    typename std::conditional<METHOD == 0, typename DifMachine<kM>::CoreInts, int>::type ints_regs{};
    if constexpr (METHOD == 0) {
      cold0.itmax = lm_uniform(sm.c.itmax);
      cold0.n = lm_uniform(sm.c.n);
      cold0.want_covar = lm_uniform(sm.c.want_covar);
      cold0.refresh = lm_uniform(sm.c.refresh);
      cold0.speculative = lm_uniform(sm.c.speculative);
      cold0.multi = lm_uniform(sm.c.multi);
      cold0.o.forward = lm_uniform(sm.c.o.forward);
      cold0.o.tau = scalar_copy(sm.c.o.tau);
      cold0.o.eps1 = scalar_copy(sm.c.o.eps1);
      cold0.o.eps2 = scalar_copy(sm.c.o.eps2);
      cold0.o.eps2sq = scalar_copy(sm.c.o.eps2sq);
      cold0.o.eps3 = scalar_copy(sm.c.o.eps3);
      cold0.o.delta = scalar_copy(sm.c.o.delta);
      ints_regs = sm.h;
      Machine::uniform_ints(ints_regs);
    }
    (void)cold0, (void)ints_regs;
    const long long t_first = (long long)wall_clock64();
    unsigned epoch = 0;
    for (;; ++epoch) {
      const int kind = sm.h.req.kind;
      if (kind == RQ_DONE) break;
      if (kind == RQ_JAC || kind == RQ_DIF_JAC) ++n_jac;
      RTRACE(ctx, epoch, 0, wall_clock64());
      {
        bool pend;
        double acc[kSums];
#pragma unroll
        for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
        double mx = 0.0;
        const double dpp[kM] = {dp_prev[0], dp_prev[1], dp_prev[2]};
        if constexpr (kControlFromLds) {
#ifndef BRDF_EXP_CONTROL_REGS
          decisions(ls, pend);
          sweep_pass<MODEL, METHOD, FAST, !BATCHED>(kind, su, ls, jl, tid, nk, nfull, okm, pend, dpp, acc, mx);
#else
          // (measured and lost, -DBRDF_EXP_CONTROL_REGS: the parked samples brought into registers for the sweep by one burst of ds_reads,
          // the other waves' unrolled register sweep, what the pass changed parked again.  The rolled LDS loop below pays the LDS latency
          // of a sample's seven fields -- 7.0k ticks per sweep against the register waves' ~5k -- but the burst variant spills 46 VGPRs
          // inside the sweep: 480 against 425 us per 10^6-sample fit.)
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int k = 0; k < kRSpt; ++k) rs.v[f][k] = ls.get(f, k);
          decisions(rs, pend);
          sweep_pass<MODEL, METHOD, FAST, !BATCHED>(kind, su, rs, jl, tid, nk, nfull, okm, pend, dpp, acc, mx);
          if constexpr (METHOD == 0) {
#pragma unroll
            for (int k = 0; k < kRSpt; ++k) {
              ls.set(kFhx, k, rs.v[kFhx][k]);
              ls.set(kFwrk, k, rs.v[kFwrk][k]);
              ls.set(kFtb, k, rs.v[kFtb][k]);
            }
          }
#endif
        } else {
          decisions(rs, pend);
          sweep_pass<MODEL, METHOD, FAST, !BATCHED>(kind, su, rs, jl, tid, nk, nfull, okm, pend, dpp, acc, mx);
        }
        RSTAMP(5);  // the control wave's own sweep
        RTRACE(ctx, epoch, 1, wall_clock64());
        reduce_pass<METHOD>(kind, acc, mx, red, sums, st_, last_);  // X1, X2: sums[] hold this workgroup's partial sums
      }
      RSTAMP(1);  // reduction + waiting for the slowest wave
      RTRACE(ctx, epoch, 2, wall_clock64());
      bool alive = true;
      if constexpr (BATCHED && FAST) {  // every wave looked at its cosines while loading the tile
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
       if (gridDim.x > 1) {      // ... and so is a single fit of <= 4096 samples: no exchange, no visibility hops
        if constexpr (METHOD == 0) {
          switch (kind) {
          case RQ_DIF_JAC: alive = control_exchange<SumLayout<kM>::DIF_JAC, false>(ctx, epoch, sums, &s_abort, st_, last_); break;
          case RQ_DIF_TRIAL: alive = control_exchange<kTrialSums, false>(ctx, epoch, sums, &s_abort, st_, last_); break;
          case RQ_EVAL_MULTI: alive = control_exchange<kMaxCand, false>(ctx, epoch, sums, &s_abort, st_, last_); break;
          default: alive = control_exchange<1, false>(ctx, epoch, sums, &s_abort, st_, last_); break;
          }
        } else {
          switch (kind) {
          case RQ_JAC: alive = control_exchange<SumLayout<kM>::JAC, false>(ctx, epoch, sums, &s_abort, st_, last_); break;
          case RQ_EVAL_MULTI: alive = control_exchange<kMaxCand, false>(ctx, epoch, sums, &s_abort, st_, last_); break;
          default: alive = control_exchange<1, METHOD == 1>(ctx, epoch, sums, &s_abort, st_, last_); break;
          }
        }
       }
      }
      if (!alive) {  // give up: the host sees no `done`, reads ctl->abort and falls back
        __syncthreads();  // B (the other waves read s_abort behind it)
        return;
      }
      if (kind == RQ_DIF_TRIAL) {
        if constexpr (METHOD == 0) {
          if constexpr (kCoreInRegs)
            expand_trial_sums(hcore, sm.h.cool, su.dp, sums);
          else
            expand_trial_sums(static_cast<const typename Machine::Core &>(sm.h), sm.h.cool, su.dp, sums);
        }
#pragma unroll
        for (int j = 0; j < kM; ++j) dp_prev[j] = su.dp[j];
        dp_prev[kM] = su.dp_l2;
      }
      // The LM step.  kCoreInRegs: the machine's busy half (Machine::Core) lives in this wave's REGISTERS for the whole fit
      // (hcore, loaded before the loop); Cool and the request stay in LDS, where the other waves read the request.
      if constexpr (kCoreInRegs) {
        if constexpr (METHOD == 0)
          Machine::template run<true, true>(sm.c, hcore, sm.h.cool, sm.h.req, sums, sums[kSums]);
        else
          Machine::template run<true>(sm.c, hcore, sm.h.req, sums, sums[kSums]);
      } else {
        if constexpr (METHOD == 0) {  // (+ chains of rejections, several trial points to a sweep)
          typename Machine::Cold cc;  // (results are written by the finishing step only and stored right behind it: nothing carried)
          cc.itmax = cold0.itmax, cc.n = cold0.n, cc.want_covar = cold0.want_covar, cc.refresh = cold0.refresh;
          cc.speculative = cold0.speculative, cc.multi = cold0.multi, cc.o = cold0.o;
          for (int i = 0; i < kInfoSz; ++i) cc.info[i] = 0.0;
          for (int i = 0; i < kM * kM; ++i) cc.covar[i] = 0.0;
          cc.ret = kLmError;
          Machine::template run<true, true>(cc, ints_regs, static_cast<typename Machine::CoreReals &>(sm.h), sm.h.cool, sm.h.req, sums, sums[kSums]);
          Machine::uniform_ints(ints_regs);  // (assignments under formally divergent branches lose their uniformity: re-assert it)
          if (sm.h.req.kind == RQ_DONE) {
            for (int i = 0; i < kInfoSz; ++i) sm.c.info[i] = cc.info[i];
            for (int i = 0; i < kM * kM; ++i) sm.c.covar[i] = cc.covar[i];
            sm.c.ret = cc.ret;
          }
        } else
        if constexpr (METHOD == 1)
          sm.template step<true, true, false, !BATCHED>(sums, sums[kSums]);  // (+ candidates evaluated by Jacobian passes: single fits only)
        else
          sm.template step<true>(sums, sums[kSums]);
      }
      RSTAMP(7);  // the step alone
      if (sm.h.req.kind != RQ_DONE) {
        // (lane-parallel for the single box-constrained fit only: the dif kernels have no register for it -- 5 -> 10 spilled VGPRs --
        // and hardly a request that gains; the batched bc kernel spills 8 with it)
        if constexpr (METHOD == 1 && !BATCHED)
          build_uniforms_wave(su, sm.h.req, /*need_base=*/false, sm.c.analytic_jac != 0);
        else if constexpr (METHOD == 1)
          su.build(sm.h.req, /*need_base=*/false, sm.c.analytic_jac != 0);
        else
          su.build(sm.h.req, /*need_base=*/false, METHOD == 2);
      }
      RTRACE(ctx, epoch, 5, wall_clock64());
      __syncthreads();  // B: the next request and its uniforms are in LDS
      RSTAMP(4);
    }
    if constexpr (kCoreInRegs) static_cast<typename Machine::Core &>(sm.h) = hcore;
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
      mb->domain_bad = __hip_atomic_load(&ctx.ctl->domain_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ctx.launch_id ? 1 : 0;
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

  // ============================= waves 1..7: register-resident samples ===========================================
  long long wst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wlast_ = 0;  // (stamps are the control wave's; these are never read)
#ifdef BRDF_TRACE_WORKERS
  unsigned wepoch = 0;
#endif
  for (;;) {
    const int kind = sm.h.req.kind;
    if (kind == RQ_DONE) break;
    bool pend;
    decisions(rs, pend);
    double acc[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
    double mx = 0.0;
    const double dpp[kM] = {dp_prev[0], dp_prev[1], dp_prev[2]};
    sweep_pass<MODEL, METHOD, FAST, !BATCHED>(kind, su, rs, jl, tid, nk, nfull, okm, pend, dpp, acc, mx);
#ifdef BRDF_TRACE_WORKERS  // (diagnostic: when do the register-resident waves finish their sweeps? slots 6, 7 = waves 4, 7)
    if constexpr (!BATCHED) {
      if (wave == 4) RTRACE(ctx, wepoch, 6, wall_clock64());
      if (wave == 7) RTRACE(ctx, wepoch, 7, wall_clock64());
      ++wepoch;
    }
#endif
    reduce_pass<METHOD>(kind, acc, mx, red, sums, wst_, wlast_);
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

struct RWorkspace {
  int device = -1, cus = 0;
  char *d_block = nullptr;  // ctl | machine | rows[2][kRowWords][kRowStride]
  Mailbox *h_mbox = nullptr, *d_mbox = nullptr;
  static constexpr size_t kMachineBytes = 4096;
  static constexpr size_t off_machine = sizeof(ResidentCtl);
  static constexpr size_t off_rows = off_machine + kMachineBytes;
  static constexpr size_t rows_bytes = sizeof(u64) * (kRowsGranules + kGroupsGranules);  // rows + group rows
  static constexpr size_t trace_bytes = sizeof(long long) * 8 * kRowStride;
  long long h_trace[8 * (kRowStride + 1)] = {0};  // + one row: the sections of the LM step (LM_STAMP)
  unsigned tag_base = 0;
  FitStats stats{};
  LaunchTimer timer;
  // After a launch that could not run co-resident (GPU shared with other kernels / ranks: every workgroup burns its
  // spin budget before the launch drains) the resident path steps aside for the next `skip` fits, doubling up to
  // 1024 while it keeps failing, instead of paying that budget on every fit.
  int backoff = 0, skip = 0;

  // the blocks belong to `device`: drain and free them THERE, whatever device is current now
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
  ~RWorkspace() { release(); }
  int ensure(int dev) {
    if (device == dev && d_block) return 0;
    release();
    device = dev;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
    HIP_OK(hipMalloc(&d_block, off_rows + rows_bytes + trace_bytes));
    HIP_OK(hipMemset(d_block, 0, off_rows + rows_bytes + trace_bytes));
    tag_base = 0;
    HIP_OK(hipHostMalloc(&h_mbox, sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void **)&d_mbox, h_mbox, 0));
    return 0;
  }
};
template <int MODEL, int METHOD, bool FAST>
int resident_attempt(const StreamFitArgs &a, RWorkspace &ws, bool *retry_exact, bool *unavailable) {
  using Machine = RMachine<METHOD>;
  static_assert(sizeof(Machine) <= 4096, "resident workspace layout");
  *retry_exact = *unavailable = false;
  const int G = (int)std::min<long long>(ws.cus, std::max<long long>(1, ((long long)a.n + 1023) / 1024));
  Machine m;  // started here for the entry point's argument checks and warnings only: the kernel starts its own
  memset(&m, 0, sizeof m);
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
  if (ws.tag_base > 0xF0000000u) {  // tag space nearly used up: start over from zeroed rows (and control words)
    HIP_OK(hipMemsetAsync(ws.d_block, 0, RWorkspace::off_rows + RWorkspace::rows_bytes, a.stream));
    ws.tag_base = 0;
  }

  ResidentCtx c;
  c.c0 = a.d_angles;
  c.c1 = a.d_angles + a.n;
  c.c2 = a.d_angles + 2 * (size_t)a.n;
  c.x = a.d_x;
  c.ctl = reinterpret_cast<ResidentCtl *>(ws.d_block);
  c.rows = reinterpret_cast<u64 *>(ws.d_block + RWorkspace::off_rows);
  c.groups = c.rows + kRowsGranules;
  c.launch_id = ws.tag_base + 1u;  // (tag_base grows by passes + 2 with every launch)
  for (int i = 0; i < kM; ++i) {
    c.p0[i] = a.p[i];
    c.lb[i] = a.lb ? a.lb[i] : 0.0;
    c.ub[i] = a.ub ? a.ub[i] : 0.0;
    c.dscl[i] = a.dscl ? a.dscl[i] : 1.0;
  }
  for (int i = 0; i < 5; ++i) c.opts[i] = a.opts ? a.opts[i] : 0.0;
  c.itmax = a.itmax;
  c.has_opts = a.opts != nullptr;
  c.has_lb = METHOD == 1 && a.lb != nullptr;
  c.has_ub = METHOD == 1 && a.ub != nullptr;
  c.has_dscl = METHOD == 1 && a.dscl != nullptr;
  c.want_covar = a.covar != nullptr;
  c.multi = pg_candidates();
  c.chain = dif_chain_candidates();
  c.spec_jac = bc_spec_jac_enabled() ? 1 : 0;
  c.analytic = a.analytic ? 1 : 0;
  c.mbox = ws.d_mbox;
  c.n = a.n;
  c.tag_base = ws.tag_base;
  c.spin_ticks = kSpinBudgetTicks;
  c.sabotage_epoch = -1;
  c.replicas = kReplicas;
  c.trace = nullptr;
  c.trace_epoch = -1;
#ifdef BRDF_STAMPS
  c.trace = reinterpret_cast<long long *>(ws.d_block + RWorkspace::off_rows + RWorkspace::rows_bytes);
  c.trace_epoch = 20;
  if (const char *e = getenv("BRDF_HIP_RESIDENT_TRACE_EPOCH")) c.trace_epoch = atoi(e);
#endif
  if (const char *e = getenv("BRDF_HIP_RESIDENT_REPLICAS")) c.replicas = std::min(kReplicas, std::max(1, atoi(e)));
  if (const char *e = getenv("BRDF_HIP_RESIDENT_SPIN_MS")) c.spin_ticks = std::max(1LL, atoll(e)) * 100000LL;
  if (const char *e = getenv("BRDF_HIP_RESIDENT_SABOTAGE")) c.sabotage_epoch = atoi(e);  // tests only: forces the fallback

  {  // one workgroup per CU must be able to live there at all (registers, LDS): checked once per kernel
    static int per_cu = -1;
    if (per_cu < 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, resident_fit_kernel<MODEL, METHOD, FAST, false>, kRThreads, 0) != hipSuccess)
      per_cu = 0;
    if (per_cu < 1) {
      *unavailable = true;
      return 0;
    }
  }
  ws.timer.before(a.stream);
  hipLaunchKernelGGL((resident_fit_kernel<MODEL, METHOD, FAST, false>), dim3(G), dim3(kRThreads), 0, a.stream, c, BatchCtx{});
  HIP_OK(hipGetLastError());
  ws.timer.after(a.stream);
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
    (void)hipMemsetAsync(ws.d_block, 0, RWorkspace::off_rows + RWorkspace::rows_bytes, a.stream);
    ws.tag_base = 0;
    *unavailable = true;
    return 0;
  }
  ws.tag_base += (unsigned)mb.passes + 2u;
#ifdef BRDF_STAMPS
  (void)hipMemcpy(ws.h_trace, c.trace, RWorkspace::trace_bytes, hipMemcpyDeviceToHost);
  (void)hipMemcpyFromSymbol(ws.h_trace + 8 * kRowStride, HIP_SYMBOL(g_rlm_stamps), sizeof(long long) * 8);
  {
    long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rlm_stamps), zero, sizeof zero);
  }
#endif
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
  ws.stats.kernel_us = ws.timer.elapsed_us();
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

// one translation unit per (MODEL, METHOD): resident_inst.hip defines these two entry points for its pair
#define BRDF_RESIDENT_INSTANCE(MODEL_, METHOD_, NAME_)                                                                        \
  int resident_run_##NAME_(const StreamFitArgs &a, RWorkspace &ws, bool *unavailable) { return resident_run_mm<MODEL_, METHOD_>(a, ws, unavailable); } \
  int resident_batch_##NAME_(bool fast, const BatchCtx &c, hipStream_t stream) { return resident_batch_mm<MODEL_, METHOD_>(fast, c, stream); }

}  // namespace brdf
