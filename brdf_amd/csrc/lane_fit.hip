// lane_fit.hip -- batched regime for the application's own fit size: n <= 16 samples (16 lights per surfel,
// brdfdata.h:58), dlevmar_bc_dif -- the call CBRDFdata::SolveEquation makes once per pixel and colour channel
// (brdfdata.cpp:1077-1136, loop at :1195-1220).
//
// ONE LANE PER FIT.  At this size a fit has no data parallelism worth a wave: 16 samples x one transcendental is ~1000
// instructions per pass, while the scalar LM step between two passes is ~2000 dependent instructions.  With one wave
// (or one DPP row) per fit the step runs on one lane while 63 (15) idle, and it was 80-99 % of the kernel.  Here every
// lane owns a whole fit:
//
//   * its LM state machine (BcMachine<3>, lm_machine.h) is a per-lane object -- 64 machines step side by side in
//     one wave, each in its own phase (the phase switch diverges; every phase body is paid once per round for up to
//     64 fits instead of once per fit);
//   * its <= 16 prepared samples live in LDS, structure-of-arrays over lanes: smp[sample][plane][lane], so a lane's
//     read of (sample, plane) is one conflict-free ds_read_b64 for the wave;
//   * the sweep over its own samples is serial, in the REFERENCE'S summation order: the descending "small problem"
//     loop for J^T J / J^T e (lmbc_core.c:595-615, n*m < 1024) and the four-accumulator walk of dlevmar_L2nrmxmy
//     (misc_core.c:721-807).  No tree, no cross-lane traffic: with the exact model path the only difference left to
//     the CPU reference is pow() itself (ocml vs glibc, last-bit);
//   * lanes pull fits from a global queue (one wave-aggregated atomicAdd per refill), so a lane whose fit ends after
//     10 iterations starts the next one while its neighbours are still in their line searches.
//
// Fits with a cosine <= 0 cannot use the cached-log path: the FAST kernel marks them (flags[]) and the exact twin,
// launched behind it, fits exactly those with the reference's pow() -- the same protocol as batch_fit.hip.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#ifdef BRDF_LANE_STAMPS
#include <hip/hip_runtime.h>
// per-phase cycles of the machines' run(): a wave walks serially over the phase blocks its lanes are in; the first active
// lane of every block entered closes the previous interval (one wave per workgroup here, so the state lives in LDS)
__shared__ long long lp_acc[32];
__shared__ long long lp_t;
__shared__ int lp_prev;
__device__ __forceinline__ bool lane_phase_enter(int x) {
  const unsigned long long act = __ballot(1);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) {
    const long long now = clock64();
    lp_acc[lp_prev & 31] += now - lp_t;
    lp_prev = x;
    lp_t = now;
  }
  return true;
}
#define LM_PHASE_ENTER(X) lane_phase_enter((int)(X))
#endif
#include "batch_fit.h"
#include "stream_fit.h"

namespace brdf {

#ifdef BRDF_LANE_STAMPS
__device__ long long g_lane_phase[32];  // diagnostic build (make variant EXTRA=-DBRDF_LANE_STAMPS): where a wave's cycles go, summed over all waves
__device__ long long g_lane_stamps[16];
#define LSTAMP(i) do { const long long now_ = clock64(); lst_[i] += now_ - llast_; llast_ = now_; } while (0)
#define LCOUNT(i, v) do { lst_[i] += (v); } while (0)
#else
#define LSTAMP(i) do {} while (0)
#define LCOUNT(i, v) do {} while (0)
#endif

// W = waves per SIMD the register allocator plans for (512 / 256 / 128 registers per lane at 1 / 2 / 4); LDS allows
// 5-6 waves per CU at n = 16.  BRDF_HIP_LANE_WAVES picks the variant (measurements: DESIGN.md section 6).
template <int MODEL, bool FAST, int W>
__global__ __launch_bounds__(kWave, W) void lane_fit_kernel(BatchCtx ctx, int *queue) {
  using Mdl = BrdfModel<MODEL>;
  constexpr int NP = 3 + (Mdl::prep_planes == 2 ? 1 : 0);  // c0, q1, [q2], x
  extern __shared__ double smp[];                          // [n][NP][64]
  const int lane = threadIdx.x;
  const int n = ctx.n;
  const int S = ctx.S;
  // (locals, not pointers into the by-value ctx: taking its members' addresses parks the whole struct in scratch)
  const double ov[5] = {ctx.opts[0], ctx.opts[1], ctx.opts[2], ctx.opts[3], ctx.opts[4]};
  const double lbv[kM] = {ctx.lb[0], ctx.lb[1], ctx.lb[2]}, ubv[kM] = {ctx.ub[0], ctx.ub[1], ctx.ub[2]};
  const double *opts = ctx.has_opts ? ov : nullptr;
  const double *lb = ctx.has_lb ? lbv : nullptr;
  const double *ub = ctx.has_ub ? ubv : nullptr;

  BcMachine<kM> m;
  // the half of the machine every fit of the batch shares is configured once, here, in wave-uniform control flow: options,
  // box and limits then live in scalar registers instead of 64 identical per-lane copies
  m.configure(n, lb, ub, nullptr, ctx.itmax, opts, 0, 1);
  m.c.analytic_jac = ctx.analytic;
  m.h.req.kind = RQ_DONE;
  int fit = -1;
  bool more = true;  // wave-uniform: the queue may still hold fits
  int since_heavy = 0;
#ifdef BRDF_LANE_STAMPS
  long long lst_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long llast_ = clock64();
  const long long lbegin_ = llast_;
  if (lane < 32) lp_acc[lane] = 0;
  if (lane == 0) { lp_prev = 0; lp_t = llast_; }
  __syncthreads();
#endif

  // Rounds.  In a round every lane with an evaluation request sweeps its samples and steps its machine.  The EXPENSIVE
  // things -- a Jacobian sweep with the 3x3 solve behind it, the line-search prologue (pow, square roots), the epilogue of
  // a fit, fetching and preparing the next fit -- are needed by a lane about once per LM iteration, i.e. in ~10 % of its
  // rounds, but with 64 lanes somebody needs each of them in EVERY round, and a wave pays for a phase body whenever one
  // lane runs it.  So they are gated (BcMachine::run<GATED>): lanes that reach one wait, and the wave runs a "heavy" round
  // for all of them together once a quorum waits (or nobody has anything else to do, or a lane has waited long enough).
  // Light rounds then cost an evaluation sweep plus the cheap phases only.  Scheduling cannot change a result.
  for (;;) {
    // a fit's results are written by the step that ends it and stored right behind that step (below): nothing of them
    // is carried from one round to the next, and saying so keeps 2 x 20 registers per lane free between the two
    for (int i = 0; i < kInfoSz; ++i) m.c.info[i] = 0.0;
    for (int i = 0; i < kM * kM; ++i) m.c.covar[i] = 0.0;
    m.c.ret = kLmError;
    const int kind0 = (fit >= 0) ? m.h.req.kind : (int)RQ_DONE;
    const bool wants_heavy = (fit < 0 && more) || kind0 == RQ_JAC || kind0 == RQ_YIELD;
    const bool light_work = fit >= 0 && kind0 != RQ_JAC && kind0 != RQ_YIELD;
    const int nh = __popcll(__ballot(wants_heavy));
    const int nl = __popcll(__ballot(light_work));
    if (nh == 0 && nl == 0) break;
    const bool heavy = nl == 0 || nh >= ctx.lane_quorum || since_heavy >= ctx.lane_maxwait;
    since_heavy = heavy ? 0 : since_heavy + 1;
    LSTAMP(0);  // round bookkeeping
    LCOUNT(heavy ? 8 : 7, 1);
    LCOUNT(heavy ? 10 : 9, nl);
    LCOUNT(11, nh);

    // ---- refill (heavy rounds): every idle lane takes the next fit of the queue ------------------------------
    bool want = heavy && fit < 0;
    while (more && __any(want)) {
      const unsigned long long mask = __ballot(want);
      const int cnt = __popcll(mask);
      const int first = __ffsll((long long)mask) - 1;
      int base = 0;
      if (lane == first) base = atomicAdd(queue, cnt);
      base = __shfl(base, first);
      if (base + cnt >= S) more = false;
      if (want) {
        const int f = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (f < S && (FAST || ctx.flags[f] == kNeedsExact)) {
          const double *a = ctx.angles + (size_t)f * 3 * n;
          const double *xs = ctx.x + (size_t)f * n;
          bool bad = false;
          for (int i = 0; i < n; ++i) {
            const double c0 = a[i];
            const double r1 = Mdl::uses_c1 ? a[n + i] : 0.0;
            const double r2 = Mdl::uses_c2 ? a[2 * n + i] : 0.0;
            const Prep q = Mdl::template prepare<FAST>(c0, r1, r2);
            if (FAST && !Mdl::domain_ok(c0, r1, r2)) bad = true;
            double *d = smp + (size_t)i * NP * kWave + lane;
            d[0] = c0;
            d[kWave] = q.q1;
            if (NP == 4) d[2 * kWave] = q.q2;
            d[(NP - 1) * kWave] = xs[i];
          }
          if (FAST) ctx.flags[f] = bad ? kNeedsExact : 0;
          if (!(FAST && bad)) {
            const double p0[kM] = {ctx.p[(size_t)f * kM], ctx.p[(size_t)f * kM + 1], ctx.p[(size_t)f * kM + 2]};
            m.begin(p0);
            if (m.h.req.kind == RQ_DONE) {  // refused by start() (n < m, inconsistent box): lmbc_core.c:440-454
              if (ctx.ret) ctx.ret[f] = kLmError;
              if (ctx.info)
                for (int i = 0; i < kInfoSz; ++i) ctx.info[(size_t)f * kInfoSz + i] = 0.0;
            } else {
              fit = f;
              want = false;
            }
          }
        }
        if (f >= S) want = false;  // nothing left for this lane
      }
    }

    LSTAMP(1);  // refill
    if (fit >= 0) {
      // ---- one pass over this lane's samples, sums in the reference's order -----------------------------------
      const Request<kM> &r = m.h.req;
      const int kind = r.kind;
      double s[kSlots];
#pragma unroll
      for (int k = 0; k < kSlots; ++k) s[k] = 0.0;
      double mx = 0.0;
      const double *sp = smp + lane;
      bool do_step = false;
      if (kind == RQ_JAC) {
        if (heavy) {  // lmbc_core.c:595-615: for l = n-1..0 { jtj[i][j] += row[j]*row[i]; jte[i] += row[i]*e[l] }
          PassUniforms<MODEL> u;
          u.build(r, true, ctx.analytic != 0);
          auto row = [&](int l, double &e, double *j) {
            const double *d = sp + (size_t)l * NP * kWave;
            const Prep q{d[kWave], NP == 4 ? d[2 * kWave] : 0.0};
            double f0 = 0.0;
            if (ctx.analytic)  // dlevmar_bc_der: the model's analytic Jacobian (wave-uniform branch)
              model_an_row<MODEL, FAST>(u, d[0], q, f0, j);
            else
              model_fd_row<MODEL, FAST>(u, d[0], q, true, f0, 0.0, false, j);
            e = d[(NP - 1) * kWave] - f0;
          };
          // two rows per trip (one wave per SIMD: nothing else hides a row's dependent exp chains), accumulated in the
          // reference's order all the same
          int l = n;
          for (; W == 1 && l >= 2; l -= 2) {
            double ea, eb, ja[kM], jb[kM];
            row(l - 1, ea, ja);
            row(l - 2, eb, jb);
            acc_normal_eq(ja, ea, s, s + kNL);
            acc_normal_eq(jb, eb, s, s + kNL);
          }
          for (; l >= 1; --l) {  // (two waves per SIMD hide each other's chains: one row per trip, fewer registers)
            double ea, ja[kM];
            row(l - 1, ea, ja);
            acc_normal_eq(ja, ea, s, s + kNL);
          }
          do_step = true;
        }
      } else if (kind == RQ_YIELD) {
        do_step = heavy;
      } else {
        PassUniforms<MODEL> u;  // an evaluation reads l0, n0 (and scal) only
        u.l0 = Mdl::lin(r.p);
        u.n0 = Mdl::nl(r.p);
        u.scal = r.scal;
        if (kind == RQ_SCALED) {  // lmbc_core.c:163-166, descending
          for (int l = n; l-- > 0;) {
            const double *d = sp + (size_t)l * NP * kWave;
            const Prep q{d[kWave], NP == 4 ? d[2 * kWave] : 0.0};
            const double t = (d[(NP - 1) * kWave] - model_value<MODEL, FAST>(u, d[0], q)) / u.scal;
            s[0] += t * t;
          }
        } else {  // RQ_EVAL: misc_core.c:721-807 -- blocks of 8 from the top down, accumulator (top - j) & 3; then the tail upwards
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
          const int body = (n >> 3) << 3;
          for (int jb = body - 4; jb >= 0; jb -= 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const double *d = sp + (size_t)(jb + 3 - k) * NP * kWave;
              const Prep q{d[kWave], NP == 4 ? d[2 * kWave] : 0.0};
              const double e = d[(NP - 1) * kWave] - model_value<MODEL, FAST>(u, d[0], q);
              const double e2 = e * e;
              if (k == 0) a0 += e2;
              if (k == 1) a1 += e2;
              if (k == 2) a2 += e2;
              if (k == 3) a3 += e2;
              mx = fmax(mx, fabs(e));
            }
          }
          for (int t = body; t < n; ++t) {
            const double *d = sp + (size_t)t * NP * kWave;
            const Prep q{d[kWave], NP == 4 ? d[2 * kWave] : 0.0};
            const double e = d[(NP - 1) * kWave] - model_value<MODEL, FAST>(u, d[0], q);
            const double e2 = e * e;
            const int k = (7 - (n - t)) & 3;
            a0 += (k == 0) ? e2 : 0.0;
            a1 += (k == 1) ? e2 : 0.0;
            a2 += (k == 2) ? e2 : 0.0;
            a3 += (k == 3) ? e2 : 0.0;
            mx = fmax(mx, fabs(e));
          }
          s[0] = a0 + a1 + a2 + a3;
        }
        do_step = true;
      }
      LSTAMP(heavy ? 3 : 2);  // sweeps of a heavy / light round
#ifdef BRDF_LANE_STAMPS
      lane_phase_enter(heavy ? 30 : 31);  // (the interval up to here -- sweeps, bookkeeping -- is not a phase's)
#endif
      if (do_step) {
        m.template step<false, false, true>(s, mx, heavy);
        if (m.h.req.kind == RQ_DONE) {
          double *po = ctx.p + (size_t)fit * kM;
          for (int i = 0; i < kM; ++i) po[i] = m.h.p[i];
          if (ctx.info)
            for (int i = 0; i < kInfoSz; ++i) ctx.info[(size_t)fit * kInfoSz + i] = m.c.info[i];
          if (ctx.ret) ctx.ret[fit] = m.c.ret;
          fit = -1;
        }
      }
      LSTAMP(heavy ? 5 : 4);  // steps (+ result write-out) of a heavy / light round
    }
#ifdef BRDF_LANE_STAMPS
    lane_phase_enter(0);
#endif
  }
#ifdef BRDF_LANE_STAMPS
  lst_[12] = clock64() - lbegin_;
  if (lane == 0)
    for (int i = 0; i < 13; ++i) atomicAdd((unsigned long long *)&g_lane_stamps[i], (unsigned long long)lst_[i]);
  __syncthreads();
  if (lane < 32) atomicAdd((unsigned long long *)&g_lane_phase[lane], (unsigned long long)lp_acc[lane]);
#endif
}

#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return kLmError;                                                                \
    }                                                                                 \
  } while (0)

namespace {
using LaneFn = void (*)(BatchCtx, int *);
template <int W>
LaneFn lane_kernel_w(int model, bool fast) {
  static const LaneFn table[2][MODEL_COUNT] = {
      {lane_fit_kernel<0, false, W>, lane_fit_kernel<1, false, W>, nullptr},  // Ward's prepared path has no domain restriction
      {lane_fit_kernel<0, true, W>, lane_fit_kernel<1, true, W>, lane_fit_kernel<2, true, W>},
  };
  return table[fast ? 1 : 0][model];
}
int lane_waves_per_simd() {
  const char *e = getenv("BRDF_HIP_LANE_WAVES");
  const int w = e ? atoi(e) : 1;  // measured (2^20 Blinn-Phong fits): 1.43e7 / 1.12e7 / 5.5e6 fits/s at 1 / 2 / 4 (spills beat occupancy)
  return (w == 2 || w == 4) ? w : 1;
}
LaneFn lane_kernel(int model, bool fast, int w) {
  return w == 1 ? lane_kernel_w<1>(model, fast) : (w == 4 ? lane_kernel_w<4>(model, fast) : lane_kernel_w<2>(model, fast));
}
}  // namespace

// dlevmar_bc_dif, n <= kLaneMaxN.  c.flags: S ints; queue: 2 ints, zeroed by the caller on `stream`.
int lane_fit_enqueue(int model, bool fast, const BatchCtx &c, int *queue, hipStream_t stream) {
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  int cus = 0;
  HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int np = (model == MODEL_WARD) ? 4 : 3;
  const size_t lds = sizeof(double) * (size_t)c.n * np * kWave;
  const int w = lane_waves_per_simd();
  long long per_cu = (160 * 1024) / (long long)std::max<size_t>(lds, 1);
  if (per_cu > 4 * w) per_cu = 4 * w;
  if (per_cu < 1) per_cu = 1;
  BatchCtx cc = c;
  cc.lane_quorum = 24;  // measured (2^20 fits, one wave per SIMD): quorum 1 (no gating) 1.18e7, 8 1.38e7, 16 1.44e7, 24 1.47e7, 32 1.45e7, 40 1.37e7 fits/s
  cc.lane_maxwait = 6;
  if (const char *e = getenv("BRDF_HIP_LANE_QUORUM")) cc.lane_quorum = std::max(1, atoi(e));
  if (const char *e = getenv("BRDF_HIP_LANE_MAXWAIT")) cc.lane_maxwait = std::max(0, atoi(e));
  long long waves = (long long)cus * per_cu;
  const long long need = ((long long)c.S + kWave - 1) / kWave;
  if (waves > need) waves = need;
  if (fast) {
    hipLaunchKernelGGL(lane_kernel(model, true, w), dim3((unsigned)waves), dim3(kWave), lds, stream, cc, queue);
    HIP_OK(hipGetLastError());
    if (model != MODEL_WARD) {  // (a launch over an empty set costs one queue sweep: every lane's first fetch finds no marked fit)
      hipLaunchKernelGGL(lane_kernel(model, false, w), dim3((unsigned)waves), dim3(kWave), lds, stream, cc, queue + 1);
      HIP_OK(hipGetLastError());
    }
  } else {
    HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c.flags), kNeedsExact, (size_t)c.S, stream));
    hipLaunchKernelGGL(lane_kernel(model, false, w), dim3((unsigned)waves), dim3(kWave), lds, stream, cc, queue);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

#ifdef BRDF_LANE_STAMPS
extern "C" int brdf_hip_lane_stamps(long long *out) {  // out[0..15] sections, out[16..47] phases; reads and clears (diagnostic builds only)
  long long zero[32] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_stamps), 16 * sizeof(long long)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_lane_phase), 32 * sizeof(long long)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_phase), zero, 32 * sizeof(long long)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stamps), zero, 16 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace brdf
