// batch_fit.hip -- batched regime: S independent fits, one workgroup (or one wavefront) per fit.
//
// Replaces the serial pixel x colour-channel loop of CBRDFdata::CalcBRDFEquation
// (brdfdata.cpp:1195-1220): each of its iterations packs 16 samples and calls dlevmar_bc_dif
// (brdfdata.cpp:1077-1136).  Here the fits are independent workgroups; a fit's samples are read from
// HBM exactly once (24-32 B per sample) into REGISTERS, its per-sample invariants (brdf_models.h: Prep)
// are derived once, and the entire LM iteration -- model evaluation, residuals, FD Jacobian, J^T J /
// J^T e, Broyden update, 3x3 solve, line search -- runs out of registers + ~3 KB of LDS.  A "pass" is a
// sweep over the lane's SPT cached samples followed by a DPP/LDS workgroup reduction; the scalar LM
// state machine (lm_machine.h) lives in LDS and is stepped by lane 0.  Fits finish after different
// numbers of passes; the hardware workgroup scheduler backfills, which is the load balancing.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/brdf_levmar.h"
#include "batch_fit.h"
#include "stream_fit.h"

namespace brdf {

// internal METHOD: 0 dlevmar_dif, 1 dlevmar_bc_dif / dlevmar_bc_der (ctx.analytic), 2 dlevmar_der
template <int METHOD>
using BatchMachine = typename std::conditional<METHOD == 0, DifMachine<kM>, typename std::conditional<METHOD == 1, BcMachine<kM>, DerMachine<kM>>::type>::type;

// Occupancy: the LM step between two passes runs on one lane (~3700 cycles on gfx950) while the rest of the
// workgroup waits, so throughput comes from OTHER workgroups on the same CU filling that time.  Measured: asking
// for 4 waves per SIMD (<= 128 VGPRs) pays for the one-sample-per-lane kernels (n <= 64: +20..50 % fits/s) but
// makes the larger geometries spill their register-resident samples (n = 4096 bc_dif: 3x slower), so only the
// smallest geometry is constrained.
template <int THREADS, int SPT>
constexpr int batch_waves_per_simd() { return (THREADS == 64 && SPT == 1) ? 4 : 2; }

template <int MODEL, int METHOD, bool FAST, int THREADS, int SPT>
__global__ __launch_bounds__(THREADS, (batch_waves_per_simd<THREADS, SPT>())) void batch_fit_kernel(BatchCtx ctx) {
  using Machine = BatchMachine<METHOD>;
  using Mdl = BrdfModel<MODEL>;
  __shared__ Machine sm;
  __shared__ PassUniforms<MODEL> su;
  __shared__ double red[reduce_buf_doubles<THREADS>()];
  __shared__ double sums[kSlots];
  __shared__ int bad_domain;
  __shared__ double dp_prev[kM + 1];  // dif: Dp and ||Dp||^2 of the last trial

  const int fit = blockIdx.x;
  const int tid = threadIdx.x;
  const bool first_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < kWave;  // see stream_fit.hip
  const int n = ctx.n;
  if (!FAST && ctx.flags[fit] != kNeedsExact) return;  // exact kernel: only the fits the fast kernel declined

  // ---- the fit's samples: one HBM read, then registers --------------------------------------------------
  const double *__restrict__ c0 = ctx.angles + (size_t)fit * 3 * n;
  const double *__restrict__ c1 = c0 + n;
  const double *__restrict__ c2 = c0 + 2 * (size_t)n;
  const double *__restrict__ xs = ctx.x + (size_t)fit * n;
  double s0[SPT], sx[SPT];
  Prep pq[SPT];
  bool ok[SPT];
  if (tid == 0) bad_domain = 0;
  __syncthreads();
  {
    double r1[SPT], r2[SPT];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      const int i = tid + k * THREADS;
      ok[k] = i < n;
      const int ii = ok[k] ? i : 0;
      s0[k] = c0[ii];
      r1[k] = Mdl::uses_c1 ? c1[ii] : 0.0;
      r2[k] = Mdl::uses_c2 ? c2[ii] : 0.0;
      sx[k] = xs[ii];
    }
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      pq[k] = Mdl::template prepare<FAST>(s0[k], r1[k], r2[k]);
      if (FAST && ok[k] && !Mdl::domain_ok(s0[k], r1[k], r2[k])) bad = true;
    }
    if (FAST && bad) bad_domain = 1;  // benign race: every writer stores 1
  }
  __syncthreads();
  if (FAST) {
    if (bad_domain) {  // log of a non-positive cosine: leave this fit to the exact kernel
      if (tid == 0) ctx.flags[fit] = kNeedsExact;
      return;
    }
    if (tid == 0) ctx.flags[fit] = 0;
  }

  // ---- LM state machine in LDS, stepped by lane 0 ---------------------------------------------------------
  if (tid == 0) {
    // (locals, not pointers into the by-value ctx: taking its members' addresses parks the whole struct in scratch)
    const double p0[kM] = {ctx.p[(size_t)fit * kM], ctx.p[(size_t)fit * kM + 1], ctx.p[(size_t)fit * kM + 2]};
    const double ov[5] = {ctx.opts[0], ctx.opts[1], ctx.opts[2], ctx.opts[3], ctx.opts[4]};
    const double lbv[kM] = {ctx.lb[0], ctx.lb[1], ctx.lb[2]}, ubv[kM] = {ctx.ub[0], ctx.ub[1], ctx.ub[2]};
    const double *opts = ctx.has_opts ? ov : nullptr;
    if constexpr (METHOD == 0) {
      sm.start(p0, n, ctx.itmax, opts, 0, /*speculative=*/1);
    } else if constexpr (METHOD == 1) {
      sm.start(p0, n, ctx.has_lb ? lbv : nullptr, ctx.has_ub ? ubv : nullptr, nullptr, ctx.itmax, opts, 0, ctx.multi);
      sm.c.analytic_jac = ctx.analytic;
    } else {
      sm.start(p0, n, ctx.itmax, opts, 0);
    }
  }
  __syncthreads();

  constexpr int DS = (METHOD == 0) ? SPT : 1;
  double hx[DS], wrk[DS], jac[DS][kM];  // dif only: f(p), f(q) and the secant Jacobian rows of this lane's samples
#pragma unroll
  for (int k = 0; k < DS; ++k) {
    hx[k] = wrk[k] = 0.0;
    jac[k][0] = jac[k][1] = jac[k][2] = 0.0;
  }

  int cur_sel_hx = 0, cur_sel_j = 0;
  for (;;) {
    const int kind = sm.h.req.kind;
    if (kind == RQ_DONE) break;
    if (first_wave) su.build(sm.h.req, /*need_base=*/METHOD != 0, METHOD == 2 || (METHOD == 1 && ctx.analytic));  // (dif keeps f(p) per sample: a trial evaluates f at q only)
    __syncthreads();
    const PassUniforms<MODEL> &u = su;
    if constexpr (METHOD == 0) {  // commit what the machine decided about the previous trial (speculative protocol,
                                  // one pass per LM iteration: lm_machine.h, DifMachine)
      if (sm.h.req.sel_j != cur_sel_j) {  // adopt the Broyden update J += ((wrk - hx - J Dp)/||Dp||^2) Dp^T, lm_core.c:760-766
        const double dpp[kM] = {dp_prev[0], dp_prev[1], dp_prev[2]};
#pragma unroll
        for (int k = 0; k < DS; ++k) {
          double jn[kM];
          broyden_row(jac[k], wrk[k], hx[k], dpp, dp_prev[kM], jn);
          jac[k][0] = jn[0];
          jac[k][1] = jn[1];
          jac[k][2] = jn[2];
        }
        cur_sel_j = sm.h.req.sel_j;
      }
      if (sm.h.req.sel_hx != cur_sel_hx) {  // step accepted: hx <- f(p + Dp)
#pragma unroll
        for (int k = 0; k < DS; ++k) hx[k] = wrk[k];
        cur_sel_hx = sm.h.req.sel_hx;
      }
    }
    double acc[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
    double mx = 0.0;

    switch (kind) {
    case RQ_EVAL:  // (each kernel is compiled with the request kinds its entry point issues only: register pressure)
      if constexpr (METHOD != 0) {
#pragma unroll
      for (int k = 0; k < SPT; ++k) {
        const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double e = ok[k] ? sx[k] - f : 0.0;
        acc[0] += e * e;
        mx = fmax(mx, fabs(e));
      }
      block_reduce<1, THREADS>(acc, mx, red, sums);
      }
      break;
    case RQ_EVAL_MULTI:
      if constexpr (METHOD == 1 && THREADS < 512) {  // the 512 x 8 geometry never asks for it (see batch_fit_enqueue): keeping the
                                       // unrolled 8 x 8 body out of that kernel keeps its samples in registers
        if (u.ncand == kMaxCand) {  // the usual case: no guard between a sample's candidates, four independent exp chains per basic
                                    // block (resident_fit_impl.h, RQ_EVAL_MULTI: a guarded candidate's chain runs alone)
#pragma unroll
          for (int k = 0; k < SPT; ++k) {
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
          for (int k = 0; k < SPT; ++k) {
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j)
              if (j < u.ncand) {
                const double e = ok[k] ? sx[k] - model_value_k<MODEL, FAST>(u, j, s0[k], pq[k]) : 0.0;
                acc[j] += e * e;
              }
          }
        }
        block_reduce<kMaxCand, THREADS>(acc, mx, red, sums);
      }
      break;
    case RQ_SCALED:
      if constexpr (METHOD == 1) {  // (only dlevmar_bc_*'s overflow guard issues it)
#pragma unroll
      for (int k = 0; k < SPT; ++k) {
        const double f = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double t = ok[k] ? (sx[k] - f) / u.scal : 0.0;
        acc[0] += t * t;
      }
      block_reduce<1, THREADS>(acc, mx, red, sums);
      }
      break;
    case RQ_JAC:
      if constexpr (METHOD != 0) {
      // (the row's kind -- analytic, central or forward differences -- is the same for every sample: chosen outside the sample
      // loop, whose bodies then are straight-line code in which the exp chains interleave; see resident_fit_impl.h, RQ_JAC)
      auto jac_rows = [&](auto jk) {
        constexpr int JK = decltype(jk)::value;
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
          double f0 = 0.0, j[kM];
          if (JK == 2)  // dlevmar_der / dlevmar_bc_der: the model's analytic Jacobian
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
      if (METHOD == 2 || u.analytic)
        jac_rows(std::integral_constant<int, 2>{});
      else if (u.central)
        jac_rows(std::integral_constant<int, 1>{});
      else
        jac_rows(std::integral_constant<int, 0>{});
      block_reduce<SumLayout<kM>::JAC, THREADS>(acc, mx, red, sums);
      }
      break;
    case RQ_DIF_INIT:
      if constexpr (METHOD == 0) {
#pragma unroll
      for (int k = 0; k < SPT; ++k) {
        hx[k] = model_value<MODEL, FAST>(u, s0[k], pq[k]);
        const double e = ok[k] ? sx[k] - hx[k] : 0.0;
        acc[0] += e * e;
      }
      block_reduce<1, THREADS>(acc, mx, red, sums);
      }
      break;
    case RQ_DIF_JAC:
      if constexpr (METHOD == 0) {
#pragma unroll
      for (int k = 0; k < SPT; ++k) {
        double f0 = 0.0;
        model_fd_row<MODEL, FAST>(u, s0[k], pq[k], false, f0, hx[k], true, jac[k]);
        double e = sx[k] - hx[k];
        if (!ok[k]) e = jac[k][0] = jac[k][1] = jac[k][2] = 0.0;
        acc_normal_eq(jac[k], e, acc, acc + kNL);
      }
      block_reduce<SumLayout<kM>::DIF_JAC, THREADS>(acc, mx, red, sums);
      }
      break;
    case RQ_DIF_TRIAL:  // speculative protocol: the Broyden-updated row is formed for the sums only; the rows
                        // themselves are updated (from wrk, hx) at the top of the next pass if the machine adopts it
      if constexpr (METHOD == 0) {
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
          const double w = model_value_q<MODEL, FAST>(u, s0[k], pq[k]);
          // broyden_row() (lm_core.c:760-766) spelled out: its scalar t is summed too -- the trial sweep leaves the NINE sums
          // of device_common.h (kTrialSums), which the stepping wave expands into the machine's thirteen
          double t = 0.0, jn[kM];
#pragma unroll
          for (int l = 0; l < kM; ++l) t += jac[k][l] * u.dp[l];
          t = (w - hx[k] - t) / u.dp_l2;
#pragma unroll
          for (int j = 0; j < kM; ++j) jn[j] = jac[k][j] + t * u.dp[j];
          double en = sx[k] - w, eo = sx[k] - hx[k];
          if (!ok[k]) en = eo = jn[0] = jn[1] = jn[2] = t = 0.0;
          wrk[k] = w;
          acc[0] += en * en;
#pragma unroll
          for (int j = 0; j < kM; ++j) acc[1 + j] += jac[k][j] * t;
          acc[1 + kM] += t * t;
#pragma unroll
          for (int j = 0; j < kM; ++j) acc[2 + kM + j] += jn[j] * en;
          acc[2 + 2 * kM] += t * eo;
        }
        block_reduce<kTrialSums, THREADS>(acc, mx, red, sums);
      }
      break;
    default: break;
    }
    if (first_wave) {
      if constexpr (METHOD == 0) {
        if (kind == RQ_DIF_TRIAL) {
          expand_trial_sums(static_cast<const typename Machine::Core &>(sm.h), sm.h.cool, su.dp, sums);
#pragma unroll
          for (int j = 0; j < kM; ++j) dp_prev[j] = su.dp[j];
          dp_prev[kM] = su.dp_l2;
        }
      }
      sm.template step<true>(sums, sums[kSums]);
    }
    __syncthreads();
  }

  if (tid == 0) {
    double *po = ctx.p + (size_t)fit * kM;
    for (int i = 0; i < kM; ++i) po[i] = sm.h.p[i];
    if (ctx.info)
      for (int i = 0; i < kInfoSz; ++i) ctx.info[(size_t)fit * kInfoSz + i] = sm.c.info[i];
    if (ctx.ret) ctx.ret[fit] = sm.c.ret;
  }
}

// ---------------------------------------------------------------------------------------------------
// n <= 16 (the application's own size: 16 lights per surfel, brdfdata.h:58): FOUR fits per wavefront.
// A fit owns one DPP row of 16 lanes (one sample per lane); its sums are reduced with four row_shr steps and
// land in the row's last lane, which also owns the fit's LM machine (LDS) and steps it -- the four row leaders
// step concurrently.  Rows pull fits from a global queue, so a row whose fit ends early starts the next one
// while its neighbours are still iterating (fits take 10..100+ LM iterations each).
// ---------------------------------------------------------------------------------------------------
constexpr int kRowLanes = 16;
constexpr int kRowsPerWave = kWave / kRowLanes;

// Occupancy, measured at n = 16 (2^18 fits): dlevmar_dif fits in 128 VGPRs with 39 spilled and still gains from four
// waves per SIMD (1.65e7 vs 1.17e7 fits/s at two); dlevmar_bc_dif spills 138 VGPRs at that budget and is 4-8 % faster
// with two waves per SIMD and none spilled.
template <int MODEL, int METHOD, bool FAST>
__global__ __launch_bounds__(kWave, (METHOD == 0 ? 4 : 2)) void batch_fit_rows_kernel(BatchCtx ctx, int *queue) {
  using Machine = BatchMachine<METHOD>;
  using Mdl = BrdfModel<MODEL>;
  __shared__ Machine sm[kRowsPerWave];
  __shared__ PassUniforms<MODEL> su[kRowsPerWave];
  __shared__ int s_fit[kRowsPerWave];
  __shared__ double s_dp[kRowsPerWave][kM + 1];  // dif: Dp and ||Dp||^2 of the row's last trial

  const int lane = threadIdx.x;
  const int row = lane / kRowLanes;
  const int j = lane % kRowLanes;
  const bool leader = (j == kRowLanes - 1);
  const int n = ctx.n;
  Machine &m = sm[row];
  const PassUniforms<MODEL> &u = su[row];

  bool row_live = true;  // false once the queue is empty for this row
  while (__any(row_live)) {
    // ---- fetch the next fit for every row that needs one ----------------------------------------------------
    if (row_live && leader) {
      int f;
      for (;;) {
        f = atomicAdd(queue, 1);
        if (f >= ctx.S) {
          f = -1;
          break;
        }
        if (FAST || ctx.flags[f] == kNeedsExact) break;  // exact twin: only the fits the fast kernel declined
      }
      s_fit[row] = f;
    }
    __syncthreads();
    const int fit = row_live ? s_fit[row] : -1;
    if (fit < 0) row_live = false;

    double s0 = 1.0, sx = 0.0, hx = 0.0, wrk = 0.0, jac[kM] = {0.0, 0.0, 0.0};
    Prep pq{1.0, 1.0};
    const bool ok = row_live && j < n;
    bool declined = false;
    if (row_live) {
      const double *c0 = ctx.angles + (size_t)fit * 3 * n;
      double r1 = 1.0, r2 = 1.0;
      if (ok) {
        s0 = c0[j];
        r1 = Mdl::uses_c1 ? c0[n + j] : 0.0;
        r2 = Mdl::uses_c2 ? c0[2 * (size_t)n + j] : 0.0;
        sx = ctx.x[(size_t)fit * n + j];
      }
      pq = Mdl::template prepare<FAST>(s0, r1, r2);
      if (FAST) {  // a cosine <= 0 anywhere in this fit (row-wide OR through the DPP max): leave it to the exact twin
        const double bad = row_reduce_to_last<OpMax>((ok && !Mdl::domain_ok(s0, r1, r2)) ? 1.0 : 0.0);
        if (leader) ctx.flags[fit] = (bad > 0.0) ? kNeedsExact : 0;
        if (leader) s_fit[row] = (bad > 0.0) ? -2 : fit;
      }
    }
    __syncthreads();
    if (FAST && row_live && s_fit[row] == -2) declined = true;
    const bool run = row_live && !declined;

    if (run && leader) {
      const double p0[kM] = {ctx.p[(size_t)fit * kM], ctx.p[(size_t)fit * kM + 1], ctx.p[(size_t)fit * kM + 2]};
      const double ov[5] = {ctx.opts[0], ctx.opts[1], ctx.opts[2], ctx.opts[3], ctx.opts[4]};
      const double lbv[kM] = {ctx.lb[0], ctx.lb[1], ctx.lb[2]}, ubv[kM] = {ctx.ub[0], ctx.ub[1], ctx.ub[2]};
      const double *opts = ctx.has_opts ? ov : nullptr;
      if constexpr (METHOD == 0)
        m.start(p0, n, ctx.itmax, opts, 0, /*speculative=*/1);
      else
        m.start(p0, n, ctx.has_lb ? lbv : nullptr, ctx.has_ub ? ubv : nullptr, nullptr, ctx.itmax, opts, 0);
    }
    __syncthreads();

    // ---- the fit: passes until this row's machine is done (rows are in different passes: divergent switch) --
    bool busy = run;
    int cur_sel_hx = 0, cur_sel_j = 0;  // dif, speculative protocol: what this lane's hx / Jacobian row currently reflect
    while (__any(busy)) {
      if (busy && leader) {
        if (m.h.req.kind != RQ_DONE) su[row].build(m.h.req, /*need_base=*/METHOD != 0);
      }
      __syncthreads();
      const int kind = busy ? m.h.req.kind : RQ_DONE;
      if (kind == RQ_DONE) busy = false;
      if constexpr (METHOD == 0) {
        if (busy) {  // commit what the machine decided about the previous trial (see batch_fit_kernel)
          if (m.h.req.sel_j != cur_sel_j) {
            const double dpp[kM] = {s_dp[row][0], s_dp[row][1], s_dp[row][2]};
            double jn[kM];
            broyden_row(jac, wrk, hx, dpp, s_dp[row][kM], jn);
            jac[0] = jn[0];
            jac[1] = jn[1];
            jac[2] = jn[2];
            cur_sel_j = m.h.req.sel_j;
          }
          if (m.h.req.sel_hx != cur_sel_hx) {
            hx = wrk;
            cur_sel_hx = m.h.req.sel_hx;
          }
        }
      }
      double acc[kSums];
#pragma unroll
      for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
      double mx = 0.0;
      int ns = 0;
      if (busy) {
        switch (kind) {
        case RQ_EVAL: {
          const double e = ok ? sx - model_value<MODEL, FAST>(u, s0, pq) : 0.0;
          acc[0] = e * e;
          mx = fabs(e);
          ns = 1;
          break;
        }
        case RQ_SCALED: {
          const double t = ok ? (sx - model_value<MODEL, FAST>(u, s0, pq)) / u.scal : 0.0;
          acc[0] = t * t;
          ns = 1;
          break;
        }
        case RQ_JAC: {
          double f0 = 0.0, jr[kM];
          model_fd_row<MODEL, FAST>(u, s0, pq, true, f0, 0.0, false, jr);
          double e = sx - f0;
          if (!ok) e = jr[0] = jr[1] = jr[2] = 0.0;
          acc_normal_eq(jr, e, acc, acc + kNL);
          acc[kNL + kM] = e * e;
          ns = SumLayout<kM>::JAC;
          break;
        }
        case RQ_DIF_INIT: {
          hx = model_value<MODEL, FAST>(u, s0, pq);
          const double e = ok ? sx - hx : 0.0;
          acc[0] = e * e;
          ns = 1;
          break;
        }
        case RQ_DIF_JAC: {
          double f0 = 0.0;
          model_fd_row<MODEL, FAST>(u, s0, pq, false, f0, hx, true, jac);
          double e = sx - hx;
          if (!ok) e = jac[0] = jac[1] = jac[2] = 0.0;
          acc_normal_eq(jac, e, acc, acc + kNL);
          ns = SumLayout<kM>::DIF_JAC;
          break;
        }
        case RQ_DIF_TRIAL: {  // speculative protocol: sums of the Broyden-updated row; the row itself is updated next pass
          const double w = model_value_q<MODEL, FAST>(u, s0, pq);
          double jn[kM];
          broyden_row(jac, w, hx, u.dp, u.dp_l2, jn);
          double en = sx - w, eo = sx - hx;
          if (!ok) en = eo = jn[0] = jn[1] = jn[2] = 0.0;
          wrk = w;
          acc[0] = en * en;
          acc_normal_eq(jn, en, acc + 1, acc + 1 + kNL);
          acc[1 + kNL + kM + 0] = jn[0] * eo;
          acc[1 + kNL + kM + 1] = jn[1] * eo;
          acc[1 + kNL + kM + 2] = jn[2] * eo;
          ns = SumLayout<kM>::DIF_TRIAL;
          break;
        }
        default: break;
        }
      }
      // row reductions: DPP only (every lane takes part; idle rows reduce zeros)
      double sums[kSlots];
      // (all thirteen sums here, not the nine of the other kernels: expanding them on four divergent row leaders costs more
      // than the four row reductions it saves -- measured 1.66e7 against 1.94e7 fits/s)
#pragma unroll
      for (int k = 0; k < kSums; ++k) sums[k] = row_reduce_to_last<OpSum>(acc[k]);
      sums[kSums] = row_reduce_to_last<OpMax>(mx);
      (void)ns;
      if (busy && leader) {
        if (METHOD == 0 && kind == RQ_DIF_TRIAL) {
          for (int k = 0; k < kM; ++k) s_dp[row][k] = u.dp[k];
          s_dp[row][kM] = u.dp_l2;
        }
        m.step(sums, sums[kSums]);
      }
      __syncthreads();
    }

    if (run && leader) {
      double *po = ctx.p + (size_t)fit * kM;
      for (int i = 0; i < kM; ++i) po[i] = m.h.p[i];
      if (ctx.info)
        for (int i = 0; i < kInfoSz; ++i) ctx.info[(size_t)fit * kInfoSz + i] = m.c.info[i];
      if (ctx.ret) ctx.ret[fit] = m.c.ret;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// synthetic sample generator (bench support): the counter stream of brdf_amd/synth.py on the device
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double synth_uniform(unsigned long long seed, unsigned long long index) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (index + 1ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <int MODEL>
__global__ __launch_bounds__(256) void synth_kernel(unsigned long long seed, long long first, int count, int n,
                                                    const double *__restrict__ truth, double *__restrict__ angles,
                                                    double *__restrict__ x) {
  using Mdl = BrdfModel<MODEL>;
  const long long total = (long long)count * n;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long s = g / n;
    const int i = (int)(g - s * n);
    const unsigned long long base = (unsigned long long)(first + s) * 4ULL;
    double c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      c[k] = 0.05 + 0.95 * synth_uniform(seed, (base + k) * (unsigned long long)n + i);
      angles[((size_t)s * 3 + k) * n + i] = c[k];
    }
    const double p[kM] = {truth[s * 3 + 0], truth[s * 3 + 1], truth[s * 3 + 2]};
    const Lin l = Mdl::lin(p);
    const Nl nl = Mdl::nl(p);
    const double f = Mdl::combine(l, c[0], Mdl::template shape<false>(nl, c[0], Mdl::template prepare<false>(c[0], c[1], c[2])));
    x[(size_t)s * n + i] = f + 0.01 * (synth_uniform(seed, (base + 3) * (unsigned long long)n + i) - 0.5);
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
namespace {

using BatchFn = void (*)(BatchCtx);

template <int THREADS, int SPT>
BatchFn pick(int model, int method, bool fast) {
  static const BatchFn table[2][MODEL_COUNT][3] = {
      {{batch_fit_kernel<0, 0, false, THREADS, SPT>, batch_fit_kernel<0, 1, false, THREADS, SPT>, batch_fit_kernel<0, 2, false, THREADS, SPT>},
       {batch_fit_kernel<1, 0, false, THREADS, SPT>, batch_fit_kernel<1, 1, false, THREADS, SPT>, batch_fit_kernel<1, 2, false, THREADS, SPT>},
       {nullptr, nullptr, nullptr}},  // Ward's prepared path has no domain restriction: no exact twin needed
      {{batch_fit_kernel<0, 0, true, THREADS, SPT>, batch_fit_kernel<0, 1, true, THREADS, SPT>, batch_fit_kernel<0, 2, true, THREADS, SPT>},
       {batch_fit_kernel<1, 0, true, THREADS, SPT>, batch_fit_kernel<1, 1, true, THREADS, SPT>, batch_fit_kernel<1, 2, true, THREADS, SPT>},
       {batch_fit_kernel<2, 0, true, THREADS, SPT>, batch_fit_kernel<2, 1, true, THREADS, SPT>, batch_fit_kernel<2, 2, true, THREADS, SPT>}},
  };
  return table[fast ? 1 : 0][model][method];
}

struct Geometry {
  int threads, spt;
};
// (measured and rejected for 1024 < n <= 4096, bc_dif: 256 threads x 16 samples per lane to get two workgroups per
// CU -- the fully unrolled 16-sample sweep spills ~1.2 KB per lane and runs 2x slower than 512 x 8)
// one wavefront per fit up to 256 samples, one workgroup per fit up to 4096
bool geometry_for(int n, Geometry *g) {
  if (n <= 64) *g = {64, 1};
  else if (n <= 256) *g = {64, 4};
  else if (n <= 1024) *g = {256, 4};
  else if (n <= 4096) *g = {512, 8};
  else return false;
  return true;
}

BatchFn kernel_for(const Geometry &g, int model, int method, bool fast) {
  if (g.threads == 64 && g.spt == 1) return pick<64, 1>(model, method, fast);
  if (g.threads == 64 && g.spt == 4) return pick<64, 4>(model, method, fast);
  if (g.threads == 256) return pick<256, 4>(model, method, fast);
  return pick<512, 8>(model, method, fast);
}

// Per-thread scratch of a batch call: flags[S] (kNeedsExact marks, read by the second launch) and two work-queue
// counters.  A call is asynchronous on the caller's stream, so the NEXT call of this thread -- possibly on another
// stream -- must not touch the scratch before the previous call's launches have finished with it: every call waits
// (device-side, hipStreamWaitEvent) on the event the previous call recorded behind its last launch.
struct BatchScratch {
  int *ptr = nullptr;
  size_t cap = 0;  // ints, without the two queue words
  int device = -1;
  hipEvent_t last_use = nullptr;
  bool in_use = false;
};
thread_local BatchScratch g_scratch;

}  // namespace

#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return kLmError;                                                                \
    }                                                                                 \
  } while (0)

namespace {

using RowsFn = void (*)(BatchCtx, int *);
RowsFn rows_kernel(int model, int method, bool fast) {
  static const RowsFn table[2][MODEL_COUNT][2] = {
      {{batch_fit_rows_kernel<0, 0, false>, batch_fit_rows_kernel<0, 1, false>},
       {batch_fit_rows_kernel<1, 0, false>, batch_fit_rows_kernel<1, 1, false>},
       {nullptr, nullptr}},
      {{batch_fit_rows_kernel<0, 0, true>, batch_fit_rows_kernel<0, 1, true>},
       {batch_fit_rows_kernel<1, 0, true>, batch_fit_rows_kernel<1, 1, true>},
       {batch_fit_rows_kernel<2, 0, true>, batch_fit_rows_kernel<2, 1, true>}},
  };
  return table[fast ? 1 : 0][model][method];
}

// n <= 16: four fits per wavefront or one wave per fit?  Measured (2^18 fits of 16 samples, Blinn-Phong / Ward):
//   dlevmar_dif     rows kernel 1.95e7 / 1.46e7 fits/s, wave per fit 8.2e6 / 6.3e6   -> rows kernel
//   dlevmar_bc_dif  rows kernel 2.99e6 / 1.27e6,        wave per fit 3.95e6 / 2.09e6 -> wave per fit: bc_dif's long,
//                   divergent steps (line search, projected gradient) serialise over the four row leaders, and the
//                   wave-per-fit kernel evaluates up to 8 projected-gradient candidates per pass
// BRDF_HIP_ROWS=0: never the rows kernel; BRDF_HIP_ROWS=1: the rows kernel for both entry points.
bool rows_path_enabled(int method) {
  const char *e = getenv("BRDF_HIP_ROWS");
  if (e && e[0] == '0') return false;
  if (e && e[0] == '1') return true;
  return method == 0;
}

// four fits per wavefront, rows pull work from a queue: a few waves per SIMD on every CU are enough
int rows_enqueue(const BatchFitArgs &a, const BatchCtx &c, bool fast, int *queue) {
  int dev = 0, cus = 0;
  HIP_OK(hipGetDevice(&dev));
  HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  long long waves = (long long)cus * 16;  // 4 waves per SIMD
  const long long need = ((long long)a.S + kRowsPerWave - 1) / kRowsPerWave;
  if (waves > need) waves = need;
  if (fast) {
    hipLaunchKernelGGL(rows_kernel(a.model, a.method, true), dim3((unsigned)waves), dim3(kWave), 0, a.stream, c, queue);
    HIP_OK(hipGetLastError());
    if (a.model != MODEL_WARD) {
      hipLaunchKernelGGL(rows_kernel(a.model, a.method, false), dim3((unsigned)waves), dim3(kWave), 0, a.stream, c, queue + 1);
      HIP_OK(hipGetLastError());
    }
  } else {
    HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c.flags), kNeedsExact, (size_t)a.S, a.stream));
    hipLaunchKernelGGL(rows_kernel(a.model, a.method, false), dim3((unsigned)waves), dim3(kWave), 0, a.stream, c, queue);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

// n <= 16, dlevmar_bc_dif: one lane per fit (lane_fit.hip).  Measured (2^20 fits of 16 samples): see DESIGN.md section 6.
// BRDF_HIP_LANE=0: the wave-per-fit / rows kernels of this file instead.
bool lane_path_enabled() {
  const char *e = getenv("BRDF_HIP_LANE");
  return !(e && e[0] == '0');
}

}  // namespace

// BRDF_HIP_BATCH_BIG=0: the symmetric 512 x 8 geometry of this file instead of the control-wave kernel for 1024 < n <= 4096
// BRDF_HIP_BATCH_DIF_CHAIN=k: the eight-wave batched dlevmar_dif kernel's trial points per sweep in a chain of rejections
// (default: the single fits' setting, BRDF_HIP_DIF_CHAIN)
static int batch_dif_chain() {
  const char *e = getenv("BRDF_HIP_BATCH_DIF_CHAIN");
  if (!e) return dif_chain_candidates();
  const int k = atoi(e);
  return k < 1 ? 1 : (k > kMaxCand ? kMaxCand : k);
}

static bool big_path_enabled() {
  const char *e = getenv("BRDF_HIP_BATCH_BIG");
  return !(e && e[0] == '0');
}

namespace {
int batch_fit_launches(const BatchFitArgs &a, const Geometry &g, int *flags, int *queue) {
  // the C ABI's method -> the kernels' machine (0 Dif, 1 Bc, 2 Der) + where the Jacobian rows come from
  const int method = (a.method == BRDF_METHOD_BC_DER) ? 1 : (a.method == BRDF_METHOD_DER ? 2 : a.method);
  BatchCtx c;
  memset(&c, 0, sizeof c);
  c.analytic = (a.method == BRDF_METHOD_BC_DER || a.method == BRDF_METHOD_DER) ? 1 : 0;
  c.angles = a.d_angles;
  c.x = a.d_x;
  c.p = a.d_p;
  c.info = a.d_info;
  c.ret = a.d_ret;
  c.flags = flags;
  c.S = a.S;
  c.n = a.n;
  c.itmax = a.itmax;
  c.has_opts = a.opts != nullptr;
  c.has_lb = a.lb != nullptr;
  c.has_ub = a.ub != nullptr;
  // projected-gradient candidates per sweep: pays where the LM step dominates a pass; the 512 x 8 geometry would
  // spill its register-resident samples with 8 unrolled candidates (measured 2.4x slower), so it stays at one
  c.multi = (g.threads == 512) ? 1 : pg_candidates();
  c.chain = 1;     // (only the eight-wave kernel compiles the chains in)
  c.spec_jac = 0;  // batched fits are bound by arithmetic: a Jacobian pass that is not used costs three evaluations
  for (int i = 0; i < 5; ++i) c.opts[i] = a.opts ? a.opts[i] : 0.0;
  for (int i = 0; i < kM; ++i) {
    c.lb[i] = a.lb ? a.lb[i] : 0.0;
    c.ub[i] = a.ub ? a.ub[i] : 0.0;
  }
  HIP_OK(hipMemsetAsync(queue, 0, 2 * sizeof(int), a.stream));
  const bool fast = brdf_fast_path_enabled() || a.model == MODEL_WARD;
  if (a.n <= kLaneMaxN && method == 1 && lane_path_enabled()) return lane_fit_enqueue(a.model, fast, c, queue, a.stream);
  if (a.n <= kRowLanes && method != 2 && !c.analytic && rows_path_enabled(method)) {
    BatchFitArgs b = a;
    b.method = method;
    return rows_enqueue(b, c, fast, queue);
  }
  if (g.threads == 512 && (big_path_enabled() || method == 2 || c.analytic)) {  // 1024 < n <= 4096: eight waves per fit (resident_fit.hip)
    c.multi = pg_candidates();
    c.chain = batch_dif_chain();
    if (!fast) HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flags), kNeedsExact, (size_t)a.S, a.stream));
    return resident_batch_enqueue(a.model, method, fast, c, a.stream);
  }
  const dim3 grid(a.S), block(g.threads);
  if (fast) {
    hipLaunchKernelGGL(kernel_for(g, a.model, method, true), grid, block, 0, a.stream, c);
    HIP_OK(hipGetLastError());
    if (a.model != MODEL_WARD) {  // fits with a cosine <= 0 marked themselves: second launch on the exact path
      hipLaunchKernelGGL(kernel_for(g, a.model, method, false), grid, block, 0, a.stream, c);
      HIP_OK(hipGetLastError());
    }
  } else {
    // BRDF_HIP_EXACT_POW=1: mark every fit for the exact kernel
    HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flags), kNeedsExact, (size_t)a.S, a.stream));
    hipLaunchKernelGGL(kernel_for(g, a.model, method, false), grid, block, 0, a.stream, c);
    HIP_OK(hipGetLastError());
  }
  return 0;
}
}  // namespace

// n > 4096 samples per fit: a fit no longer fits one workgroup's registers + LDS, but it fits the CHIP -- the fits run one
// after the other through the single-fit path (resident regime up to #CUs * 4096 samples, the launch chain beyond), each
// using every CU.  The starting points travel to the host and the results back: synchronous on a.stream.
namespace {
int batch_of_large_fits(const BatchFitArgs &a) {
  std::vector<double> p((size_t)a.S * kM), info((size_t)a.S * kInfoSz);
  std::vector<int> ret(a.S);
  HIP_OK(hipMemcpyAsync(p.data(), a.d_p, sizeof(double) * p.size(), hipMemcpyDeviceToHost, a.stream));
  HIP_OK(hipStreamSynchronize(a.stream));
  for (int s = 0; s < a.S; ++s) {
    StreamFitArgs f;
    f.method = (a.method == BRDF_METHOD_BC_DER) ? 1 : (a.method == BRDF_METHOD_DER ? 2 : a.method);
    f.analytic = (a.method == BRDF_METHOD_BC_DER || a.method == BRDF_METHOD_DER) ? 1 : 0;
    f.model = a.model;
    f.d_angles = a.d_angles + (size_t)s * 3 * a.n;
    f.d_x = a.d_x + (size_t)s * a.n;
    f.n = a.n;
    f.p = p.data() + (size_t)s * kM;
    f.lb = a.lb;
    f.ub = a.ub;
    f.dscl = nullptr;
    f.itmax = a.itmax;
    f.opts = a.opts;
    f.info = info.data() + (size_t)s * kInfoSz;
    f.covar = nullptr;
    f.stream = a.stream;
    ret[s] = stream_fit_run(f);
  }
  HIP_OK(hipMemcpyAsync(a.d_p, p.data(), sizeof(double) * p.size(), hipMemcpyHostToDevice, a.stream));
  if (a.d_info) HIP_OK(hipMemcpyAsync(a.d_info, info.data(), sizeof(double) * info.size(), hipMemcpyHostToDevice, a.stream));
  if (a.d_ret) HIP_OK(hipMemcpyAsync(a.d_ret, ret.data(), sizeof(int) * ret.size(), hipMemcpyHostToDevice, a.stream));
  HIP_OK(hipStreamSynchronize(a.stream));  // (the host vectors above are about to go away)
  return 0;
}
}  // namespace

int batch_fit_enqueue(const BatchFitArgs &a) {
  if (a.model < 0 || a.model >= MODEL_COUNT || a.method < 0 || a.method > BRDF_METHOD_DER) {
    set_error("brdf_hip_fit_batch_dev(): unknown model %d / method %d", a.model, a.method);
    return kLmError;
  }
  if (!a.d_angles || !a.d_x || !a.d_p || a.S <= 0 || a.n <= 0) {
    set_error("brdf_hip_fit_batch_dev(): null device pointer or non-positive S/n");
    return kLmError;
  }
  Geometry g;
  if (!geometry_for(a.n, &g)) return batch_of_large_fits(a);  // n > 4096: one fit after the other, each spread over the chip
  if ((a.method == 1 || a.method == BRDF_METHOD_BC_DER) && a.lb && a.ub)
    for (int i = 0; i < kM; ++i)
      if (a.lb[i] > a.ub[i]) {  // lmbc_core.c:451-454
        set_error("dlevmar_bc_dif(): at least one lower bound exceeds the upper one");
        return kLmError;
      }
  (void)hipGetLastError();
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  BatchScratch &sc = g_scratch;
  if (sc.device != dev || sc.cap < (size_t)a.S) {
    if (sc.ptr) {
      HIP_OK(hipDeviceSynchronize());  // a previous batch (any stream) may still use the old block
      (void)hipFree(sc.ptr);
      if (sc.last_use) (void)hipEventDestroy(sc.last_use);
      sc = BatchScratch{};
    }
    HIP_OK(hipMalloc(&sc.ptr, sizeof(int) * ((size_t)a.S + 2)));
    HIP_OK(hipEventCreateWithFlags(&sc.last_use, hipEventDisableTiming));
    sc.cap = (size_t)a.S;
    sc.device = dev;
  }
  if (sc.in_use) HIP_OK(hipStreamWaitEvent(a.stream, sc.last_use, 0));
  const int rc = batch_fit_launches(a, g, sc.ptr, sc.ptr + sc.cap);
  sc.in_use = true;  // (also after a failed enqueue: some launches may be in flight)
  HIP_OK(hipEventRecord(sc.last_use, a.stream));
  return rc;
}

int synth_enqueue(int model, unsigned long long seed, long long first, int count, int n, const double *d_truth,
                  double *d_angles, double *d_x, hipStream_t stream) {
  if (model < 0 || model >= MODEL_COUNT || !d_truth || !d_angles || !d_x || count <= 0 || n <= 0) {
    set_error("brdf_hip_synth_dev(): bad arguments");
    return kLmError;
  }
  (void)hipGetLastError();
  const long long total = (long long)count * n;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  switch (model) {
  case 0: hipLaunchKernelGGL(synth_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, seed, first, count, n, d_truth, d_angles, d_x); break;
  case 1: hipLaunchKernelGGL(synth_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, seed, first, count, n, d_truth, d_angles, d_x); break;
  default: hipLaunchKernelGGL(synth_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, seed, first, count, n, d_truth, d_angles, d_x); break;
  }
  HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace brdf
