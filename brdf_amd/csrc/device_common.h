// device_common.h -- shared device-side pieces of the BRDF fitter kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include "brdf_models.h"

namespace brdf {

constexpr int kWave = 64;                       // CDNA wavefront
constexpr int kSums = SumLayout<kM>::MAX;        // 13 reduced sums at most
constexpr int kSlots = kSums + 1;                // + one max-reduced slot (max |e|)
constexpr int kNL = SumLayout<kM>::NL;           // 6 lower-triangle entries

// ---- wave64 reductions on the VALU data-parallel-primitive path (no LDS traffic, no ds_bpermute) --------
// v_mov_b32_dpp moves a lane's dword to another lane inside the SIMD datapath: row_shr:n shifts inside a
// row of 16 lanes, row_bcast:15 / row_bcast:31 carry a row's last lane into the following row(s).  Four
// shift/add steps + two broadcasts leave the sum of all 64 lanes in lane 63 (the classic GCN/CDNA tree).
// A double is moved as two dwords.  Lanes with no source read `idv` (the identity of the operator).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_move(double v, double idv) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(idv), __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(idv), __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}

struct OpSum {
  static __device__ __forceinline__ double id() { return 0.0; }
  static __device__ __forceinline__ double apply(double a, double b) { return a + b; }
};
struct OpMax {  // operands are >= 0 here (|e|), so 0 is the identity
  static __device__ __forceinline__ double id() { return 0.0; }
  static __device__ __forceinline__ double apply(double a, double b) { return fmax(a, b); }
};

// result is valid in lane 63 only
template <class OP>
__device__ __forceinline__ double wave_reduce_to_last(double v) {
  v = OP::apply(v, dpp_move<0x111, 0xf, 0xf>(v, OP::id()));  // row_shr:1
  v = OP::apply(v, dpp_move<0x112, 0xf, 0xf>(v, OP::id()));  // row_shr:2
  v = OP::apply(v, dpp_move<0x114, 0xf, 0xf>(v, OP::id()));  // row_shr:4
  v = OP::apply(v, dpp_move<0x118, 0xf, 0xf>(v, OP::id()));  // row_shr:8   -> lane 15 of each row = row total
  v = OP::apply(v, dpp_move<0x142, 0xa, 0xf>(v, OP::id()));  // row_bcast:15 into rows 1 and 3
  v = OP::apply(v, dpp_move<0x143, 0xc, 0xf>(v, OP::id()));  // row_bcast:31 into rows 2 and 3 -> lane 63 = total
  return v;
}

// two independent reductions side by side (the six steps of one tree are a dependent chain: interleaving two fills it)
template <class OPA, class OPB>
__device__ __forceinline__ void wave_reduce2_to_last(double &a, double &b) {
  a = OPA::apply(a, dpp_move<0x111, 0xf, 0xf>(a, OPA::id()));
  b = OPB::apply(b, dpp_move<0x111, 0xf, 0xf>(b, OPB::id()));
  a = OPA::apply(a, dpp_move<0x112, 0xf, 0xf>(a, OPA::id()));
  b = OPB::apply(b, dpp_move<0x112, 0xf, 0xf>(b, OPB::id()));
  a = OPA::apply(a, dpp_move<0x114, 0xf, 0xf>(a, OPA::id()));
  b = OPB::apply(b, dpp_move<0x114, 0xf, 0xf>(b, OPB::id()));
  a = OPA::apply(a, dpp_move<0x118, 0xf, 0xf>(a, OPA::id()));
  b = OPB::apply(b, dpp_move<0x118, 0xf, 0xf>(b, OPB::id()));
  a = OPA::apply(a, dpp_move<0x142, 0xa, 0xf>(a, OPA::id()));
  b = OPB::apply(b, dpp_move<0x142, 0xa, 0xf>(b, OPB::id()));
  a = OPA::apply(a, dpp_move<0x143, 0xc, 0xf>(a, OPA::id()));
  b = OPB::apply(b, dpp_move<0x143, 0xc, 0xf>(b, OPB::id()));
}

// reduction over one DPP row (16 lanes); result valid in the row's last lane (15, 31, 47, 63)
template <class OP>
__device__ __forceinline__ double row_reduce_to_last(double v) {
  v = OP::apply(v, dpp_move<0x111, 0xf, 0xf>(v, OP::id()));  // row_shr:1
  v = OP::apply(v, dpp_move<0x112, 0xf, 0xf>(v, OP::id()));  // row_shr:2
  v = OP::apply(v, dpp_move<0x114, 0xf, 0xf>(v, OP::id()));  // row_shr:4
  v = OP::apply(v, dpp_move<0x118, 0xf, 0xf>(v, OP::id()));  // row_shr:8
  return v;
}

// Workgroup reduction of NS sums and one max.  The order is a pure function of THREADS: results are
// reproducible run to run.  out[0..NS) receive the sums, out[kSums] the max; visible to all threads on return.
//
// A DPP chain per value per wave would serialise ~14 dependent chains in every wave (Jacobian / Broyden passes reduce 9..13
// sums), so the values are transposed through LDS instead: every thread parks its NS+1 values as buf[slot][thread]; then
// wave w owns slots {w, w+NW, ..}: each lane adds THREADS/64 entries of the slot in a fixed order and ONE DPP tree per slot
// finishes it.  The order in which one slot is summed does NOT depend on NS (an evaluation pass used to have a path of its
// own: per-wave trees first): the machines compare sums of squares that come from passes of different kinds -- a trial point
// evaluated by an evaluation pass, by a multi-candidate pass or by the Jacobian pass at that point (lm_machine.h: spec_jac)
// -- and with one order those are the same bits, as they are in the reference, where one function forms them all.
template <int THREADS>
constexpr int reduce_buf_doubles() { return kSlots * THREADS; }

template <int NS, int THREADS>
__device__ __forceinline__ void block_reduce(const double *acc, double mx, double *buf, double *out) {
  constexpr int NW = THREADS / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  {
#pragma unroll
    for (int k = 0; k < NS; ++k) buf[k * THREADS + threadIdx.x] = acc[k];
    buf[NS * THREADS + threadIdx.x] = mx;
    __syncthreads();
    for (int k = wave; k <= NS; k += NW) {  // wave-uniform loop: slot k belongs to this wave
      const double *src = buf + k * THREADS + lane;
      double s = src[0];
      if (k < NS) {
#pragma unroll
        for (int j = 1; j < NW; ++j) s += src[j * kWave];
        s = wave_reduce_to_last<OpSum>(s);
        if (lane == kWave - 1) out[k] = s;
      } else {
#pragma unroll
        for (int j = 1; j < NW; ++j) s = fmax(s, src[j * kWave]);
        s = wave_reduce_to_last<OpMax>(s);
        if (lane == kWave - 1) out[kSums] = s;
      }
    }
    __syncthreads();
  }
}

// The same over `count` (<= THREADS) per-workgroup partial rows of the previous launch, already loaded by the
// caller (one value per slot per thread, zero beyond `count`).
template <int NS, int THREADS>
__device__ __forceinline__ void fold_rows(const double *v, double *buf, double *out) {
  block_reduce<NS, THREADS>(v, v[kSums], buf, out);
}

// a wave-uniform value (read from LDS into a vector register) moved into scalar registers
__device__ __forceinline__ double scalar_copy(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ Lin scalar_copy(const Lin &l) { return Lin{scalar_copy(l.a), scalar_copy(l.b)}; }
__device__ __forceinline__ Nl scalar_copy(const Nl &n) { return Nl{scalar_copy(n.u0), scalar_copy(n.u1)}; }

// what an evaluation / a Jacobian pass reads of a request's uniforms (PassUniforms<MODEL>'s field names: the model functions of
// brdf_models.h take either), small enough to live in scalar registers for a sweep
struct EvalUniforms {
  Lin l0;
  Nl n0;
  double scal;
};
struct JacUniforms {
  Lin l0, lp[kM], lm[kM];
  Nl n0, np2, nm2;
  double dinv[kM], an[2];
  int central, analytic;
};

__device__ __forceinline__ EvalUniforms scalar_copy(const EvalUniforms &u) { return EvalUniforms{scalar_copy(u.l0), scalar_copy(u.n0), scalar_copy(u.scal)}; }
__device__ __forceinline__ JacUniforms scalar_copy(const JacUniforms &u) {  // (only the fields its `analytic` / `central` say are set)
  JacUniforms r;
  r.l0 = scalar_copy(u.l0);
  r.n0 = scalar_copy(u.n0);
  r.analytic = u.analytic;
  r.central = u.central;
  if (u.analytic) {
    r.an[0] = scalar_copy(u.an[0]);
    r.an[1] = scalar_copy(u.an[1]);
  } else {
    for (int j = 0; j < kM; ++j) {
      r.lp[j] = scalar_copy(u.lp[j]);
      r.dinv[j] = scalar_copy(u.dinv[j]);
      if (u.central) r.lm[j] = scalar_copy(u.lm[j]);
    }
    r.np2 = scalar_copy(u.np2);
    if (u.central) r.nm2 = scalar_copy(u.nm2);
  }
  return r;
}

// ---- the nine sums of a dlevmar_dif trial sweep (resident_fit.hip explains them where they are accumulated) -------------
constexpr int kTrialSums = 3 + 2 * kM;  // what a dlevmar_dif trial sweep reduces: [e'^2, J^T t (3), t^T t, J'^T e' (3), t^T e]

// ... and what the machine wants (SumLayout::DIF_TRIAL: [e'^2, J'^T J' lower (6), J'^T e' (3), J'^T e (3)]), from them and from the
// products of the CURRENT Jacobian the machine holds while a trial is out: jtj (its diagonal carries mu: the plain one is in
// diag) and jte.  Executed by the control wave, all lanes the same values; in place.
template <class Core, class Cool>
__device__ __forceinline__ void expand_trial_sums(const Core &core, const Cool &cool, const double *dpv, double *sums) {
  const double s0 = sums[0], a[kM] = {sums[1], sums[2], sums[3]}, b = sums[1 + kM];
  const double d[kM] = {sums[2 + kM], sums[3 + kM], sums[4 + kM]}, te = sums[2 + 2 * kM];
  const double dp[kM] = {dpv[0], dpv[1], dpv[2]};
  double out[SumLayout<kM>::DIF_TRIAL];
  out[0] = s0;
  int c = 1;
#pragma unroll
  for (int i = 0; i < kM; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j, ++c) {
      double v = (i == j) ? cool.diag[i] : core.jtj[i * kM + j];
      v = fma(a[i], dp[j], v);
      v = fma(dp[i], a[j], v);
      out[c] = fma(b * dp[i], dp[j], v);
    }
#pragma unroll
  for (int j = 0; j < kM; ++j) {
    out[1 + kNL + j] = d[j];
    out[1 + kNL + kM + j] = fma(dp[j], te, core.jte[j]);
  }
#pragma unroll
  for (int j = 0; j < SumLayout<kM>::DIF_TRIAL; ++j) sums[j] = out[j];
}


// number of sum slots a request kind produces
__host__ __device__ __forceinline__ int slots_of(int kind) {
  switch (kind) {
  case RQ_EVAL:
  case RQ_SCALED:
  case RQ_DIF_INIT: return 1;
  case RQ_JAC: return SumLayout<kM>::JAC;
  case RQ_DIF_JAC:
  case RQ_DIF_UPDATE: return SumLayout<kM>::DIF_JAC;
  case RQ_DIF_TRIAL: return SumLayout<kM>::DIF_TRIAL;
  case RQ_EVAL_MULTI: return kMaxCand;
  default: return 0;
  }
}

// accumulate J^T J (lower triangle, row-major order (0,0),(1,0),(1,1),(2,0),(2,1),(2,2)) and J^T e
__device__ __forceinline__ void acc_normal_eq(const double *j, double e, double *jtj6, double *jte3) {
  jtj6[0] += j[0] * j[0];
  jtj6[1] += j[0] * j[1];
  jtj6[2] += j[1] * j[1];
  jtj6[3] += j[0] * j[2];
  jtj6[4] += j[1] * j[2];
  jtj6[5] += j[2] * j[2];
  jte3[0] += j[0] * e;
  jte3[1] += j[1] * e;
  jte3[2] += j[2] * e;
}

}  // namespace brdf
