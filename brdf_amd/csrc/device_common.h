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

// Workgroup reduction of NS sums and one max: DPP tree inside each wave, the last lane of every wave
// parks its value in LDS, NS+1 threads fold the per-wave values in wave order.  Two barriers.  The order
// is a pure function of (THREADS, NS): results are reproducible run to run.
// out[0..NS) receive the sums, out[kSums] the max; visible to all threads on return.
template <int THREADS>
constexpr int reduce_buf_doubles() { return kSlots * (THREADS / kWave); }

template <int NS, int THREADS>
__device__ __forceinline__ void block_reduce(const double *acc, double mx, double *buf, double *out) {
  constexpr int NW = THREADS / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double v = wave_reduce_to_last<OpSum>(acc[k]);
    if (lane == kWave - 1) buf[k * NW + wave] = v;
  }
  {
    const double v = wave_reduce_to_last<OpMax>(mx);
    if (lane == kWave - 1) buf[kSums * NW + wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    const double *src = buf + threadIdx.x * NW;
    double s = src[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += src[w];
    out[threadIdx.x] = s;
  } else if (threadIdx.x == kSums) {
    const double *src = buf + kSums * NW;
    double s = src[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s = fmax(s, src[w]);
    out[kSums] = s;
  }
  __syncthreads();
}

// Same tree over `count` (<= THREADS) per-workgroup partial rows that already sit in global memory:
// part[slot * row_stride + b], b < count.  Every thread issues its NS+1 loads up front so the fold
// pays ONE global-memory round trip (the rows were written by other CUs in the previous launch and
// miss in this CU's L1/L2), then the LDS tree of block_reduce takes over.
template <int NS, int THREADS>
__device__ __forceinline__ void fold_rows(const double *part, int row_stride, int count, double *buf, double *out) {
  const int t = threadIdx.x;
  double v[NS + 1];
#pragma unroll
  for (int k = 0; k < NS; ++k) v[k] = (t < count) ? part[(size_t)k * row_stride + t] : 0.0;
  v[NS] = (t < count) ? part[(size_t)kSums * row_stride + t] : 0.0;
  block_reduce<NS, THREADS>(v, v[NS], buf, out);
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
