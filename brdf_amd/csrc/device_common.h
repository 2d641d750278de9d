// device_common.h -- shared device-side pieces of the BRDF fitter kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include "brdf_models.h"

namespace brdf {

constexpr int kWave = 64;                       // CDNA wavefront
constexpr int kSums = SumLayout<kM>::MAX;        // 13 reduced sums at most
constexpr int kSlots = kSums + 1;                // + one max-reduced slot (max |e|)
constexpr int kNL = SumLayout<kM>::NL;           // 6 lower-triangle entries

// Workgroup reduction of NS sums and one max, as a fixed three-stage tree through LDS (no cross-lane
// shuffles: a chain of ds_bpermute-based double shuffles costs ~5 us per call on gfx950, this ~0.4 us):
//   stage 1  every thread stores its NS+1 values            buf[slot][thread]
//   stage 2  (NS+1)*32 workers each fold THREADS/32 values   stage[slot][32]   (stride-32 walk, in order)
//   stage 3  NS+1 workers each fold 32 values                out[slot]
// The order is a pure function of (THREADS, NS): results are reproducible run to run.
// out[0..NS) receive the sums, out[kSums] the max; visible to all threads on return.
constexpr int kRedFan = 32;
constexpr int kRedStride = kRedFan + 1;  // pad: stage-3 workers walk different banks

template <int THREADS>
constexpr int reduce_buf_doubles() { return kSlots * THREADS + kSlots * kRedStride; }

__device__ __forceinline__ void reduce_stage3(int ns, const double *stage, double *out) {
  const int t = threadIdx.x;
  if (t <= ns) {
    const double *src = stage + t * kRedStride;
    double s = src[0];
    if (t < ns) {
      for (int l = 1; l < kRedFan; ++l) s += src[l];
      out[t] = s;
    } else {
      for (int l = 1; l < kRedFan; ++l) s = fmax(s, src[l]);
      out[kSums] = s;
    }
  }
  __syncthreads();
}

template <int NS, int THREADS>
__device__ __forceinline__ void block_reduce(const double *acc, double mx, double *buf, double *out) {
  static_assert((NS + 1) * kRedFan <= THREADS || THREADS >= kRedFan, "worker layout");
  const int t = threadIdx.x;
  double *stage = buf + kSlots * THREADS;
#pragma unroll
  for (int k = 0; k < NS; ++k) buf[k * THREADS + t] = acc[k];
  buf[NS * THREADS + t] = mx;
  __syncthreads();
  constexpr int PER = THREADS / kRedFan;
  for (int item = t; item < (NS + 1) * kRedFan; item += THREADS) {
    const int k = item / kRedFan, l = item % kRedFan;
    const double *src = buf + k * THREADS + l;
    double s = src[0];
    if (k < NS) {
#pragma unroll
      for (int j = 1; j < PER; ++j) s += src[j * kRedFan];
    } else {
#pragma unroll
      for (int j = 1; j < PER; ++j) s = fmax(s, src[j * kRedFan]);
    }
    stage[k * kRedStride + l] = s;
  }
  __syncthreads();
  reduce_stage3(NS, stage, out);
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
  case RQ_DIF_JAC: return SumLayout<kM>::DIF_JAC;
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
