// batch_fit.h -- "batched" regime: S independent fits, one workgroup (or one wavefront) per fit, the
// whole LM loop inside the kernel.  Replaces the serial pixel x channel loop of
// CBRDFdata::CalcBRDFEquation (brdfdata.cpp:1195-1220), each iteration of which is one
// dlevmar_bc_dif call (brdfdata.cpp:1119).
#pragma once

#include "device_common.h"

namespace brdf {

struct BatchFitArgs {
  int method, model;  // method: BRDF_METHOD_* of include/brdf_levmar.h (0 dif, 1 bc_dif, 2 bc_der, 3 der)
  const double *d_angles, *d_x;
  int S, n;
  double *d_p;
  const double *lb, *ub;
  int itmax;
  const double *opts;
  double *d_info;
  int *d_ret;
  hipStream_t stream;
};
int batch_fit_enqueue(const BatchFitArgs &a);

constexpr int kNeedsExact = -2;  // flag value: this fit has a cosine <= 0 and must take the exact model path

struct BatchCtx {
  const double *angles;  // [S][3][n]
  const double *x;       // [S][n]
  double *p;             // [S][3] in/out
  double *info;          // [S][10] or null
  int *ret;              // [S] or null
  int *flags;            // [S] internal: kNeedsExact marks fits handed to the exact kernel
  int S, n, itmax;
  int has_opts, has_lb, has_ub;
  int multi;  // bc_dif: projected-gradient candidates per sweep (workgroup/wave-per-fit kernels)
  int lane_quorum, lane_maxwait;  // lane_fit.hip: lanes waiting for / rounds between two heavy rounds
  int analytic;  // RQ_JAC rows from the model's analytic Jacobian (dlevmar_bc_der / dlevmar_der) instead of finite differences
  int chain;     // dlevmar_dif: trial points per sweep in a chain of rejections (eight-wave kernel; 1 = one at a time)
  int spec_jac;  // dlevmar_bc_dif: candidates evaluated by Jacobian passes (off in the batched kernels: they are bound by arithmetic)
  double opts[5], lb[kM], ub[kM];
};

// 1024 < n <= 4096 samples per fit: one workgroup per fit, control wave + seven sample waves (resident_fit.hip).
// fast = false: only the fits whose flag is kNeedsExact are fitted (exact model path)
int resident_batch_enqueue(int model, int method, bool fast, const BatchCtx &c, hipStream_t stream);


// n <= kLaneMaxN, dlevmar_bc_dif: one lane per fit (lane_fit.hip).  queue: two zeroed ints
constexpr int kLaneMaxN = 16;
int lane_fit_enqueue(int model, bool fast, const BatchCtx &c, int *queue, hipStream_t stream);

int synth_enqueue(int model, unsigned long long seed, long long first, int count, int n, const double *d_truth,
                  double *d_angles, double *d_x, hipStream_t stream);

}  // namespace brdf
