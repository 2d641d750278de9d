// batch_fit.h -- "batched" regime: S independent fits, one workgroup (or one wavefront) per fit, the
// whole LM loop inside the kernel.  Replaces the serial pixel x channel loop of
// CBRDFdata::CalcBRDFEquation (brdfdata.cpp:1195-1220), each iteration of which is one
// dlevmar_bc_dif call (brdfdata.cpp:1119).
#pragma once

#include "device_common.h"

namespace brdf {

struct BatchFitArgs {
  int method, model;
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

int synth_enqueue(int model, unsigned long long seed, long long first, int count, int n, const double *d_truth,
                  double *d_angles, double *d_x, hipStream_t stream);

}  // namespace brdf
