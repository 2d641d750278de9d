// exp_ilp.hip -- how fast can ONE SIMD evaluate the sweeps' exp (brdf_models.h: exp_nonpos, ocml's sequence)?  Cycles per exp for
// ILP = 1, 2, 4, 8 independent chains per wave and 1 or 2 waves per SIMD (256 / 512-thread workgroups, one per CU), and the same for a
// bare dependent v_fma_f64 chain.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../brdf_amd/csrc exp_ilp.hip -o exp_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#include "brdf_models.h"
using namespace brdf;

template <int ILP, bool EXP>
__global__ void k(double *out, long long *cyc, int iters, double seed) {
  double v[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) v[i] = -seed * (1.0 + 0.01 * (threadIdx.x + 64 * i));
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (EXP)
        v[i] = -exp_nonpos(v[i]) - 0.25;
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[i] = fma(v[i], 0.999, -0.001);
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int ILP, bool EXP>
void run(int threads, double *d_out, long long *d_cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<ILP, EXP>), dim3(256), dim3(threads), 0, 0, d_out, d_cyc, iters, 0.7);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
  const double per = (double)c / iters / ILP / (EXP ? 1 : 16);
  const int waves_per_simd = threads / 256;
  printf("%s ILP %d waves/SIMD %d: %.1f cycles per %s per wave -> %.1f per SIMD-issued %s\n", EXP ? "exp" : "fma", ILP, waves_per_simd, per,
         EXP ? "exp" : "fma", per / waves_per_simd, EXP ? "exp" : "fma");
}

int main() {
  double *d_out;
  long long *d_cyc;
  hipMalloc(&d_out, sizeof(double) * 256 * 512);
  hipMalloc(&d_cyc, 8);
  for (int threads : {256, 512}) {
    run<1, false>(threads, d_out, d_cyc);
    run<2, false>(threads, d_out, d_cyc);
    run<4, false>(threads, d_out, d_cyc);
    run<8, false>(threads, d_out, d_cyc);
    run<1, true>(threads, d_out, d_cyc);
    run<2, true>(threads, d_out, d_cyc);
    run<4, true>(threads, d_out, d_cyc);
    run<8, true>(threads, d_out, d_cyc);
  }
  return 0;
}
