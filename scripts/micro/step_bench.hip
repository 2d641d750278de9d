// micro-benchmark: cycles per LM state-machine step on one lane, machine resident in LDS (as in the kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../brdf_amd/csrc/lm_machine.h"
using namespace brdf;

template <int VARIANT>
__global__ void step_kernel(long long *out, int *phases, int nsteps) {
  __shared__ BcMachine<3> sm;
  __shared__ double sums[16];
  if (threadIdx.x == 0) {
    const double p0[3] = {0.5, 1.0, 1.0}, lb[3] = {0, 0, 0}, ub[3] = {100, 100, 100};
    const double opts[5] = {1e-3, 1e-15, 1e-15, 1e-20, 1e-6};
    sm.start(p0, 1000, lb, ub, nullptr, 100000, opts, 0);
    for (int i = 0; i < 16; ++i) sums[i] = 0.0;
  }
  __syncthreads();
  for (int it = 0; it < nsteps; ++it) {
#ifdef WHOLE_WAVE
    {  // variant: the whole wave executes the step redundantly behind a wave-uniform branch
#else
    if (threadIdx.x == 0) {
#endif
      const int kind = sm.h.req.kind;
      if (kind == RQ_JAC) {  // a fixed SPD system
        sums[0] = 4.0; sums[1] = 1.0; sums[2] = 3.0; sums[3] = 0.5; sums[4] = 0.2; sums[5] = 2.0;
        sums[6] = 0.3; sums[7] = -0.2; sums[8] = 0.1; sums[9] = 10.0;
      } else {
        sums[0] = (it == 0) ? 10.0 : 10.0 + 1e-3;  // never an improvement: line search, then projected gradient
      }
      if (threadIdx.x == 0) phases[it] = sm.h.phase * 1000 + kind;
      const long long t0 = clock64();
      sm.template step<true>(sums, 1.0);
      const long long dt_ = clock64() - t0;
      if (threadIdx.x == 0) out[it] = dt_;
    }
    __syncthreads();
    if (sm.h.req.kind == RQ_DONE) break;
  }
}

int main() {
  const int N = 700;
  long long *out; int *ph;
  hipMalloc(&out, N * 8); hipMalloc(&ph, N * 4);
  hipMemset(out, 0, N * 8); hipMemset(ph, 0, N * 4);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(step_kernel<0>, dim3(1), dim3(64), 0, 0, out, ph, N);
  hipDeviceSynchronize();
  static long long h[N]; static int p[N];
  hipMemcpy(h, out, N * 8, hipMemcpyDeviceToHost); hipMemcpy(p, ph, N * 4, hipMemcpyDeviceToHost);
  long long sum[64][8] = {{0}}; int cnt[64][8] = {{0}};
  for (int i = 0; i < N; ++i) if (h[i]) { sum[p[i] / 1000][p[i] % 1000] += h[i]; cnt[p[i] / 1000][p[i] % 1000]++; }
  for (int a = 0; a < 64; ++a) for (int k = 0; k < 8; ++k) if (cnt[a][k]) printf("phase %2d kind %d : n=%4d mean %6lld cycles\n", a, k, cnt[a][k], sum[a][k] / cnt[a][k]);
  return 0;
}
