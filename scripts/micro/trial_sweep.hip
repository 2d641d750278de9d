// trial_sweep.hip -- the resident dlevmar_dif trial sweep in isolation: 512 threads (two waves per SIMD), 8 register-resident samples per
// lane (c0, x, q1, q2, hx, wrk, tb), the secant Jacobian in 96 KB of LDS, nine accumulators; variants of the loop body.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../brdf_amd/csrc trial_sweep.hip -o trial_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lm_machine.h"
#include "brdf_models.h"
using namespace brdf;
constexpr int T = 512, SPT = 8, CAP = T * SPT;

__device__ __forceinline__ double div_by(double x, double d, double r) {
  const double q = x * r;
  return fma(fma(-q, d, x), r, q);
}

template <int VARIANT>
__global__ __launch_bounds__(T) void k(const double *in, double *out, long long *cyc, int iters, int nk, PassUniforms<2> u0, int pend) {
  __shared__ double jl[3 * CAP];
  const int tid = threadIdx.x;
  double c0[SPT], x[SPT], q1[SPT], q2[SPT], hx[SPT], wrk[SPT], tb[SPT];
#pragma unroll
  for (int k = 0; k < SPT; ++k) {
    const double v = in[(blockIdx.x * T + tid) * SPT + k];
    c0[k] = 0.3 + 0.5 * v; x[k] = 0.2 * v; q1[k] = 0.1 + v; q2[k] = 1.0 + v; hx[k] = 0.19 * v; wrk[k] = 0; tb[k] = 1e-3 * v;
    jl[k * T + tid] = v; jl[CAP + k * T + tid] = 0.5 * v; jl[2 * CAP + k * T + tid] = -v;
  }
  __syncthreads();
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const double dpp[3] = {1e-3, -2e-3, 5e-4};
  __shared__ PassUniforms<2> su;
  if (tid < 64) su = u0;
  __syncthreads();
  const PassUniforms<2> ucopy = u0;
  const PassUniforms<2> &u = (VARIANT >= 10) ? su : ucopy;   // variants 10+: the uniforms stay in LDS, as in the kernel
  const double rinv = 1.0 / u.dp_l2;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    auto body = [&](int k) {
      const int s = k * T + tid;
      const double h = hx[k];
      const double w = model_value_q<2, true>(u, c0[k], Prep{q1[k], q2[k]});
      double jo[3] = {jl[s], jl[CAP + s], jl[2 * CAP + s]};
      if (pend) {
        const double tp = tb[k];
#pragma unroll
        for (int j = 0; j < 3; ++j) jo[j] = fma(tp, dpp[j], jo[j]);
        jl[s] = jo[0]; jl[CAP + s] = jo[1]; jl[2 * CAP + s] = jo[2];
      }
      double t = jo[0] * u.dp[0], jn[3];
      t = fma(jo[1], u.dp[1], t);
      t = fma(jo[2], u.dp[2], t);
      t = div_by(w - h - t, u.dp_l2, rinv);
#pragma unroll
      for (int j = 0; j < 3; ++j) jn[j] = fma(t, u.dp[j], jo[j]);
      const double en = x[k] - w, eo = x[k] - h;
      wrk[k] = w;
      tb[k] = t;
      acc[0] = fma(en, en, acc[0]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[1 + j] = fma(jo[j], t, acc[1 + j]);
      acc[4] = fma(t, t, acc[4]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[5 + j] = fma(jn[j], en, acc[5 + j]);
      acc[8] = fma(t, eo, acc[8]);
    };
    if (VARIANT % 10 == 0) {  // one sample per guarded block (the kernel's form)
#pragma unroll
      for (int k = 0; k < SPT; ++k)
        if (k < nk) body(k);
    } else if (VARIANT % 10 == 1) {  // no guards: one block of eight
#pragma unroll
      for (int k = 0; k < SPT; ++k) body(k);
    } else {  // pairs
#pragma unroll
      for (int k = 0; k < SPT; k += 2)
        if (k + 1 < nk) { body(k); body(k + 1); } else if (k < nk) body(k);
    }
    __syncthreads();
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 9; ++i) s += acc[i];
  for (int k = 0; k < SPT; ++k) s += wrk[k] + tb[k];
  out[blockIdx.x * T + tid] = s;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int V>
void run(const char *name, const double *d_in, double *d_out, long long *d_cyc, int pend) {
  PassUniforms<2> u{};
  double q[3] = {0.36, 0.24, 0.16};
  u.lq = BrdfModel<2>::lin(q); u.nq = BrdfModel<2>::nl(q);
  u.dp[0] = 1e-3; u.dp[1] = -2e-3; u.dp[2] = 5e-4; u.dp_l2 = 5.25e-6;
  const int iters = 200;
  hipLaunchKernelGGL((k<V>), dim3(256), dim3(T), 0, 0, d_in, d_out, d_cyc, iters, 8, u, pend);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
  int regs = 0;
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k<V>); regs = fa.numRegs;
  printf("%-28s pend %d: %.0f ticks per sweep of 8 samples per lane (two waves per SIMD), %.0f per sample and SIMD; %d VGPRs, %zu B scratch\n", name, pend,
         (double)c / iters, (double)c / iters / 16, regs, (size_t)fa.localSizeBytes);
}
int main() {
  double *d_in, *d_out; long long *d_cyc;
  hipMalloc(&d_in, 8 * 256 * T * SPT); hipMalloc(&d_out, 8 * 256 * T); hipMalloc(&d_cyc, 8);
  std::vector<double> h(256 * T * SPT);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.1 + 0.8 * ((i * 2654435761u) % 1000) / 1000.0;
  hipMemcpy(d_in, h.data(), 8 * h.size(), hipMemcpyHostToDevice);
  for (int pend = 0; pend < 2; ++pend) {
    run<0>("guarded, one per block", d_in, d_out, d_cyc, pend);
    run<1>("unguarded, eight per block", d_in, d_out, d_cyc, pend);
    run<2>("pairs", d_in, d_out, d_cyc, pend);
    run<10>("guarded, uniforms in LDS", d_in, d_out, d_cyc, pend);
    run<11>("unguarded, uniforms in LDS", d_in, d_out, d_cyc, pend);
  }
  return 0;
}
