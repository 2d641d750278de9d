// micro-benchmark: cost of a dependent chain of near-empty launches vs grid geometry (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void noop(int *flag) { if (*flag) return; }
template <int LDS> __global__ void noop_lds(int *flag) { __shared__ double buf[LDS]; if (threadIdx.x == 0) buf[0] = 1.0; __syncthreads(); if (*flag && buf[0] > 2.0) flag[1] = 1; }
int main() {
  int *flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 2000;
  struct G { int b, t; } gs[] = {{1,64},{8,256},{64,256},{128,256},{256,64},{256,128},{256,256},{256,512},{256,1024},{512,256},{512,512},{1024,256},{2048,256}};
  for (auto g : gs) {
    for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(noop, dim3(g.b), dim3(g.t), 0, s, flag);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(noop, dim3(g.b), dim3(g.t), 0, s, flag);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(noop_lds<7500>, dim3(g.b), dim3(g.t), 0, s, flag);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms2; hipEventElapsedTime(&ms2, e0, e1);
    printf("grid %5d x %4d : %.2f us/launch   with 60KB LDS+sync: %.2f us/launch\n", g.b, g.t, 1e3 * ms / reps, 1e3 * ms2 / reps);
  }
  return 0;
}
