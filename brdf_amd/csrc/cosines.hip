// cosines.hip -- vectors -> cosines: the step immediately before the fit.
//
// Replaces CBRDFdata::GetCosLN / GetCosNH / GetCosRV (brdfdata.cpp:859-899, :902-943, :799-857), which the reference
// calls once per pixel (CalcBRDFEquation, brdfdata.cpp:1203-1205) or once per face (:1150-1160), each time rebuilding
// the face centroid for every one of the 16 lights.  Here ONE launch fills the three cosine planes of S surfels in the
// batched fitter's own layout angles[S][3][L] (the SoA planes of struct extraData, brdfdata.cpp:962-966, per surfel):
// one lane per (surfel, light), the 16 lanes of a DPP row share a surfel (their vertex gathers hit the same lines),
// consecutive lanes write consecutive doubles (128 B per surfel and plane).
//
// Arithmetic: the reference's own operations in the reference's order (Eigen's normalize() divides every component by
// sqrt(squared norm); a 3-term cwiseProduct().sum() is ((a0 b0 + a1 b1) + a2 b2)), compiled with -ffp-contract=off:
// the planes are bit-identical to the C restatement in oracle/cosines_oracle.c.
//   rv_mode 0  GetCosRV exactly as written, including its two slips (brdfdata.cpp:835: the light vector is built from
//              the centroid's x for all three components; :849: the result is R.P, not R.V) -- SURVEY.md appendix A
//   rv_mode 1  the geometry the comments describe: R = reflect(-L, N), cos = R . V
// HBM-bound: per surfel 4 B (index) + 12 B (face) + 72 B (3 vertices, gathered) + 24 B (normal) in, 3 * L * 8 B out
// (L = 16: 384 B), i.e. 496 B of algorithmic traffic per surfel.
#include <cstdio>

#include "stream_fit.h"

namespace brdf {

constexpr int kMaxLights = 64;

struct CosArgs {
  const double *vertices;  // [nv][3]
  const int *faces;        // [nf][3]
  const double *normals;   // [nf][3]
  const int *surfels;      // [S] face index per surfel, or nullptr: surfel s = face s
  double *angles;          // [S][3][L]
  long long S;
  int L, rv_mode;
  double view[3];
  double leds[kMaxLights][3];
};

__device__ __forceinline__ double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void normalize3(double *v) {  // Eigen: if (squaredNorm() > 0) v /= sqrt(squaredNorm())
  const double z = dot3(v, v);
  if (z > 0.0) {
    const double nrm = sqrt(z);
    v[0] /= nrm;
    v[1] /= nrm;
    v[2] /= nrm;
  }
}

__global__ __launch_bounds__(256) void cosines_kernel(CosArgs a) {
  __shared__ double led[kMaxLights][3];
  for (int t = threadIdx.x; t < 3 * a.L; t += blockDim.x) led[t / 3][t % 3] = a.leds[t / 3][t % 3];
  __syncthreads();
  const long long total = a.S * a.L;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long s = g / a.L;
    const int i = (int)(g - s * a.L);
    const int f = a.surfels ? a.surfels[s] : (int)s;
    const int v0 = a.faces[3 * (size_t)f], v1 = a.faces[3 * (size_t)f + 1], v2 = a.faces[3 * (size_t)f + 2];
    double c[3];  // centre of the triangle: x = 0; x += each vertex; x /= 3.0   (brdfdata.cpp:816-827)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double t = 0.0;
      t += a.vertices[3 * (size_t)v0 + k];
      t += a.vertices[3 * (size_t)v1 + k];
      t += a.vertices[3 * (size_t)v2 + k];
      c[k] = t / 3.0;
    }
    const double nrm[3] = {a.normals[3 * (size_t)f], a.normals[3 * (size_t)f + 1], a.normals[3 * (size_t)f + 2]};
    const double l[3] = {led[i][0], led[i][1], led[i][2]};

    // cos(L.N), brdfdata.cpp:886-895
    double ld[3] = {l[0] - c[0], l[1] - c[1], l[2] - c[2]};
    normalize3(ld);
    const double cos_ln = dot3(ld, nrm);

    // cos(N.H), brdfdata.cpp:930-939: H = (led - c) + (view - c), written as led - 2*c + view
    double h[3] = {l[0] - 2 * c[0] + a.view[0], l[1] - 2 * c[1] + a.view[1], l[2] - 2 * c[2] + a.view[2]};
    normalize3(h);
    const double cos_nh = dot3(h, nrm);

    // cos(R.V), brdfdata.cpp:828-853
    double cos_rv;
    if (a.rv_mode == 0) {
      double md[3] = {c[0] - l[0], c[0] - l[1], c[0] - l[2]};  // (sic) "x - m_led(i,1)", "x - m_led(i,2)"
      normalize3(md);
      const double sf = dot3(nrm, md);
      const double P[3] = {sf * nrm[0], sf * nrm[1], sf * nrm[2]};
      const double R[3] = {md[0] - 2 * P[0], md[1] - 2 * P[1], md[2] - 2 * P[2]};
      cos_rv = dot3(R, P);  // (sic) R.P
    } else {
      double vd[3] = {a.view[0] - c[0], a.view[1] - c[1], a.view[2] - c[2]};
      normalize3(vd);
      double md[3] = {c[0] - l[0], c[1] - l[1], c[2] - l[2]};  // -1 * light vector
      normalize3(md);
      const double sf = dot3(nrm, md);
      const double P[3] = {sf * nrm[0], sf * nrm[1], sf * nrm[2]};
      const double R[3] = {md[0] - 2 * P[0], md[1] - 2 * P[1], md[2] - 2 * P[2]};
      cos_rv = dot3(R, vd);
    }
    double *out = a.angles + (size_t)s * 3 * a.L + i;
    out[0] = cos_ln;
    out[a.L] = cos_nh;
    out[2 * a.L] = cos_rv;
  }
}

int cosines_run(const double *d_vertices, const int *d_faces, const double *d_normals, const int *d_surfels, long long S,
                const double *leds, int L, const double *view, int rv_mode, double *d_angles, hipStream_t stream) {
  if (!d_vertices || !d_faces || !d_normals || !leds || !view || !d_angles || S <= 0 || L <= 0 || L > kMaxLights ||
      (rv_mode != 0 && rv_mode != 1)) {
    set_error("brdf_hip_cosines_dev(): bad arguments (need device vertices/faces/normals/angles, host leds[L<=%d][3] and view[3], S > 0, rv_mode 0|1)",
              kMaxLights);
    return kLmError;
  }
  (void)hipGetLastError();
  CosArgs a;
  a.vertices = d_vertices;
  a.faces = d_faces;
  a.normals = d_normals;
  a.surfels = d_surfels;
  a.angles = d_angles;
  a.S = S;
  a.L = L;
  a.rv_mode = rv_mode;
  for (int k = 0; k < 3; ++k) a.view[k] = view[k];
  for (int i = 0; i < kMaxLights; ++i)
    for (int k = 0; k < 3; ++k) a.leds[i][k] = i < L ? leds[3 * i + k] : 0.0;
  long long blocks = (S * L + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond 64 workgroups per CU
  hipLaunchKernelGGL(cosines_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("cosines_kernel launch failed: %s", hipGetErrorString(e));
    return kLmError;
  }
  return 0;
}

// CBRDFdata::InitLEDs (brdfdata.cpp:683-752): the 16 measured LED positions of the capture rig, row-major [16][3]
void led_table(double *out) {
  const double x = 303.5, min_y = -157.1, max_y = -2.3, min_z = 555.3, max_z = 645.8;
  const double y_step = (max_y - min_y) / 3, z_step = (max_z - min_z) / 3;
  const double ys[4] = {max_y, max_y - y_step, min_y + y_step, min_y};
  const double zs[4] = {min_z, min_z + z_step, max_z - z_step, max_z};
  for (int i = 0; i < 16; ++i) {
    const int row = i / 4, col = i % 4;
    out[3 * i + 0] = x;
    out[3 * i + 1] = (row % 2 == 0) ? ys[col] : ys[3 - col];  // the rig is wired boustrophedon: rows 0, 2 run max_y -> min_y
    out[3 * i + 2] = zs[row];
  }
}

}  // namespace brdf
