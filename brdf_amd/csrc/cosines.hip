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
#include <cstdlib>

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
// x / d, correctly rounded, given r = RN(1/d): q = RN(x r) is within an ulp of the quotient, the residual x - q d is exact
// in an fma, and RN(q + residual * r) is the correctly rounded quotient (Markstein's theorem) -- as long as nothing on the
// way is subnormal, which the callers' range checks guarantee.  Three instructions instead of the ~12 (one of them a
// quarter-rate v_rcp_f64) of an fp64 division; the kernel divides nine to twelve times per lane, always three times by the
// same divisor.  Checked against true division on 3e8 random operand pairs on the host and, bit for bit, by the tests
// against the C restatement.
__device__ __forceinline__ double div_by_r(double x, double d, double r) {
  const double q = x * r;
  return fma(fma(-q, d, x), r, q);
}
__device__ __forceinline__ void normalize3(double *v) {  // Eigen: if (squaredNorm() > 0) v /= sqrt(squaredNorm())
  const double z = dot3(v, v);
  if (z > 0.0) {
    const double nrm = sqrt(z);
    if (z > 1e-200 && z < 1e200) {  // |v[k]| <= nrm: quotients and residuals stay normal (a zero component comes out as +0
                                    // whatever its sign: the same VALUE as the division's, and no cosine can tell)
      const double r = 1.0 / nrm;
      v[0] = div_by_r(v[0], nrm, r);
      v[1] = div_by_r(v[1], nrm, r);
      v[2] = div_by_r(v[2], nrm, r);
    } else {
      v[0] /= nrm;
      v[1] /= nrm;
      v[2] /= nrm;
    }
  }
}

// what a lane computes once it holds its surfel's centre c, normal nrm and its light l: the three cosines
__device__ __forceinline__ void cosines_of(const CosArgs &a, const double *c, const double *nrm, const double *l, double &cos_ln, double &cos_nh,
                                           double &cos_rv) {
  // cos(L.N), brdfdata.cpp:886-895
  double ld[3] = {l[0] - c[0], l[1] - c[1], l[2] - c[2]};
  normalize3(ld);
  cos_ln = dot3(ld, nrm);

  // cos(N.H), brdfdata.cpp:930-939: H = (led - c) + (view - c), written as led - 2*c + view
  double h[3] = {l[0] - 2 * c[0] + a.view[0], l[1] - 2 * c[1] + a.view[1], l[2] - 2 * c[2] + a.view[2]};
  normalize3(h);
  cos_nh = dot3(h, nrm);

  // cos(R.V), brdfdata.cpp:828-853
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
}

// lane N's value in all 16 lanes of its DPP row (row_newbcast:N)
template <int N>
__device__ __forceinline__ int row_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ double row_bcast_d(double v) {
  return __hiloint2double(row_bcast_i<N>(__double2hiint(v)), row_bcast_i<N>(__double2loint(v)));
}

// L = 16 (the capture rig, brdfdata.h:58): a DPP row of 16 lanes is one surfel's 16 lights.  The surfel's 112 bytes -- index, three
// vertex indices, nine vertex coordinates, three normal components -- are loaded ONCE per row (lane j < 3 a vertex index, then lane
// j < 9 a coordinate and lanes 9..11 the normal) and handed round by row broadcasts, instead of once per lane (sixteen lanes each
// chasing surfel -> face -> vertices: 16 load instructions per lane, three deep).  Same arithmetic per lane afterwards: same bits.
__global__ __launch_bounds__(256) void cosines_rows_kernel(CosArgs a) {
  __shared__ double led[16][3];
  if (threadIdx.x < 48) led[threadIdx.x / 3][threadIdx.x % 3] = a.leds[threadIdx.x / 3][threadIdx.x % 3];
  __syncthreads();
  const int i = (int)threadIdx.x & 15, local = (int)threadIdx.x >> 4;
  const double l[3] = {led[i][0], led[i][1], led[i][2]};
  for (long long s0 = (long long)blockIdx.x * 16; s0 < a.S; s0 += (long long)gridDim.x * 16) {
    const long long s = s0 + local;
    const bool live = s < a.S;  // (whole rows: every lane of a row agrees)
    const long long ss = live ? s : a.S - 1;
    const int f = a.surfels ? a.surfels[ss] : (int)ss;
    const int vmine = a.faces[3 * (size_t)f + (i < 3 ? i : 0)];  // lanes 0..2: the face's vertex indices
    const int v0 = row_bcast_i<0>(vmine), v1 = row_bcast_i<1>(vmine), v2 = row_bcast_i<2>(vmine);
    const int vsel = i < 3 ? v0 : (i < 6 ? v1 : v2);
    const int comp = i < 9 ? i % 3 : 0;
    const double *src = (i < 9) ? a.vertices + 3 * (size_t)vsel + comp : a.normals + 3 * (size_t)f + (i < 12 ? i - 9 : 0);
    const double mine = *src;  // lanes 0..8: vertex coordinates (v0.xyz, v1.xyz, v2.xyz); 9..11: the normal
    const double x0[3] = {row_bcast_d<0>(mine), row_bcast_d<1>(mine), row_bcast_d<2>(mine)};
    const double x1[3] = {row_bcast_d<3>(mine), row_bcast_d<4>(mine), row_bcast_d<5>(mine)};
    const double x2[3] = {row_bcast_d<6>(mine), row_bcast_d<7>(mine), row_bcast_d<8>(mine)};
    const double nrm[3] = {row_bcast_d<9>(mine), row_bcast_d<10>(mine), row_bcast_d<11>(mine)};
    double c[3];  // centre of the triangle: x = 0; x += each vertex; x /= 3.0   (brdfdata.cpp:816-827)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double t = 0.0;
      t += x0[k];
      t += x1[k];
      t += x2[k];
      const double at = fabs(t);
      c[k] = (at > 1e-290 && at < 1e300) ? div_by_r(t, 3.0, 1.0 / 3.0) : t / 3.0;  // (t == 0 takes the division)
    }
    double cos_ln, cos_nh, cos_rv;
    cosines_of(a, c, nrm, l, cos_ln, cos_nh, cos_rv);
    if (live) {
      double *out = a.angles + (size_t)s * 48 + i;
      out[0] = cos_ln;
      out[16] = cos_nh;
      out[32] = cos_rv;
    }
  }
}

__global__ __launch_bounds__(256) void cosines_kernel(CosArgs a) {
  __shared__ double led[kMaxLights][3];
  for (int t = threadIdx.x; t < 3 * a.L; t += blockDim.x) led[t / 3][t % 3] = a.leds[t / 3][t % 3];
  __syncthreads();
  // a workgroup covers 256 / L whole surfels per trip (L = 16: all 256 lanes busy), so a lane's light index never
  // changes and its surfel index advances by a constant: no division in the loop
  const int per_block = (int)blockDim.x / a.L;
  const int i = (int)threadIdx.x % a.L;
  const int local = (int)threadIdx.x / a.L;
  if (local >= per_block) return;
  for (long long s = (long long)blockIdx.x * per_block + local; s < a.S; s += (long long)gridDim.x * per_block) {
    const int f = a.surfels ? a.surfels[s] : (int)s;
    const int v0 = a.faces[3 * (size_t)f], v1 = a.faces[3 * (size_t)f + 1], v2 = a.faces[3 * (size_t)f + 2];
    double c[3];  // centre of the triangle: x = 0; x += each vertex; x /= 3.0   (brdfdata.cpp:816-827)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double t = 0.0;
      t += a.vertices[3 * (size_t)v0 + k];
      t += a.vertices[3 * (size_t)v1 + k];
      t += a.vertices[3 * (size_t)v2 + k];
      const double at = fabs(t);
      c[k] = (at > 1e-290 && at < 1e300) ? div_by_r(t, 3.0, 1.0 / 3.0) : t / 3.0;  // (t == 0 takes the division)
    }
    const double nrm[3] = {a.normals[3 * (size_t)f], a.normals[3 * (size_t)f + 1], a.normals[3 * (size_t)f + 2]};
    const double l[3] = {led[i][0], led[i][1], led[i][2]};

    double cos_ln, cos_nh, cos_rv;
    cosines_of(a, c, nrm, l, cos_ln, cos_nh, cos_rv);
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
  const long long per_block = 256 / L;  // whole surfels per workgroup and trip
  long long blocks = (S + per_block - 1) / per_block;
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond 64 workgroups per CU
  static const bool rows_off = [] { const char *e = getenv("BRDF_HIP_COSINES_ROWS"); return e && e[0] == '0'; }();
  if (L == 16 && !rows_off)
    hipLaunchKernelGGL(cosines_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  else
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
