// capture_fit.hip -- the pixel loop of CBRDFdata::CalcBRDFEquation (brdfdata.cpp:1188-1227) as a handful of launches
// (SURVEY.md section 8, row f2).
//
// The reference walks the image x-major (x outer, y inner), and for every pixel whose pixel-map entry names a face:
// rebuilds the three cosine planes of that face (:1203-1205), and for each colour channel (B, G, R) gathers the 16
// intensities of the pixel (GetIntensities_FromPixel, :945-960: row height-1-y, value / 255.0), fits them with
// dlevmar_bc_dif (SolveEquation, :1077-1136) and stores {kd, ks, n} in brdf_surfaces(face, channel)
// (SaveValuesToSurface, :368-377) -- a later pixel of the same face overwrites an earlier one.  Here:
//
//   compact   the pixels that carry a face, in the reference's own x-major order (deterministic two-pass compaction),
//             and the LAST such pixel of every face (atomicMax of the compacted index)
//   gather    x[3S][L] = image_i(H-1-y, x)[c] / 255.0 for the 3 S (pixel, channel) fits
//   cosines   angles[3S][3][L] through cosines.hip (a pixel's three channels share a face)
//   fit       ONE batched dlevmar_bc_dif launch over the 3 S fits (batch_fit.hip)
//   store     brdf_surfaces[face][channel] <- the fit of the face's last pixel; sums of kd, ks, n over all fits
//
// Everything stays in HBM between the steps; the host sees S and the three averages.
#include <cstdio>
#include <vector>

#include "batch_fit.h"
#include "stream_fit.h"

namespace brdf {

namespace {

constexpr int kCT = 256;

// x-major index g = x * H + y  <->  pixel_map[y][x]                                   (brdfdata.cpp:1195-1197)
__device__ __forceinline__ int face_of(const int *pm, int H, int W, long long g) {
  const int x = (int)(g / H), y = (int)(g % H);
  return pm[(size_t)y * W + x];
}

__global__ __launch_bounds__(kCT) void count_kernel(const int *pm, int H, int W, int nf, int *block_count) {
  __shared__ int wave_cnt[kCT / 64];
  const long long g = (long long)blockIdx.x * kCT + threadIdx.x;
  const int f = (g < (long long)H * W) ? face_of(pm, H, W, g) : -1;
  const bool valid = f > -1 && f < nf;
  const unsigned long long m = __ballot(valid);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(kCT) void compact_kernel(const int *pm, int H, int W, int nf, const long long *block_offset,
                                                      long long *pixel_of, int *face_of_surfel, long long *last_of_face) {
  __shared__ int wave_cnt[kCT / 64];
  const long long g = (long long)blockIdx.x * kCT + threadIdx.x;
  const int f = (g < (long long)H * W) ? face_of(pm, H, W, g) : -1;
  const bool valid = f > -1 && f < nf;
  const unsigned long long m = __ballot(valid);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(m);
  __syncthreads();
  if (!valid) return;
  long long s = block_offset[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) s += wave_cnt[w];
  pixel_of[s] = g;
  face_of_surfel[s] = f;
  atomicMax(reinterpret_cast<unsigned long long *>(last_of_face) + f, (unsigned long long)(s + 1));  // 0 = no pixel
}

// fit id q = 3 s + c: the three channels of pixel s are consecutive, as in the reference's loop nest
__global__ __launch_bounds__(kCT) void gather_kernel(const unsigned char *images, int L, int H, int W, const long long *pixel_of,
                                                     const int *face_of_surfel, long long S, const double *p0, double *x,
                                                     int *faces3, double *p) {
  const long long total = 3 * S * L;
  for (long long t = (long long)blockIdx.x * kCT + threadIdx.x; t < total; t += (long long)gridDim.x * kCT) {
    const long long q = t / L;
    const int i = (int)(t - q * L);
    const long long s = q / 3;
    const int c = (int)(q - 3 * s);
    const long long g = pixel_of[s];
    const int px = (int)(g / H), py = (int)(g % H);
    const unsigned char v = images[(((size_t)i * H + (size_t)(H - 1 - py)) * W + px) * 3 + c];
    x[q * L + i] = v / 255.0;  // "intensity.val[colorChannel]/255.0", brdfdata.cpp:956
    if (i == 0) {
      faces3[q] = face_of_surfel[s];
      p[3 * q + 0] = p0[0];
      p[3 * q + 1] = p0[1];
      p[3 * q + 2] = p0[2];
    }
  }
}

// brdf_surfaces(face, channel) <- the last pixel's fit; per-block partial sums of kd, ks, n in a fixed order
__global__ __launch_bounds__(kCT) void store_kernel(const double *p, const int *face_of_surfel, const long long *last_of_face,
                                                    long long S, double *brdf_surfaces, double *block_sums) {
  __shared__ double red[3][kCT];
  const long long q = (long long)blockIdx.x * kCT + threadIdx.x;
  double v[3] = {0.0, 0.0, 0.0};
  if (q < 3 * S) {
    const long long s = q / 3;
    const int c = (int)(q - 3 * s);
    const int f = face_of_surfel[s];
    v[0] = p[3 * q];
    v[1] = p[3 * q + 1];
    v[2] = p[3 * q + 2];
    if (last_of_face[f] == s + 1) {
      double *dst = brdf_surfaces + ((size_t)f * 3 + c) * 3;
      dst[0] = v[0];
      dst[1] = v[1];
      dst[2] = v[2];
    }
  }
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int w = kCT / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x < 3) block_sums[(size_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// ---- single-BRDF variant (CalcBRDFEquation_SingleBRDF, brdfdata.cpp:1138-1186) --------------------------------
__global__ __launch_bounds__(kCT) void face_count_kernel(const long long *last_of_face, int nf, int *block_count) {
  __shared__ int wave_cnt[kCT / 64];
  const int f = blockIdx.x * kCT + threadIdx.x;
  const bool valid = f < nf && last_of_face[f] != 0;
  const unsigned long long m = __ballot(valid);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// the faces some pixel carries, in face order, each with the x-major index of its LAST pixel
__global__ __launch_bounds__(kCT) void face_compact_kernel(const long long *last_of_face, const long long *pixel_of, int nf,
                                                           const long long *block_offset, int *face_list, long long *pixel_list) {
  __shared__ int wave_cnt[kCT / 64];
  const int f = blockIdx.x * kCT + threadIdx.x;
  const bool valid = f < nf && last_of_face[f] != 0;
  const unsigned long long m = __ballot(valid);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(m);
  __syncthreads();
  if (!valid) return;
  long long r = block_offset[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) r += wave_cnt[w];
  face_list[r] = f;
  pixel_list[r] = pixel_of[last_of_face[f] - 1];
}

// one fit over all F faces x L lights: planes[3][F*L] (face-major inside a plane, "x[i*m_numImages+j] = I(i,j)",
// brdfdata.cpp:1016) from the per-face planes of cosines.hip, and the measurements of the three channels x3[3][F*L]
__global__ __launch_bounds__(kCT) void single_pack_kernel(const unsigned char *images, int L, int H, int W, const long long *pixel_list,
                                                          const double *angles_f, long long F, double *planes, double *x3) {
  const long long n = F * L;
  for (long long t = (long long)blockIdx.x * kCT + threadIdx.x; t < n; t += (long long)gridDim.x * kCT) {
    const long long r = t / L;
    const int i = (int)(t - r * L);
    const long long g = pixel_list[r];
    const int px = (int)(g / H), py = (int)(g % H);
#pragma unroll
    for (int k = 0; k < 3; ++k) planes[k * n + t] = angles_f[(r * 3 + k) * L + i];
    const unsigned char *pxl = images + (((size_t)i * H + (size_t)(H - 1 - py)) * W + px) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) x3[c * n + t] = pxl[c] / 255.0;
  }
}

struct DevBuf {
  void *ptr = nullptr;
  ~DevBuf() {
    if (ptr) (void)hipFree(ptr);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&ptr, bytes ? bytes : 1); }
  template <class T>
  T *as() const { return static_cast<T *>(ptr); }
};

#define CAP_OK(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      set_error("brdf_hip_fit_capture_dev(): %s failed: %s", #call, hipGetErrorString(e_)); \
      return kLmError;                                                                    \
    }                                                                                     \
  } while (0)

}  // namespace

int capture_fit_run(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                    const double *d_vertices, const int *d_faces, const double *d_normals, int nf, const double *leds,
                    const double *view, int rv_mode, const double *p0, const double *lb, const double *ub, int itmax,
                    const double *opts, double *d_brdf_surfaces, double *avg, long long *n_pixels, hipStream_t stream) {
  if (!d_images || !d_pixel_map || !d_vertices || !d_faces || !d_normals || !leds || !view || !p0 || !d_brdf_surfaces ||
      L <= 0 || L > 64 || H <= 0 || W <= 0 || nf <= 0) {
    set_error("brdf_hip_fit_capture_dev(): bad arguments");
    return kLmError;
  }
  (void)hipGetLastError();
  const long long npx = (long long)H * W;
  const int nb = (int)((npx + kCT - 1) / kCT);
  DevBuf counts, offsets, last;
  CAP_OK(counts.alloc(sizeof(int) * nb));
  CAP_OK(offsets.alloc(sizeof(long long) * nb));
  CAP_OK(last.alloc(sizeof(long long) * nf));
  CAP_OK(hipMemsetAsync(last.ptr, 0, sizeof(long long) * nf, stream));
  hipLaunchKernelGGL(count_kernel, dim3(nb), dim3(kCT), 0, stream, d_pixel_map, H, W, nf, counts.as<int>());
  std::vector<int> h_counts(nb);
  CAP_OK(hipMemcpyAsync(h_counts.data(), counts.ptr, sizeof(int) * nb, hipMemcpyDeviceToHost, stream));
  CAP_OK(hipStreamSynchronize(stream));
  std::vector<long long> h_off(nb);
  long long S = 0;
  for (int b = 0; b < nb; ++b) {
    h_off[b] = S;
    S += h_counts[b];
  }
  if (n_pixels) *n_pixels = S;
  if (avg) avg[0] = avg[1] = avg[2] = 0.0;
  if (S == 0) return 0;
  if (3 * S > 0x7fffffffLL) {
    set_error("brdf_hip_fit_capture_dev(): %lld pixels carry a face; the batched fitter takes at most 2^31-1 fits per call", S);
    return kLmError;
  }
  CAP_OK(hipMemcpyAsync(offsets.ptr, h_off.data(), sizeof(long long) * nb, hipMemcpyHostToDevice, stream));
  DevBuf pixel_of, face_s, faces3, x, p, angles, ret, p0d, sums;
  CAP_OK(pixel_of.alloc(sizeof(long long) * S));
  CAP_OK(face_s.alloc(sizeof(int) * S));
  CAP_OK(faces3.alloc(sizeof(int) * 3 * S));
  CAP_OK(x.alloc(sizeof(double) * 3 * S * L));
  CAP_OK(p.alloc(sizeof(double) * 9 * S));
  CAP_OK(angles.alloc(sizeof(double) * 9 * S * L));
  CAP_OK(ret.alloc(sizeof(int) * 3 * S));
  CAP_OK(p0d.alloc(sizeof(double) * 3));
  CAP_OK(hipMemcpyAsync(p0d.ptr, p0, sizeof(double) * 3, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(kCT), 0, stream, d_pixel_map, H, W, nf, offsets.as<long long>(),
                     pixel_of.as<long long>(), face_s.as<int>(), last.as<long long>());
  long long gb = (3 * S * L + kCT - 1) / kCT;
  if (gb > 256 * 64) gb = 256 * 64;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)gb), dim3(kCT), 0, stream, d_images, L, H, W, pixel_of.as<long long>(),
                     face_s.as<int>(), S, p0d.as<double>(), x.as<double>(), faces3.as<int>(), p.as<double>());
  CAP_OK(hipGetLastError());
  if (cosines_run(d_vertices, d_faces, d_normals, faces3.as<int>(), 3 * S, leds, L, view, rv_mode, angles.as<double>(), stream) != 0)
    return kLmError;
  BatchFitArgs a;
  a.method = 1;  // dlevmar_bc_dif, brdfdata.cpp:1119
  a.model = model;
  a.d_angles = angles.as<double>();
  a.d_x = x.as<double>();
  a.S = (int)(3 * S);
  a.n = L;
  a.d_p = p.as<double>();
  a.lb = lb;
  a.ub = ub;
  a.itmax = itmax;
  a.opts = opts;
  a.d_info = nullptr;
  a.d_ret = ret.as<int>();
  a.stream = stream;
  if (batch_fit_enqueue(a) != 0) return kLmError;
  const int sb = (int)((3 * S + kCT - 1) / kCT);
  CAP_OK(sums.alloc(sizeof(double) * 3 * sb));
  hipLaunchKernelGGL(store_kernel, dim3(sb), dim3(kCT), 0, stream, p.as<double>(), face_s.as<int>(), last.as<long long>(), S,
                     d_brdf_surfaces, sums.as<double>());
  CAP_OK(hipGetLastError());
  std::vector<double> h_sums((size_t)3 * sb);
  CAP_OK(hipMemcpyAsync(h_sums.data(), sums.ptr, sizeof(double) * 3 * sb, hipMemcpyDeviceToHost, stream));
  CAP_OK(hipStreamSynchronize(stream));
  if (avg) {
    double t[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < sb; ++b)
      for (int k = 0; k < 3; ++k) t[k] += h_sums[(size_t)3 * b + k];
    for (int k = 0; k < 3; ++k) avg[k] = t[k] / ((double)nf * 3);  // "avg_kd/(m_faces.rows()*3)", brdfdata.cpp:1224-1226
  }
  return 0;
}


// CalcBRDFEquation_SingleBRDF (brdfdata.cpp:1138-1186) + SolveEquation_SingleBRDF (:992-1062): ONE parameter triple per
// colour channel, fitted to the 16 samples of EVERY face the pixel map shows (n = 16 x faces; bunny: 402,928).  The
// reference fills phi/thetaDash/theta/I only for the faces some pixel carries and leaves the other rows of its
// matrices uninitialised, then pairs x (face-major) with planes read through a column-major linear index (:1031): both
// slips are not reproduced -- the fit sees exactly the carried faces, samples and measurements paired.
int capture_fit_single_run(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                           const double *d_vertices, const int *d_faces, const double *d_normals, int nf, const double *leds,
                           const double *view, int rv_mode, const double *p0, const double *lb, const double *ub, int itmax,
                           const double *opts, double *single_brdf, double *info, long long *n_faces_used, hipStream_t stream) {
  if (!d_images || !d_pixel_map || !d_vertices || !d_faces || !d_normals || !leds || !view || !p0 || !single_brdf || L <= 0 ||
      L > 64 || H <= 0 || W <= 0 || nf <= 0) {
    set_error("brdf_hip_fit_capture_single_dev(): bad arguments");
    return kLmError;
  }
  (void)hipGetLastError();
  const long long npx = (long long)H * W;
  const int nb = (int)((npx + kCT - 1) / kCT), fb = (nf + kCT - 1) / kCT;
  DevBuf counts, offsets, last, fcounts, foffsets;
  CAP_OK(counts.alloc(sizeof(int) * nb));
  CAP_OK(offsets.alloc(sizeof(long long) * nb));
  CAP_OK(last.alloc(sizeof(long long) * nf));
  CAP_OK(fcounts.alloc(sizeof(int) * fb));
  CAP_OK(foffsets.alloc(sizeof(long long) * fb));
  CAP_OK(hipMemsetAsync(last.ptr, 0, sizeof(long long) * nf, stream));
  hipLaunchKernelGGL(count_kernel, dim3(nb), dim3(kCT), 0, stream, d_pixel_map, H, W, nf, counts.as<int>());
  std::vector<int> h_counts(nb);
  CAP_OK(hipMemcpyAsync(h_counts.data(), counts.ptr, sizeof(int) * nb, hipMemcpyDeviceToHost, stream));
  CAP_OK(hipStreamSynchronize(stream));
  std::vector<long long> h_off(nb);
  long long S = 0;
  for (int b = 0; b < nb; ++b) {
    h_off[b] = S;
    S += h_counts[b];
  }
  if (n_faces_used) *n_faces_used = 0;
  if (S == 0) {
    set_error("brdf_hip_fit_capture_single_dev(): no pixel carries a face");
    return kLmError;
  }
  CAP_OK(hipMemcpyAsync(offsets.ptr, h_off.data(), sizeof(long long) * nb, hipMemcpyHostToDevice, stream));
  DevBuf pixel_of, face_s;
  CAP_OK(pixel_of.alloc(sizeof(long long) * S));
  CAP_OK(face_s.alloc(sizeof(int) * S));
  hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(kCT), 0, stream, d_pixel_map, H, W, nf, offsets.as<long long>(),
                     pixel_of.as<long long>(), face_s.as<int>(), last.as<long long>());
  hipLaunchKernelGGL(face_count_kernel, dim3(fb), dim3(kCT), 0, stream, last.as<long long>(), nf, fcounts.as<int>());
  std::vector<int> h_fc(fb);
  CAP_OK(hipMemcpyAsync(h_fc.data(), fcounts.ptr, sizeof(int) * fb, hipMemcpyDeviceToHost, stream));
  CAP_OK(hipStreamSynchronize(stream));
  std::vector<long long> h_fo(fb);
  long long F = 0;
  for (int b = 0; b < fb; ++b) {
    h_fo[b] = F;
    F += h_fc[b];
  }
  if (n_faces_used) *n_faces_used = F;
  const long long n = F * L;
  if (n > 0x7fffffffLL) {
    set_error("brdf_hip_fit_capture_single_dev(): %lld samples exceed one fit's index range", n);
    return kLmError;
  }
  CAP_OK(hipMemcpyAsync(foffsets.ptr, h_fo.data(), sizeof(long long) * fb, hipMemcpyHostToDevice, stream));
  DevBuf face_list, pixel_list, angles_f, planes, x3;
  CAP_OK(face_list.alloc(sizeof(int) * F));
  CAP_OK(pixel_list.alloc(sizeof(long long) * F));
  CAP_OK(angles_f.alloc(sizeof(double) * 3 * n));
  CAP_OK(planes.alloc(sizeof(double) * 3 * n));
  CAP_OK(x3.alloc(sizeof(double) * 3 * n));
  hipLaunchKernelGGL(face_compact_kernel, dim3(fb), dim3(kCT), 0, stream, last.as<long long>(), pixel_of.as<long long>(), nf,
                     foffsets.as<long long>(), face_list.as<int>(), pixel_list.as<long long>());
  CAP_OK(hipGetLastError());
  if (cosines_run(d_vertices, d_faces, d_normals, face_list.as<int>(), F, leds, L, view, rv_mode, angles_f.as<double>(), stream) != 0)
    return kLmError;
  long long gb = (n + kCT - 1) / kCT;
  if (gb > 256 * 64) gb = 256 * 64;
  hipLaunchKernelGGL(single_pack_kernel, dim3((unsigned)gb), dim3(kCT), 0, stream, d_images, L, H, W, pixel_list.as<long long>(),
                     angles_f.as<double>(), F, planes.as<double>(), x3.as<double>());
  CAP_OK(hipGetLastError());
  // "do the calculation once for each color-channel", brdfdata.cpp:1161: three dlevmar_bc_dif fits (:1058) over the SAME planes --
  // one shared resident launch where the fit fits the chip (channels_fit_impl.h), one fit after the other otherwise
  double p3[9];
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 3; ++k) p3[3 * c + k] = p0[k];
  const int worst = channels_fit_run(/*method=*/1, model, planes.as<double>(), x3.as<double>(), n, (int)n, 3, p3, lb, ub, nullptr, itmax, opts,
                                     info, nullptr, stream);
  for (int k = 0; k < 9; ++k) single_brdf[k] = p3[k];
  CAP_OK(hipStreamSynchronize(stream));
  return worst;
}

}  // namespace brdf
