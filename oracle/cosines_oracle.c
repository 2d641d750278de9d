/* cosines_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's vectors -> cosines step.
 *
 * PARITY UNPINNED: the reference computes these values with Eigen and OpenCV types (brdfdata.cpp:799-943), which
 * cannot be compiled in this image, and none of its files holds an expected value for them.  This file restates
 * the formulas operation by operation; it is checked against an independent numpy restatement
 * (tests/test_cosines.py), not against the reference itself.
 *
 *   orc_cosines    CBRDFdata::GetCosLN (brdfdata.cpp:859-899), GetCosNH (:902-943), GetCosRV (:799-857)
 *   orc_led_table  CBRDFdata::InitLEDs (brdfdata.cpp:683-752)
 */
#include <math.h>
#include <stddef.h>

static double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* Eigen's normalize(): if (squaredNorm() > 0) v /= sqrt(squaredNorm()) */
static void normalize3(double *v) {
  const double z = dot3(v, v);
  if (z > 0.0) {
    const double n = sqrt(z);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
}

/* angles[S][3][L]; surfels == NULL: surfel s = face s.  rv_mode 0: GetCosRV as written (its light vector uses the
 * centroid's x three times, :835, and it returns R.P, :849); rv_mode 1: cos(R.V) of the geometry in its comments. */
void orc_cosines(const double *vertices, const int *faces, const double *normals, const int *surfels, long long S,
                 const double *leds, int L, const double *view, int rv_mode, double *angles) {
  for (long long s = 0; s < S; ++s) {
    const int f = surfels ? surfels[s] : (int)s;
    double c[3];
    for (int k = 0; k < 3; ++k) { /* brdfdata.cpp:816-827 */
      double t = 0.0;
      for (int j = 0; j < 3; ++j) t += vertices[3 * (size_t)faces[3 * (size_t)f + j] + k];
      c[k] = t / 3.0;
    }
    const double *nrm = normals + 3 * (size_t)f;
    for (int i = 0; i < L; ++i) {
      const double *l = leds + 3 * i;
      double ld[3] = {l[0] - c[0], l[1] - c[1], l[2] - c[2]}; /* :886-895 */
      normalize3(ld);
      angles[((size_t)s * 3 + 0) * L + i] = dot3(ld, nrm);

      double h[3] = {l[0] - 2 * c[0] + view[0], l[1] - 2 * c[1] + view[1], l[2] - 2 * c[2] + view[2]}; /* :930-939 */
      normalize3(h);
      angles[((size_t)s * 3 + 1) * L + i] = dot3(h, nrm);

      double md[3], vd[3] = {view[0] - c[0], view[1] - c[1], view[2] - c[2]}; /* :828-853 */
      if (rv_mode == 0) {
        md[0] = c[0] - l[0];
        md[1] = c[0] - l[1];
        md[2] = c[0] - l[2];
      } else {
        md[0] = c[0] - l[0];
        md[1] = c[1] - l[1];
        md[2] = c[2] - l[2];
        normalize3(vd);
      }
      normalize3(md);
      const double sf = dot3(nrm, md);
      const double P[3] = {sf * nrm[0], sf * nrm[1], sf * nrm[2]};
      const double R[3] = {md[0] - 2 * P[0], md[1] - 2 * P[1], md[2] - 2 * P[2]};
      angles[((size_t)s * 3 + 2) * L + i] = (rv_mode == 0) ? dot3(R, P) : dot3(R, vd);
    }
  }
}

void orc_led_table(double *out) { /* brdfdata.cpp:695-752, written out entry by entry like the reference */
  const double x = 303.5, min_y = -157.1, max_y = -2.3, min_z = 555.3, max_z = 645.8;
  const double y_step = (max_y - min_y) / 3, z_step = (max_z - min_z) / 3;
  const double yz[16][2] = {
      {max_y, min_z},          {max_y - y_step, min_z},          {min_y + y_step, min_z},          {min_y, min_z},
      {min_y, min_z + z_step}, {min_y + y_step, min_z + z_step}, {max_y - y_step, min_z + z_step}, {max_y, min_z + z_step},
      {max_y, max_z - z_step}, {max_y - y_step, max_z - z_step}, {min_y + y_step, max_z - z_step}, {min_y, max_z - z_step},
      {min_y, max_z},          {min_y + y_step, max_z},          {max_y - y_step, max_z},          {max_y, max_z}};
  for (int i = 0; i < 16; ++i) {
    out[3 * i] = x;
    out[3 * i + 1] = yz[i][0];
    out[3 * i + 2] = yz[i][1];
  }
}

/* The pixel loop of CBRDFdata::CalcBRDFEquation (brdfdata.cpp:1188-1227) with GetIntensities_FromPixel (:945-960),
 * SolveEquation (:1077-1136) and SaveValuesToSurface (:368-377), restated on plain arrays.  PARITY UNPINNED (OpenCV /
 * Eigen types, no fixtures); the reference's slips in SolveEquation's plane packing (:1102) are not reproduced: the
 * fit reads the three planes orc_cosines produces.  images[L][H][W][3] (8-bit BGR), pixel_map[H][W].
 * brdf_surfaces[nf][3][3]; avg[3]; returns the number of pixels that carried a face. */
#include "oracle.h"

long long orc_fit_capture(int model, const unsigned char *images, int L, int H, int W, const int *pixel_map,
                          const double *vertices, const int *faces, const double *normals, int nf, const double *leds,
                          const double *view, int rv_mode, const double *p0, const double *lb, const double *ub, int itmax,
                          const double *opts, double *brdf_surfaces, double *avg) {
  long long count = 0;
  double sum[3] = {0.0, 0.0, 0.0};
  double angles[3 * 64], I[64];
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      const int f = pixel_map[(size_t)y * W + x];
      if (!(f > -1) || f >= nf) continue;
      ++count;
      orc_cosines(vertices, faces, normals, &f, 1, leds, L, view, rv_mode, angles);
      for (int c = 0; c < 3; ++c) {
        for (int i = 0; i < L; ++i) I[i] = images[(((size_t)i * H + (size_t)(H - 1 - y)) * W + x) * 3 + c] / 255.0;
        double p[3] = {p0[0], p0[1], p0[2]}, info[10];
        (void)orc_brdf_fit(1, model, angles, I, L, p, itmax, (double *)opts, (double *)lb, (double *)ub, info);
        for (int k = 0; k < 3; ++k) {
          brdf_surfaces[((size_t)f * 3 + c) * 3 + k] = p[k];
          sum[k] += p[k];
        }
      }
    }
  for (int k = 0; k < 3; ++k) avg[k] = sum[k] / ((double)nf * 3);
  return count;
}

/* CalcBRDFEquation_SingleBRDF + SolveEquation_SingleBRDF (brdfdata.cpp:1138-1186, :992-1062) on plain arrays, with the
 * two deviations of the product (include/brdf_levmar.h): only the faces some pixel carries enter the fit, and every
 * sample is paired with its own measurement.  single_brdf[3][3], info[3][10] (may be NULL); returns the number of faces
 * used, or -1 if a fit failed.  PARITY UNPINNED, like the rest of this file.  `work` = scratch supplied by the caller:
 * (3 + 3) * L * nf doubles + nf long longs. */
long long orc_fit_capture_single(int model, const unsigned char *images, int L, int H, int W, const int *pixel_map,
                                 const double *vertices, const int *faces, const double *normals, int nf, const double *leds,
                                 const double *view, int rv_mode, const double *p0, const double *lb, const double *ub, int itmax,
                                 const double *opts, double *single_brdf, double *info, double *work, long long *last_pixel) {
  long long F = 0;
  int bad = 0;
  for (int f = 0; f < nf; ++f) last_pixel[f] = -1;
  for (int x = 0; x < W; ++x) /* the walk of :1147-1159 / :1167-1177: a later pixel of a face overwrites an earlier one */
    for (int y = 0; y < H; ++y) {
      const int f = pixel_map[(size_t)y * W + x];
      if (f > -1 && f < nf) last_pixel[f] = (long long)x * H + y;
    }
  for (int f = 0; f < nf; ++f)
    if (last_pixel[f] >= 0) ++F;
  const long long n = F * L;
  double *planes = work, *x3 = work + 3 * n;
  double row[3 * 64];
  long long r = 0;
  for (int f = 0; f < nf; ++f) {
    if (last_pixel[f] < 0) continue;
    const int px = (int)(last_pixel[f] / H), py = (int)(last_pixel[f] % H);
    orc_cosines(vertices, faces, normals, &f, 1, leds, L, view, rv_mode, row);
    for (int i = 0; i < L; ++i) {
      for (int k = 0; k < 3; ++k) planes[k * n + r * L + i] = row[k * L + i];
      for (int c = 0; c < 3; ++c) x3[c * n + r * L + i] = images[(((size_t)i * H + (size_t)(H - 1 - py)) * W + px) * 3 + c] / 255.0;
    }
    ++r;
  }
  for (int c = 0; c < 3; ++c) {
    double p[3] = {p0[0], p0[1], p0[2]}, inf[10];
    if (orc_brdf_fit(1, model, planes, x3 + c * n, (int)n, p, itmax, (double *)opts, (double *)lb, (double *)ub, inf) < 0) bad = 1;
    for (int k = 0; k < 3; ++k) single_brdf[3 * c + k] = p[k];
    if (info)
      for (int k = 0; k < 10; ++k) info[10 * c + k] = inf[k];
  }
  return bad ? -1 : F;
}
