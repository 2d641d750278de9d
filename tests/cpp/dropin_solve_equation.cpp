// tests/cpp/dropin_solve_equation.cpp -- the reference's call site, recompiled against libbrdf_hip.so.
//
// A C++ translation unit written the way CBRDFdata::SolveEquation is (brdfdata.cpp:1077-1136): its own
// `struct extraData`, its own `BRDFFunc` with the reference's arithmetic (brdfdata.cpp:962-989), the same p0 /
// opts / bounds / itmax, `dlevmar_bc_dif(BRDFFunc, p, x, m, n, lower, upper, NULL, itmax, opts, info, NULL, NULL,
// data)` -- and ONE added line, brdf_hip_register_model(BRDFFunc).  It includes the header under the name the
// reference uses ("levmar.h" is provided by -include of include/brdf_levmar.h) and is linked with -lbrdf_hip
// instead of levmar/liblevmar.a.  Prints p and info so the test can compare with the oracle.
//
// usage: dropin_solve_equation <model> <n> <samples.bin>   (samples.bin = angles[3n] then x[n], raw doubles)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "brdf_levmar.h"

#define CV_PI 3.1415926535897932384626433832795

// ---- EXCERPT of the reference, kept verbatim ON PURPOSE (brdfdata.cpp:962-989: struct extraData and BRDFFunc; only the
// call counter is added): the point of this test is that the application's own callback and payload type compile
// and link UNCHANGED against include/brdf_levmar.h + libbrdf_hip.so.  Everything below the excerpt is this test's own.
struct extraData {
  double *angles;
  int modelInfo;
};

static long g_calls = 0;

/* model to be fitted to measurements -- the application's callback */
void BRDFFunc(double *p, double x[], int m, int n, void *data) {
  extraData *incommingData = (extraData *)data;
  double *angles = incommingData->angles;
  int model = incommingData->modelInfo;
  ++g_calls;
  for (int i = 0; i < n; i++) {
    double currCosPhi = angles[i];
    if (model == 0) {
      double currCosTheta = angles[i + n * 2];
      x[i] = p[0] * currCosPhi + ((p[2] + 2.0) / 2.0 * CV_PI) * p[1] * (pow(currCosTheta, p[2]));
    } else if (model == 1) {
      double currCosThetaDash = angles[i + n];
      x[i] = p[0] * currCosPhi + p[1] * (pow(currCosThetaDash, p[2]));
    }
  }
  (void)m;
}
// ---- end of the excerpt ----------------------------------------------------------------------------------------------

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const int model = atoi(argv[1]), n = atoi(argv[2]);
  std::vector<double> samples(4 * (size_t)n);  // planes [3][n], then the n measurements
  FILE *f = fopen(argv[3], "rb");
  if (!f || fread(samples.data(), sizeof(double), samples.size(), f) != samples.size()) return 3;
  fclose(f);

  brdf_hip_register_model(BRDFFunc);  // <- the one added line

  // the call-site configuration of CBRDFdata::SolveEquation (brdfdata.cpp:1085, :1107-1119): start {0.5, 1, 1}, 100
  // iterations, box [0, 100]^3, opts {tau = LM_INIT_MU, 1e-15, 1e-15, 1e-20, LM_DIFF_DELTA}, no work / covar buffers
  extraData payload{samples.data(), model};
  double fit[3] = {0.5, 1.0, 1.0};
  double box_lo[3] = {0.0, 0.0, 0.0}, box_hi[3] = {100.0, 100.0, 100.0};
  double options[LM_OPTS_SZ] = {LM_INIT_MU, 1E-15, 1E-15, 1E-20, LM_DIFF_DELTA};
  double report[LM_INFO_SZ];
  const int status = dlevmar_bc_dif(BRDFFunc, fit, samples.data() + 3 * (size_t)n, 3, n, box_lo, box_hi, NULL, 100, options, report,
                                    NULL, NULL, &payload);
  if (status == LM_ERROR) printf("Error in SolveEquation(..)\n");
  printf("RESULT %d %ld", status, g_calls);
  for (double v : fit) printf(" %a", v);
  for (double v : report) printf(" %a", v);
  printf("\n");
  return 0;
}
