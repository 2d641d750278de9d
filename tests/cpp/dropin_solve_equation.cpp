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

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const int model = atoi(argv[1]), n = atoi(argv[2]);
  std::vector<double> buf(4 * (size_t)n);
  FILE *f = fopen(argv[3], "rb");
  if (!f || fread(buf.data(), sizeof(double), buf.size(), f) != buf.size()) return 3;
  fclose(f);

  brdf_hip_register_model(BRDFFunc);  // <- the one added line

  double p[3] = {0.5, 1.0, 1.0};
  double *x = buf.data() + 3 * (size_t)n;
  extraData *data = new extraData();
  data->angles = buf.data();
  data->modelInfo = model;

  int m = 3;  // parameters
  int itmax = 100;
  double opts[LM_OPTS_SZ];
  double info[LM_INFO_SZ];
  double lower[] = {0, 0, 0};
  double upper[] = {100, 100, 100};
  opts[0] = LM_INIT_MU; opts[1] = 1E-15; opts[2] = 1E-15; opts[3] = 1E-20;
  opts[4] = LM_DIFF_DELTA;

  int error = dlevmar_bc_dif(BRDFFunc, p, x, m, n, lower, upper, NULL, itmax, opts, info, NULL, NULL, data);
  if (error == -1) printf("Error in SolveEquation(..)\n");
  printf("RESULT %d %ld", error, g_calls);
  for (int i = 0; i < 3; ++i) printf(" %a", p[i]);
  for (int i = 0; i < LM_INFO_SZ; ++i) printf(" %a", info[i]);
  printf("\n");
  delete data;
  return 0;
}
