// tests/cpp/dropin_solve_equation.cpp -- the reference's call site, recompiled against libbrdf_hip.so.
//
// A C++ translation unit written the way CBRDFdata::SolveEquation is (brdfdata.cpp:1077-1136): the reference's own
// `struct extraData` and `BRDFFunc` (brdfdata.cpp:962-989, compiled from the reference's text, see below), the same p0 /
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

// The application's payload type and callback -- `struct extraData` and `BRDFFunc` -- are the REFERENCE's own text
// (brdfdata.cpp:962-989), not a copy kept here: the Makefile cuts those lines out of /root/reference into a temporary file
// and oracle/ref_brdffunc.cpp includes it (with OpenCV's CV_PI literal; REF_COUNT_POW routes the excerpt's pow() calls through
// a counter so that this test can tell whether the callback's body ever ran).  The point of the test is that exactly that
// text compiles and links UNCHANGED against include/brdf_levmar.h + libbrdf_hip.so.  Everything below is this test's own.
long g_ref_brdffunc_pow_calls = 0;
#define REF_COUNT_POW
#define REF_BRDFFUNC_NO_EXPORT
#include "../../oracle/ref_brdffunc.cpp"
#undef pow

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
  printf("RESULT %d %ld", status, g_ref_brdffunc_pow_calls);
  for (double v : fit) printf(" %a", v);
  for (double v : report) printf(" %a", v);
  printf("\n");
  return 0;
}
