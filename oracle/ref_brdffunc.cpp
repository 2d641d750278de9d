// oracle/ref_brdffunc.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Gives the REFERENCE's own model callback a C symbol.  The text of `struct extraData` and `BRDFFunc`
// (/root/reference/brdfdata.cpp:962-989, self-contained apart from OpenCV's CV_PI macro: SURVEY.md section 8c) is NOT in this
// repository: oracle/Makefile's `ref` target cuts those lines out of the reference where it lies into a temporary file outside
// the tree, compiles this unit with -DREF_BRDFFUNC_TEXT="<that file>" into oracle/_ref/liblevmar_ref.so and deletes the file.
//   ref_BRDFFunc        the reference's BRDFFunc, unchanged, for modelInfo 0 (Phong) and 1 (Blinn-Phong)
// CV_PI is OpenCV's literal (opencv2/core/cvdef.h: #define CV_PI 3.1415926535897932384626433832795), the one thing the
// excerpt takes from a header that is not in this image.
#include <cmath>
#include <math.h>

#define CV_PI 3.1415926535897932384626433832795

#ifdef REF_COUNT_POW  // tests/cpp/dropin_solve_equation.cpp: how often did the application's callback really run?
extern long g_ref_brdffunc_pow_calls;
static inline double ref_counted_pow(double a, double b) {
  ++g_ref_brdffunc_pow_calls;
  return std::pow(a, b);
}
#define pow ref_counted_pow
#endif

#include REF_BRDFFUNC_TEXT  // struct extraData { double* angles; int modelInfo; };  void BRDFFunc(double*, double[], int, int, void*)

#ifndef REF_BRDFFUNC_NO_EXPORT
extern "C" void ref_BRDFFunc(double *p, double *x, int m, int n, void *data) { BRDFFunc(p, x, m, n, data); }
#endif
