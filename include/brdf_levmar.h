/*
 * brdf_levmar.h -- C ABI of libbrdf_hip.so, the MI355X (gfx950) replacement for the Levenberg-
 * Marquardt fitting path of ccalantzis/BRDF.
 *
 * The boundary is the reference's L3->L4 edge: levmar/levmar.h as it is called from
 * brdfdata.cpp:1058 and :1119.  Relinking brdfdata.o against this library instead of
 * levmar/liblevmar.a (BRDF.pro:30) keeps every call site unchanged; see INTEGRATION.md for the one
 * registration line that lets the library recognise the application's BRDFFunc callback.
 *
 * All entry points are plain C: pointers and sizes, no C++/torch types.  Unless a name ends in
 * `_dev`, pointers are HOST pointers.  Errors never call exit(): they return LM_ERROR (-1) after a
 * message on stderr, as the reference does for its own argument errors (lm_core.c:502-505,
 * lmbc_core.c:440-461), and brdf_hip_last_error() returns the text.
 */
#ifndef BRDF_LEVMAR_H
#define BRDF_LEVMAR_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants, same values as levmar/levmar.h:68-100 -------------------------------------------- */
#define LM_DIF_WORKSZ(npar, nmeas) (4 * (nmeas) + 4 * (npar) + (nmeas) * (npar) + (npar) * (npar))
#define LM_BC_DIF_WORKSZ(npar, nmeas) (2 * (nmeas) + 4 * (npar) + (nmeas) * (npar) + (npar) * (npar))
#define LM_DER_WORKSZ(npar, nmeas) (2 * (nmeas) + 4 * (npar) + (nmeas) * (npar) + (npar) * (npar))    /* levmar.h:68 */
#define LM_BC_DER_WORKSZ(npar, nmeas) (2 * (nmeas) + 4 * (npar) + (nmeas) * (npar) + (npar) * (npar)) /* levmar.h:74 */
#define LM_VERSION "2.6 (November 2011)" /* levmar.h:101: the interface version this library stands in for */
#define LM_OPTS_SZ 5
#define LM_INFO_SZ 10
#define LM_ERROR (-1)
#define LM_INIT_MU 1E-03
#define LM_STOP_THRESH 1E-17
#define LM_DIFF_DELTA 1E-06

/* adata payload of the BRDF callback: replaces `struct extraData`, brdfdata.cpp:962-966 (same layout).
 * angles is SoA: [0,n) cos(L.N), [n,2n) cos(N.H), [2n,3n) cos(R.V) (Phong) or cos(N.V) (Ward). */
struct brdf_extra_data {
  double *angles;
  int modelInfo; /* 0 Phong, 1 Blinn-Phong (brdfdata.h:44); 2 Ward (extension, not in the reference) */
};

enum { BRDF_MODEL_PHONG = 0, BRDF_MODEL_BLINN_PHONG = 1, BRDF_MODEL_WARD = 2 };
enum { BRDF_METHOD_DIF = 0, BRDF_METHOD_BC_DIF = 1,
       BRDF_METHOD_BC_DER = 2, /* device entry points: dlevmar_bc_der with the model's analytic Jacobian (BRDFJac_hip's) */
       BRDF_METHOD_DER = 3     /* device entry points: dlevmar_der with it */ };

/* ---- drop-in solver entry points ----------------------------------------------------------------- */

/* Replaces dlevmar_dif, levmar/levmar.h:112-115 (body lm_core.c:438-842).  Same arguments, return
 * value (#iterations or LM_ERROR) and info[0..9] meaning.  `work` is accepted and ignored (the
 * scratch lives in HBM); nothing is written to it.
 *   - `func` registered with brdf_hip_register_model (or BRDFFunc_hip): the BRDF model is evaluated by the
 *     library's device code, m must be 3 (x == NULL means a zero measurement vector, as in levmar).
 *   - any other `func`, 1 <= m <= 8: `func` is called on the host (it is the caller's code), all n-sized work
 *     around it runs on the device; small problems (n*m <= 65536) are summed in the reference's order and
 *     reproduce levmar bit for bit. */
int dlevmar_dif(void (*func)(double *p, double *hx, int m, int n, void *adata), double *p, double *x,
                int m, int n, int itmax, double *opts, double *info, double *work, double *covar,
                void *adata);

/* Replaces dlevmar_bc_dif, levmar/levmar.h:124-127 (body lmbc_core.c:1062-1129 over :369-1022): the
 * call the application actually makes.  lb/ub/dscl may be NULL as in the reference; unlike the
 * reference, lb/ub are never rescaled in place when dscl is given. */
int dlevmar_bc_dif(void (*func)(double *p, double *hx, int m, int n, void *adata), double *p, double *x,
                   int m, int n, double *lb, double *ub, double *dscl, int itmax, double *opts,
                   double *info, double *work, double *covar, void *adata);

/* Replaces dlevmar_der, levmar/levmar.h:106-110 (body lm_core.c:64-432): unconstrained LM with the caller's
 * analytic Jacobian.  Host callbacks on the host, n-sized work on the device (generic path, 1 <= m <= 8). */
int dlevmar_der(void (*func)(double *p, double *hx, int m, int n, void *adata),
                void (*jacf)(double *p, double *j, int m, int n, void *adata), double *p, double *x, int m, int n,
                int itmax, double *opts, double *info, double *work, double *covar, void *adata);

/* Replaces dlevmar_bc_der, levmar/levmar.h:118-122 (body lmbc_core.c:369-1022): box-constrained LM with the
 * caller's analytic Jacobian jacf (row-major n x m, jac[i*m+j]).  func and jacf are host callbacks and are
 * called on the host; the residuals, J^T J, J^T e run on the device (generic path, 1 <= m <= 8). */
int dlevmar_bc_der(void (*func)(double *p, double *hx, int m, int n, void *adata),
                   void (*jacf)(double *p, double *j, int m, int n, void *adata), double *p, double *x, int m, int n,
                   double *lb, double *ub, double *dscl, int itmax, double *opts, double *info, double *work,
                   double *covar, void *adata);

/* levmar/levmar.h:357-361 (misc_core.c:598-611): standard deviation / Pearson correlation of the fitted
 * parameters from the m x m covariance returned through `covar`. */
double dlevmar_stddev(double *covar, int m, int i);
double dlevmar_corcoef(double *covar, int m, int i, int j);

/* levmar/levmar.h:376 (misc_core.c:616-658): coefficient of determination R^2 = 1 - SS_err / SS_tot of the model at p.
 * func is the caller's host callback (evaluated once, on the host); the three n-sized sums run on the device, in the
 * reference's descending order for n <= 65536.  x == NULL is read as a zero vector (the reference dereferences it). */
double dlevmar_R2(void (*func)(double *p, double *hx, int m, int n, void *adata), double *p, double *x, int m, int n,
                  void *adata);

/* levmar/levmar.h:336 (Axb_core.c:1140-1277): solves the m x m system A x = B by Crout LU with implicit row scaling and
 * partial pivoting; returns 1, or 0 if a row of A is zero.  A == NULL (the reference's "release the retained buffer"
 * call) does nothing and returns 1.  A scalar O(m^3) utility on the host for callers that link it (lmdemo-style
 * programs); the fitter itself solves its 3 x 3 systems in device registers.  Re-entrant, unlike the reference's. */
int dAx_eq_b_LU_noLapack(double *A, double *B, double *x, int m);

/* ---- single-precision twins: levmar/levmar.h:208-231 (slevmar_der / _dif / _bc_der / _bc_dif; the reference instantiates
 * them from the same *_core.c sources with LM_REAL = float, lm.c:43-63), :340 (sAx_eq_b_LU_noLapack), :364 (slevmar_chkjac),
 * :381-383 (slevmar_stddev / _corcoef / _R2).  Same semantics as the d-prefixed entry points above, all arithmetic in
 * float.  The callbacks are host code evaluated on the host (generic path, 1 <= m <= 8); the n-sized work -- residuals,
 * FD Jacobian fill, J^T J / J^T e, Broyden update -- runs on the device.  The BRDF application itself is double-only
 * (brdfdata.cpp:1058, :1119), so there is no registered-model shortcut for these. */
int slevmar_der(void (*func)(float *p, float *hx, int m, int n, void *adata), void (*jacf)(float *p, float *j, int m, int n, void *adata),
                float *p, float *x, int m, int n, int itmax, float *opts, float *info, float *work, float *covar, void *adata);
int slevmar_dif(void (*func)(float *p, float *hx, int m, int n, void *adata), float *p, float *x, int m, int n, int itmax, float *opts,
                float *info, float *work, float *covar, void *adata);
int slevmar_bc_der(void (*func)(float *p, float *hx, int m, int n, void *adata), void (*jacf)(float *p, float *j, int m, int n, void *adata),
                   float *p, float *x, int m, int n, float *lb, float *ub, float *dscl, int itmax, float *opts, float *info, float *work,
                   float *covar, void *adata);
int slevmar_bc_dif(void (*func)(float *p, float *hx, int m, int n, void *adata), float *p, float *x, int m, int n, float *lb, float *ub,
                   float *dscl, int itmax, float *opts, float *info, float *work, float *covar, void *adata);
void slevmar_chkjac(void (*func)(float *p, float *hx, int m, int n, void *adata), void (*jacf)(float *p, float *j, int m, int n, void *adata),
                    float *p, int m, int n, void *adata, float *err);
float slevmar_stddev(float *covar, int m, int i);
float slevmar_corcoef(float *covar, int m, int i, int j);
float slevmar_R2(void (*func)(float *p, float *hx, int m, int n, void *adata), float *p, float *x, int m, int n, void *adata);
int sAx_eq_b_LU_noLapack(float *A, float *B, float *x, int m);

/* Declares that `func` has the semantics of the reference's BRDFFunc (brdfdata.cpp:969-989): adata
 * points to a struct laid out like brdf_extra_data and the value depends on modelInfo.  A host
 * function pointer cannot run on the GPU; registration is how the drop-in entry points know they
 * may evaluate the model with the library's own HIP device code instead.  Returns 0, or -1 if the
 * table (16 entries) is full.  Thread-safe. */
int brdf_hip_register_model(void (*func)(double *p, double *hx, int m, int n, void *adata));
int brdf_hip_unregister_model(void (*func)(double *p, double *hx, int m, int n, void *adata));

/* A BRDFFunc-compatible callback evaluated on the GPU (kernel K1 alone): hx[i] = model(p; sample i).
 * Replaces BRDFFunc, brdfdata.cpp:969-989; always treated as registered. */
void BRDFFunc_hip(double *p, double *hx, int m, int n, void *adata);

/* The analytic Jacobian of the built-in models in levmar's jacf form (n x 3, row-major), evaluated on the GPU
 * (the reference only ever differentiates BRDFFunc numerically).  Passing it as `jacf` to
 * dlevmar_bc_der together with a registered `func` keeps the whole fit on the device (RQ_JAC passes then cost one
 * transcendental per sample instead of two and count no function evaluations, lmbc_core.c:1119-1124). */
void BRDFJac_hip(double *p, double *jac, int m, int n, void *adata);

/* Replaces dlevmar_chkjac, levmar/levmar.h:361-364 (body misc_core.c:250-321): err[i] near 1 where row i of jacf's
 * Jacobian agrees with func, near 0 where it does not.  func/jacf run on the host (caller's code), the comparison of
 * the n rows runs on the device. */
void dlevmar_chkjac(void (*func)(double *p, double *hx, int m, int n, void *adata),
                    void (*jacf)(double *p, double *j, int m, int n, void *adata), double *p, int m, int n, void *adata,
                    double *err);

/* ---- device-resident and batched entry points (extensions; the reference has no batched call) ----- */

/* One fit over samples already resident in HBM.  d_angles: 3*n doubles (planes as above), d_x: n
 * doubles, both DEVICE pointers.  p (in/out, 3), lb/ub/dscl (3 or NULL), opts (5 or NULL), info (10 or
 * NULL), covar (9 or NULL) are HOST pointers.  stream: a hipStream_t (NULL = default stream).
 * Returns as dlevmar_dif / dlevmar_bc_dif do. */
int brdf_hip_fit_dev(int method, int model, const double *d_angles, const double *d_x, int n, double *p,
                     const double *lb, const double *ub, const double *dscl, int itmax,
                     const double *opts, double *info, double *covar, void *stream);

/* K fits over ONE set of planes: what the reference's callers do with the three colour channels of a capture -- one
 * dlevmar_bc_dif call after the other over the same phi / thetaDash / theta (brdfdata.cpp:1159-1181, :1202-1219).  Channel c's
 * measurements are d_x + c * x_stride (device); p [channels][3] in/out, info [channels][10], covar [channels][9] (or NULL): HOST.
 * For BRDF_METHOD_BC_DIF / BRDF_METHOD_BC_DER with channels <= 3 and a fit that fits the chip the channels share ONE resident
 * launch (the planes are read and prepared once; while one channel's sums are exchanged and its LM step runs, the others
 * sweep): every channel's p, info and covar are bit-identical to brdf_hip_fit_dev on that channel alone.  Otherwise the
 * channels are fitted one after the other (same results).  Returns 0, or LM_ERROR if any channel's fit failed. */
int brdf_hip_fit_channels_dev(int method, int model, const double *d_angles, const double *d_x, long long x_stride, int n, int channels,
                              double *p, const double *lb, const double *ub, const double *dscl, int itmax, const double *opts,
                              double *info, double *covar, void *stream);
/* counters of channel `channel` of the most recent brdf_hip_fit_channels_dev on this thread; *shared_launch = 1 if it was one launch */
int brdf_hip_last_channels_stats(int channel, int *shared_launch, long long *passes, long long *jac_passes, double *device_us);
/* diagnostic builds (-DBRDF_STAMPS) only: shader cycles per section of that channel's control wave (1 waiting for the sweeping
 * waves, 2 reduction stage 2, 3 exchange, 4 LM step, 5 uniforms), summed over its passes */
int brdf_hip_last_channels_stamps(int channel, long long *out8);

/* S independent fits of n samples each -- the loop of CBRDFdata::CalcBRDFEquation
 * (brdfdata.cpp:1195-1220) as one call.  All array arguments are DEVICE pointers:
 *   d_angles[S][3][n], d_x[S][n], d_p[S][3] (in: starting points, out: fitted), d_info[S][10] (or NULL),
 *   d_ret[S] ints (or NULL; per-fit return value).
 * lb/ub/opts are HOST pointers shared by all fits.  method: any BRDF_METHOD_*.  n <= 4096: one launch (or two), a fit
 * per lane / wavefront / workgroup by size, asynchronous on `stream`.  n > 4096: the fits run one after the other, each
 * spread over the whole chip (the single-fit regimes of brdf_hip_fit_dev), and the call returns when they are done.
 * Returns 0 on success, LM_ERROR on bad arguments / launch failure. */
int brdf_hip_fit_batch_dev(int method, int model, const double *d_angles, const double *d_x, int S, int n,
                           double *d_p, const double *lb, const double *ub, int itmax,
                           const double *opts, double *d_info, int *d_ret, void *stream);

/* Host-pointer convenience over brdf_hip_fit_batch_dev (uploads, fits, downloads, synchronises).
 * Returns the number of fits that ended in LM_ERROR, or LM_ERROR itself on argument/HIP errors. */
int brdf_hip_fit_batch(int method, int model, const double *angles, const double *x, int S, int n,
                       double *p, const double *lb, const double *ub, int itmax, const double *opts,
                       double *info, int *ret);

/* hx[i] = model(p; sample i) for device-resident planes; d_hx DEVICE pointer, p HOST pointer. */
int brdf_hip_model_eval_dev(int model, const double *d_angles, int n, const double *p, double *d_hx,
                            void *stream);

/* Synthetic sample generator on the device (bench support; bit-identical to brdf_amd/synth.py's
 * counter stream for the planes).  Fills d_angles[count][3][n], d_x[count][n] for surfels
 * [first, first+count) with per-surfel truth d_truth[count][3] (device) . */
int brdf_hip_synth_dev(int model, unsigned long long seed, long long first, int count, int n,
                       const double *d_truth, double *d_angles, double *d_x, void *stream);

/* ---- the step before the fit: vectors -> cosines ---------------------------- */
/* Replaces CBRDFdata::GetCosLN / GetCosNH / GetCosRV (brdfdata.cpp:859-899, :902-943, :799-857), which the reference
 * evaluates per pixel / per face and per light on the host.  One launch fills, for S surfels, the three cosine planes
 * in the batched fitter's layout d_angles[S][3][L] (struct extraData's SoA planes, brdfdata.cpp:962-966, per surfel;
 * for L = 16 this is exactly what brdf_hip_fit_batch_dev reads with n = 16).
 *   d_vertices[nv][3], d_faces[nf][3] (vertex indices), d_face_normals[nf][3]: device, row-major (m_vertices, m_faces,
 *   face_normals); d_surfels[S]: face index of every surfel (the pixel map's entries, brdfdata.cpp:1197), or NULL for
 *   surfel s = face s; leds[L][3], view_origin[3]: HOST (m_led, m_p), L <= 64.
 *   rv_mode 0: GetCosRV's arithmetic exactly as written, including its slips (:835 builds the light vector from the
 *   centroid's x three times, :849 returns R.P); rv_mode 1: cos(R.V) of the geometry its comments describe.
 * Returns 0, or LM_ERROR with a message in brdf_hip_last_error(). */
int brdf_hip_cosines_dev(const double *d_vertices, const int *d_faces, const double *d_face_normals, const int *d_surfels,
                         long long S, const double *leds, int L, const double *view_origin, int rv_mode, double *d_angles,
                         void *stream);
/* CBRDFdata::InitLEDs (brdfdata.cpp:683-752): the capture rig's 16 LED positions, row-major [16][3] (host) */
void brdf_hip_led_table(double *leds16x3);

/* ---- the capture loop ---------------------------------------------------------- */
/* Replaces the pixel loop of CBRDFdata::CalcBRDFEquation (brdfdata.cpp:1188-1227) with its callees
 * GetIntensities_FromPixel (:945-960), GetCos* (:799-943), SolveEquation (:1077-1136: dlevmar_bc_dif, n = L) and
 * SaveValuesToSurface (:368-377).  For every pixel (x outer, y inner, as the reference walks) whose pixel-map entry is a
 * face index: the L intensities image_i(H-1-y, x)[channel] / 255.0 of each of the three channels (B, G, R) are fitted
 * from p0 and the result is stored in d_brdf_surfaces[face][channel] = {kd, ks, n}; when several pixels carry the same
 * face the LAST one in the reference's walk wins, as in the reference.  Faces no pixel carries are left untouched.
 *   d_images[L][H][W][3]: the L captures, 8-bit BGR, row-major (cv::Mat CV_8UC3); d_pixel_map[H][W]: face index or -1
 *   (pixelMap.at<int>(y, x)); mesh, leds, view_origin, rv_mode: as brdf_hip_cosines_dev; p0/lb/ub/opts: HOST.
 *   avg (host, or NULL): sum over all fits of kd, ks, n divided by (nf * 3), the statistics of brdfdata.cpp:1224-1226;
 *   n_pixels (host, or NULL): number of pixels that carried a face (3 fits each).
 * Synchronises `stream` before returning.  Returns 0, or LM_ERROR with a message in brdf_hip_last_error(). */
int brdf_hip_fit_capture_dev(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                             const double *d_vertices, const int *d_faces, const double *d_face_normals, int nf,
                             const double *leds, const double *view_origin, int rv_mode, const double *p0, const double *lb,
                             const double *ub, int itmax, const double *opts, double *d_brdf_surfaces, double *avg,
                             long long *n_pixels, void *stream);

/* Replaces CBRDFdata::CalcBRDFEquation_SingleBRDF (brdfdata.cpp:1138-1186) with SolveEquation_SingleBRDF (:992-1062): ONE
 * {kd, ks, n} per colour channel for the whole object, fitted with dlevmar_bc_dif to the L samples of every face the pixel
 * map shows (n = L x faces; the reference's call site passes p0 = {0,0,0}, itmax = 2000, opts = {1e-3,1e-15,1e-10,1e-50,1},
 * bounds [0,100]).  A face's measurements are those of its LAST pixel in the reference's walk (as its I.row(face) = ...
 * leaves them).  Arguments as brdf_hip_fit_capture_dev; single_brdf (HOST, [3 channels B,G,R][3]) receives the fits,
 * info (HOST [3][10], or NULL) levmar's info[] per channel, n_faces_used (HOST, or NULL) the faces that entered the fit.
 * Deviations: faces no pixel shows are left out (the reference feeds their uninitialised matrix rows to the solver) and
 * samples are paired with their own measurements (the reference's linear indexing at :1031 mis-pairs them).
 * Synchronises `stream`.  Returns 0, or LM_ERROR if any channel's fit failed / on bad arguments. */
int brdf_hip_fit_capture_single_dev(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                                    const double *d_vertices, const int *d_faces, const double *d_face_normals, int nf,
                                    const double *leds, const double *view_origin, int rv_mode, const double *p0,
                                    const double *lb, const double *ub, int itmax, const double *opts, double *single_brdf,
                                    double *info, long long *n_faces_used, void *stream);

/* ---- diagnostics ----------------------------------------------------------------------------------- */
int brdf_hip_device_count(void);
const char *brdf_hip_last_error(void);
/* counters of the most recent brdf_hip_fit_dev on this thread: passes launched, Jacobian passes,
 * evaluation passes, device time in microseconds between first and last pass (HIP events). */
int brdf_hip_last_fit_stats(long long *passes, long long *jac_passes, long long *eval_passes,
                            double *device_us);

/* kernel launches the most recent brdf_hip_fit_dev enqueued (its passes + the few run-ahead launches that found
 * the fit finished and returned at once): the population a profiler averages a kernel's duration over. */
long long brdf_hip_last_fit_launches(void);
/* Launch timing (off by default).  On: a resident fit (single or shared-channel launch) is bracketed by a HIP event pair recorded
 * on the fit's own stream right in front of and right behind the kernel launch; after the call, the kernel's duration in
 * microseconds as that stream saw it -- what rocprofv3 --kernel-trace reports for the same launch, plus the two event packets.
 * -1 when timing is off or the last fit ran as a chain of launches.  bench.py's roofline uses it. */
void brdf_hip_set_launch_timing(int on);
double brdf_hip_last_fit_kernel_us(void);
double brdf_hip_last_channels_kernel_us(void);

/* only meaningful in diagnostic builds (-DBRDF_STAMPS): shader cycles spent per section of the pass
 * kernel (launch chain: load state, fold, step, uniforms, persist, sweep, reduce; resident regime: -, sweep +
 * reduce, level-1 gather, level-2 gather + fold, step + uniforms), summed over the fit's passes. */
int brdf_hip_last_fit_stamps(long long *out8);
/* diagnostic builds only: [workgroup][8] s_memrealtime stamps / counters of one LM evaluation of the last resident fit; returns rows */
int brdf_hip_last_fit_trace(long long *out, int max_rows);

#ifdef __cplusplus
}
#endif
#endif /* BRDF_LEVMAR_H */
