// stream_fit.h -- "streamed" regime (launch chain): ONE large fit spread over the whole chip, one launch per pass.
// Since resident_fit.hip took over the fits that fit the chip (n <= #CUs * 4096), this is the path of larger fits
// (n up to ~1e8), of BRDF_HIP_RESIDENT=0, and the fallback when the resident launch cannot run.
//
// Every LM evaluation is one kernel launch (a *pass*) on one stream; the scalar LM state machine
// (lm_machine.h) is re-executed redundantly at the top of each launch by every workgroup from the
// previous launch's per-workgroup partial sums, so the launches form a dependency chain with no
// host round trip and no intra-launch hand-off (the kernel boundary is the only synchronisation:
// MI355X_MICROARCH "boundary" row, ~1.5 us, cheaper than any grid barrier on this chip).
#pragma once

#include "device_common.h"

namespace brdf {

constexpr int kStreamThreads = 512;
constexpr int kStreamMaxBlocks = 512;  // width of the partial-sum rows (and of the folding reduction)
constexpr int kStreamGridCap = 256;    // workgroups actually launched: one 512-thread workgroup per CU
constexpr int kStreamWavesPerSimd = 2; // 8 waves per CU: the register allocator may use up to 256 VGPRs (no spills)

union MachineUnion {
  DifMachine<kM> dif;
  BcMachine<kM> bc;
  DerMachine<kM> der;
  __host__ __device__ MachineUnion() {}
};

// pinned, host-mapped: the finishing pass writes the result here (system scope), the host polls `done`
struct Mailbox {
  int done;
  int ret;
  int passes;
  int infeasible_mask;
  long long t_first, t_last;  // s_memrealtime stamps (100 MHz) of the first and the finishing pass
  long long n_jac, n_eval;
  double p[kM];
  double info[kInfoSz];
  double covar[kM * kM];
  int progress;  // last pass index that started (run-ahead throttle)
  int domain_bad;
  long long stamps[8];  // diagnostic builds only (-DBRDF_STAMPS): cycles per kernel section, summed over passes
};

struct StreamCtx {
  const double *c0, *c1, *c2, *x;
  double *prep[2]; // per-sample invariants written by pass 0 (FAST path): q1[n], q2[n]
  double *jac[2];  // Jacobian, SoA: plane k at jac[b] + k*n     (dif only)
  double *partials;  // [2][kSlots][kStreamMaxBlocks]
  Mailbox *mbox;     // device-visible address of the pinned mailbox
  int n, nb, method, model;
  int done;  // sticky: set by the finishing pass, read by every later pass
  int domain_bad;  // FAST path only: a cosine <= 0 was met (log undefined): the host re-runs on the exact path
  long long t_first;
  long long n_jac, n_eval;
  long long stamps[8];
#ifdef BRDF_STAMPS
  int dbg[4096 * 4];  // diagnostic builds: per pass {step cycles, kind before*100+after, phase before*100+after}
#endif
  MachineUnion m[2];  // double-buffered: pass k reads m[k&1], writes m[(k+1)&1]
};

// enqueue-and-wait driver; returns the solver's return value (>=0 iterations, or kLmError)
struct StreamFitArgs {
  int method, model;
  int analytic = 0;  // method 1 only: dlevmar_bc_der with the model's analytic Jacobian instead of finite differences
  const double *d_angles, *d_x;
  int n;
  double *p;
  const double *lb, *ub, *dscl;
  int itmax;
  const double *opts;
  double *info, *covar;
  hipStream_t stream;
};
int stream_fit_run(const StreamFitArgs &a);

struct FitStats {
  long long passes, jac_passes, eval_passes;
  long long launches;  // kernel launches enqueued for the fit, run-ahead launches that found it finished included
  double device_us;
  double kernel_us;    // launch timing on (launch_timing_enabled()): HIP events recorded on the fit's stream right in front of and
                       // right behind its resident launch -- the kernel's duration as the stream sees it; otherwise -1
  long long stamps[8];
};
FitStats stream_fit_last_stats();
// brdf_hip_set_launch_timing(): the resident regimes bracket their launch with a HIP event pair on the launch stream (bench.py's
// roofline reads it; off by default: two event records and one event wait per fit)
bool launch_timing_enabled();
void set_launch_timing(bool on);
// an event pair per host thread, created on first use (resident_fit_impl.h / channels_fit_impl.h)
struct LaunchTimer {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int device = -1;  // the events belong to the device they were created on
  bool armed = false;
  void drop() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    e0 = e1 = nullptr;
  }
  ~LaunchTimer() { drop(); }
  void before(hipStream_t s) {
    armed = false;
    if (!launch_timing_enabled()) return;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return;
    if (dev != device) {
      drop();
      device = dev;
    }
    if (!e0 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) {
      drop();
      return;
    }
    armed = hipEventRecord(e0, s) == hipSuccess;
  }
  void after(hipStream_t s) {
    if (armed) armed = hipEventRecord(e1, s) == hipSuccess;
  }
  double elapsed_us() {  // after the launch is known to have finished
    float ms = 0.0f;
    if (!armed || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0;
    return 1e3 * (double)ms;
  }
};

// resident single-launch regime (resident_fit.hip): true if it handled the fit
bool resident_fit_try(const StreamFitArgs &a, int *ret);
FitStats resident_fit_last_stats();
int resident_fit_last_trace(long long *out, int max_rows);  // diagnostic builds: [workgroup][8] stamps of one epoch

// K fits over one set of planes (channels_fit.hip): ONE shared resident launch for the box-constrained entry points (K <= 3, a fit
// that fits the chip), otherwise one fit after the other.  d_x: channel c at d_x + c * x_stride; p [K][3], info [K][10], covar [K][9]: host
int channels_fit_run(int method, int model, const double *d_angles, const double *d_x, long long x_stride, int n, int K, double *p,
                     const double *lb, const double *ub, const double *dscl, int itmax, const double *opts, double *info, double *covar,
                     hipStream_t stream);
int channels_last_shared();           // 1: the last channels_fit_run on this thread was one shared launch
FitStats channels_last_stats(int c);  // channel c of that call

int capture_fit_single_run(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                           const double *d_vertices, const int *d_faces, const double *d_normals, int nf, const double *leds,
                           const double *view, int rv_mode, const double *p0, const double *lb, const double *ub, int itmax,
                           const double *opts, double *single_brdf, double *info, long long *n_faces_used, hipStream_t stream);
// vectors -> cosines (cosines.hip)
int cosines_run(const double *d_vertices, const int *d_faces, const double *d_normals, const int *d_surfels, long long S,
                const double *leds, int L, const double *view, int rv_mode, double *d_angles, hipStream_t stream);
void led_table(double *out16x3);
// the pixel loop of CalcBRDFEquation (capture_fit.hip)
int capture_fit_run(int model, const unsigned char *d_images, int L, int H, int W, const int *d_pixel_map,
                    const double *d_vertices, const int *d_faces, const double *d_normals, int nf, const double *leds,
                    const double *view, int rv_mode, const double *p0, const double *lb, const double *ub, int itmax,
                    const double *opts, double *d_brdf_surfaces, double *avg, long long *n_pixels, hipStream_t stream);

bool brdf_fast_path_enabled();  // false when BRDF_HIP_EXACT_POW=1
int pg_candidates();            // BRDF_HIP_PG_MULTI (default kMaxCand)
int dif_chain_candidates();     // BRDF_HIP_DIF_CHAIN (default kMaxCand): dlevmar_dif trial points per sweep in a chain of rejections
bool bc_spec_jac_enabled();     // BRDF_HIP_SPEC_JAC (default on): single fits evaluate dlevmar_bc_dif candidates by Jacobian passes
void set_error(const char *fmt, ...);
const char *get_error();

}  // namespace brdf
