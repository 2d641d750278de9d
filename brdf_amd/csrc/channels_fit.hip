// channels_fit.hip -- K fits over one set of planes (brdf_hip_fit_channels_dev): dispatcher of channels_fit_impl.h's kernels,
// with the channels run one after the other through the single-fit regimes where the shared launch does not apply.
#include "channels_fit_impl.h"

namespace brdf {

int channels_run_0(const ChannelsArgs &a, CWorkspace &ws, bool *unavailable);
int channels_run_1(const ChannelsArgs &a, CWorkspace &ws, bool *unavailable);
int channels_run_2(const ChannelsArgs &a, CWorkspace &ws, bool *unavailable);

thread_local CWorkspace g_cws;
thread_local int g_channels_shared = 0;  // 1: the last call ran as ONE shared launch

int channels_last_shared() { return g_channels_shared; }
FitStats channels_last_stats(int c) { return (c >= 0 && c < kMaxChannels) ? g_cws.stats[c] : FitStats{}; }

// BRDF_HIP_CHANNELS=0: always one fit after the other
static bool channels_enabled() {
  const char *e = getenv("BRDF_HIP_CHANNELS");
  return !(e && e[0] == '0');
}

int channels_fit_run(int method, int model, const double *d_angles, const double *d_x, long long x_stride, int n, int K, double *p,
                     const double *lb, const double *ub, const double *dscl, int itmax, const double *opts, double *info, double *covar,
                     hipStream_t stream) {
  g_channels_shared = 0;
  const char *res = getenv("BRDF_HIP_RESIDENT");
  const bool resident_ok = !(res && res[0] == '0');
  int dev = 0;
  // the shared launch: box-constrained entry points, up to three channels, a fit that fits the chip
  if (channels_enabled() && resident_ok && (method == 1 || method == 2) && K >= 2 && K <= kMaxChannels && hipGetDevice(&dev) == hipSuccess &&
      g_cws.ensure(dev) == 0 && (long long)n <= (long long)g_cws.cus * kRTile && g_cws.cus <= kRowStride) {
    CWorkspace &ws = g_cws;
    if (ws.skip > 0) {
      --ws.skip;
    } else {
      ChannelsArgs a;
      a.method = method;
      a.model = model;
      a.d_angles = d_angles;
      for (int c = 0; c < kMaxChannels; ++c) a.d_x[c] = d_x + (size_t)(c < K ? c : 0) * x_stride;
      a.n = n;
      a.K = K;
      a.p = p;
      a.lb = lb;
      a.ub = ub;
      a.dscl = dscl;
      a.itmax = itmax;
      a.opts = opts;
      a.info = info;
      a.covar = covar;
      a.stream = stream;
      bool unavailable = false;
      int r;
      switch (model) {
      case 0: r = channels_run_0(a, ws, &unavailable); break;
      case 1: r = channels_run_1(a, ws, &unavailable); break;
      default: r = channels_run_2(a, ws, &unavailable); break;
      }
      if (!unavailable) {
        ws.backoff = 0;
        g_channels_shared = 1;
        return r;
      }
      ws.backoff = std::min(1024, std::max(8, ws.backoff * 2));
      ws.skip = ws.backoff;
      if (const char *e = getenv("BRDF_HIP_RESIDENT_BACKOFF")) ws.skip = std::max(0, atoi(e));
    }
  }
  int worst = 0;
  for (int c = 0; c < K; ++c) {  // one fit after the other: the single-fit regimes (resident launch or launch chain)
    StreamFitArgs a;
    a.method = (method == 2) ? 1 : (method == 3 ? 2 : method);
    a.model = model;
    a.analytic = (method == 2 || method == 3) ? 1 : 0;
    a.d_angles = d_angles;
    a.d_x = d_x + (size_t)c * x_stride;
    a.n = n;
    a.p = p + c * kM;
    a.lb = lb;
    a.ub = ub;
    a.dscl = dscl;
    a.itmax = itmax;
    a.opts = opts;
    a.info = info ? info + c * kInfoSz : nullptr;
    a.covar = covar ? covar + c * kM * kM : nullptr;
    a.stream = stream;
    const int r = stream_fit_run(a);
    g_cws.stats[c < kMaxChannels ? c : 0] = stream_fit_last_stats();
    if (r < 0) worst = kLmError;
  }
  return worst;
}

}  // namespace brdf
