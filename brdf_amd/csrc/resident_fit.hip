// resident_fit.hip -- the resident regime's dispatcher: picks the (MODEL, METHOD) instance of resident_fit_impl.h's kernels.
// The kernels themselves (and the file comment that explains the regime) live in resident_fit_impl.h; each (MODEL, METHOD)
// pair is its own translation unit (resident_inst.hip, compiled with -DRI_PAIR=<model><method>), because hipcc needs 40-60 s
// per pair and the nine of them used to be one 6.5-minute compile.
#include "resident_fit_impl.h"

namespace brdf {

#define BRDF_RESIDENT_DECLARE(NAME_)                                                         \
  int resident_run_##NAME_(const StreamFitArgs &a, RWorkspace &ws, bool *unavailable);      \
  int resident_batch_##NAME_(bool fast, const BatchCtx &c, hipStream_t stream);
#ifndef BRDF_DEV_WARD_ONLY
BRDF_RESIDENT_DECLARE(00) BRDF_RESIDENT_DECLARE(01) BRDF_RESIDENT_DECLARE(02)
BRDF_RESIDENT_DECLARE(10) BRDF_RESIDENT_DECLARE(11) BRDF_RESIDENT_DECLARE(12)
#endif
BRDF_RESIDENT_DECLARE(20) BRDF_RESIDENT_DECLARE(21) BRDF_RESIDENT_DECLARE(22)

thread_local RWorkspace g_rws;

FitStats resident_fit_last_stats() { return g_rws.stats; }
int resident_fit_last_trace(long long *out, int max_rows) {
  const int rows = std::min(max_rows, kRowStride + 1);
  memcpy(out, g_rws.h_trace, sizeof(long long) * 8 * (size_t)rows);
  return rows;
}

int resident_batch_enqueue(int model, int method, bool fast, const BatchCtx &c, hipStream_t stream) {
  switch (model * 3 + method) {
#ifndef BRDF_DEV_WARD_ONLY
  case 0: return resident_batch_00(fast, c, stream);
  case 1: return resident_batch_01(fast, c, stream);
  case 2: return resident_batch_02(fast, c, stream);
  case 3: return resident_batch_10(fast, c, stream);
  case 4: return resident_batch_11(fast, c, stream);
  case 5: return resident_batch_12(fast, c, stream);
#endif
  case 6: return resident_batch_20(fast, c, stream);
  case 7: return resident_batch_21(fast, c, stream);
  default: return resident_batch_22(fast, c, stream);
  }
}

// returns true if the resident path handled the fit (*ret is then the solver's return value)
bool resident_fit_try(const StreamFitArgs &a, int *ret) {
  // Default for every single fit that fits the chip.  Measured on MI355X, 1M-sample Ward fit: 14.3 us per dlevmar_dif
  // pass against 19.7 us for the launch chain (the secant Jacobian no longer travels through HBM) and 10.7 us per
  // dlevmar_bc_dif pass against 11.3 us.  BRDF_HIP_RESIDENT=0: always the launch chain.
  const char *e = getenv("BRDF_HIP_RESIDENT");
  if (e && e[0] == '0') return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  RWorkspace &ws = g_rws;
  if (ws.ensure(dev) != 0) return false;
  if ((long long)a.n > (long long)ws.cus * kRTile || ws.cus > kRowStride) return false;  // does not fit the chip: launch chain
  if (ws.skip > 0) {  // stepping aside after an aborted launch
    --ws.skip;
    return false;
  }
  bool unavailable = false;
  int r;
  switch (a.model * 3 + a.method) {
#ifndef BRDF_DEV_WARD_ONLY
  case 0: r = resident_run_00(a, ws, &unavailable); break;
  case 1: r = resident_run_01(a, ws, &unavailable); break;
  case 2: r = resident_run_02(a, ws, &unavailable); break;
  case 3: r = resident_run_10(a, ws, &unavailable); break;
  case 4: r = resident_run_11(a, ws, &unavailable); break;
  case 5: r = resident_run_12(a, ws, &unavailable); break;
#endif
  case 6: r = resident_run_20(a, ws, &unavailable); break;
  case 7: r = resident_run_21(a, ws, &unavailable); break;
  default: r = resident_run_22(a, ws, &unavailable); break;
  }
  if (unavailable) {
    static bool warned = false;
    if (!warned) fprintf(stderr, "libbrdf_hip: resident single-launch path unavailable (grid not co-resident?); using the launch chain\n");
    warned = true;
    ws.backoff = std::min(1024, std::max(8, ws.backoff * 2));
    ws.skip = ws.backoff;
    if (const char *e = getenv("BRDF_HIP_RESIDENT_BACKOFF")) ws.skip = std::max(0, atoi(e));  // tests
    return false;
  }
  ws.backoff = 0;
  *ret = r;
  return true;
}

}  // namespace brdf
