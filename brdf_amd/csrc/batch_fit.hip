// placeholder translation unit, replaced below
#include "batch_fit.h"
#include "stream_fit.h"
namespace brdf {
int batch_fit_enqueue(const BatchFitArgs &) { set_error("batched regime not built yet"); return kLmError; }
int synth_enqueue(int, unsigned long long, long long, int, int, const double *, double *, double *, hipStream_t) { set_error("synth not built yet"); return kLmError; }
}
