// channels_inst.hip -- ONE model's instance of the shared-plane multi-fit kernel (channels_fit_impl.h); compiled three times,
//   hipcc ... -DCI_MODEL=<model> -c channels_inst.hip -o channels_inst_<model>.o        (Makefile)
#ifndef CI_MODEL
#define CI_MODEL 2
#endif
#include "channels_fit_impl.h"

namespace brdf {
#define CI_INSTANCE_(M_) BRDF_CHANNELS_INSTANCE(M_, M_)
#define CI_INSTANCE(M_) CI_INSTANCE_(M_)
CI_INSTANCE(CI_MODEL)
}  // namespace brdf
