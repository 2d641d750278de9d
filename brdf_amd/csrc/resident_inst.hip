// resident_inst.hip -- ONE (MODEL, METHOD) instance of the resident regime's kernels (resident_fit_impl.h): compiled nine times,
//   hipcc ... -DRI_PAIR=<model><method> -c resident_inst.hip -o resident_inst_<model><method>.o        (Makefile)
// MODEL 0 Phong, 1 Blinn-Phong, 2 Ward; METHOD 0 dlevmar_dif, 1 dlevmar_bc_dif / bc_der, 2 dlevmar_der.
#ifndef RI_PAIR
#define RI_PAIR 20
#endif
#include "resident_fit_impl.h"

namespace brdf {
#define RI_CAT_(a, b) a##b
#define RI_INSTANCE_(PAIR_) BRDF_RESIDENT_INSTANCE(((1##PAIR_ - 100) / 10), ((1##PAIR_ - 100) % 10), PAIR_)
#define RI_INSTANCE(PAIR_) RI_INSTANCE_(PAIR_)
RI_INSTANCE(RI_PAIR)
}  // namespace brdf
