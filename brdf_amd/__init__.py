"""brdf_amd -- MI355X-native Levenberg-Marquardt BRDF fitter (the hot path of ccalantzis/BRDF).

The product is libbrdf_hip.so (C ABI in include/brdf_levmar.h, HIP kernels in brdf_amd/csrc).  This
package is the thin Python plumbing around it: ctypes bindings, torch tensors as device memory,
torch.distributed (RCCL) for the surfel-sharded multi-GPU driver, and the synthetic input generator.
"""
from . import synth  # noqa: F401
from ._lib import LIB_PATH, lib, last_error  # noqa: F401
from .fit import (METHOD_BC_DER, METHOD_BC_DIF, METHOD_DER, METHOD_DIF, MODEL_BLINN_PHONG, MODEL_PHONG, MODEL_WARD, FitResult, fit_batch,  # noqa: F401
                  cosines, fit_capture, fit_capture_single, fit_channels, fit_single, last_channels_stats, host_dlevmar, last_fit_stats, led_table, model_eval, chkjac, model_jacobian, set_launch_timing)
