"""ctypes binding of libbrdf_hip.so (include/brdf_levmar.h).

The library is the product: if it is missing or does not export the ABI, importing this module fails
loudly -- there is no Python or CPU fallback for the fitting path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BRDF_HIP_LIB lets a diagnostic build of the same library (e.g. -DBRDF_STAMPS) be profiled
LIB_PATH = os.environ.get("BRDF_HIP_LIB") or os.path.join(_HERE, "libbrdf_hip.so")

D = C.POINTER(C.c_double)
F = C.POINTER(C.c_float)
I = C.POINTER(C.c_int)
MODEL_FUNC = C.CFUNCTYPE(None, D, D, C.c_int, C.c_int, C.c_void_p)


class ExtraData(C.Structure):
    """struct brdf_extra_data == the reference's struct extraData (brdfdata.cpp:962-966)."""
    _fields_ = [("angles", D), ("modelInfo", C.c_int)]


# every symbol include/brdf_levmar.h declares: name -> (restype, argtypes)
ABI = {
    "dlevmar_dif": (C.c_int, [C.c_void_p, D, D, C.c_int, C.c_int, C.c_int, D, D, D, D, C.c_void_p]),
    "dlevmar_bc_dif": (C.c_int, [C.c_void_p, D, D, C.c_int, C.c_int, D, D, D, C.c_int, D, D, D, D, C.c_void_p]),
    "dlevmar_der": (C.c_int, [C.c_void_p, C.c_void_p, D, D, C.c_int, C.c_int, C.c_int, D, D, D, D, C.c_void_p]),
    "dlevmar_bc_der": (C.c_int, [C.c_void_p, C.c_void_p, D, D, C.c_int, C.c_int, D, D, D, C.c_int, D, D, D, D, C.c_void_p]),
    "dlevmar_stddev": (C.c_double, [D, C.c_int, C.c_int]),
    "dlevmar_corcoef": (C.c_double, [D, C.c_int, C.c_int, C.c_int]),
    "dlevmar_R2": (C.c_double, [C.c_void_p, D, D, C.c_int, C.c_int, C.c_void_p]),
    "dAx_eq_b_LU_noLapack": (C.c_int, [D, D, D, C.c_int]),
    "slevmar_dif": (C.c_int, [C.c_void_p, F, F, C.c_int, C.c_int, C.c_int, F, F, F, F, C.c_void_p]),
    "slevmar_bc_dif": (C.c_int, [C.c_void_p, F, F, C.c_int, C.c_int, F, F, F, C.c_int, F, F, F, F, C.c_void_p]),
    "slevmar_der": (C.c_int, [C.c_void_p, C.c_void_p, F, F, C.c_int, C.c_int, C.c_int, F, F, F, F, C.c_void_p]),
    "slevmar_bc_der": (C.c_int, [C.c_void_p, C.c_void_p, F, F, C.c_int, C.c_int, F, F, F, C.c_int, F, F, F, F, C.c_void_p]),
    "slevmar_chkjac": (None, [C.c_void_p, C.c_void_p, F, C.c_int, C.c_int, C.c_void_p, F]),
    "slevmar_stddev": (C.c_float, [F, C.c_int, C.c_int]),
    "slevmar_corcoef": (C.c_float, [F, C.c_int, C.c_int, C.c_int]),
    "slevmar_R2": (C.c_float, [C.c_void_p, F, F, C.c_int, C.c_int, C.c_void_p]),
    "sAx_eq_b_LU_noLapack": (C.c_int, [F, F, F, C.c_int]),
    "brdf_hip_register_model": (C.c_int, [C.c_void_p]),
    "brdf_hip_unregister_model": (C.c_int, [C.c_void_p]),
    "BRDFFunc_hip": (None, [D, D, C.c_int, C.c_int, C.c_void_p]),
    "BRDFJac_hip": (None, [D, D, C.c_int, C.c_int, C.c_void_p]),
    "dlevmar_chkjac": (None, [C.c_void_p, C.c_void_p, D, C.c_int, C.c_int, C.c_void_p, D]),
    "brdf_hip_fit_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, D, D, D, D, C.c_int, D, D, D,
                                   C.c_void_p]),
    "brdf_hip_fit_channels_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, D, D, D, D, C.c_int, D,
                                            D, D, C.c_void_p]),
    "brdf_hip_last_channels_stamps": (C.c_int, [C.c_int, C.POINTER(C.c_longlong)]),
    "brdf_hip_last_channels_stats": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), D]),
    "brdf_hip_fit_batch_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, D, D,
                                         C.c_int, D, C.c_void_p, C.c_void_p, C.c_void_p]),
    "brdf_hip_fit_batch": (C.c_int, [C.c_int, C.c_int, D, D, C.c_int, C.c_int, D, D, D, C.c_int, D, D, I]),
    "brdf_hip_model_eval_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, D, C.c_void_p, C.c_void_p]),
    "brdf_hip_synth_dev": (C.c_int, [C.c_int, C.c_ulonglong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "brdf_hip_cosines_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, D, C.c_int, D, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "brdf_hip_led_table": (None, [D]),
    "brdf_hip_fit_capture_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, D, D, C.c_int, D, D, D, C.c_int, D, C.c_void_p, D,
                                           C.POINTER(C.c_longlong), C.c_void_p]),
    "brdf_hip_fit_capture_single_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_int, D, D, C.c_int, D, D, D, C.c_int, D, D, D,
                                                  C.POINTER(C.c_longlong), C.c_void_p]),
    "brdf_hip_device_count": (C.c_int, []),
    "brdf_hip_last_error": (C.c_char_p, []),
    "brdf_hip_last_fit_launches": (C.c_longlong, []),
    "brdf_hip_set_launch_timing": (None, [C.c_int]),
    "brdf_hip_last_fit_kernel_us": (C.c_double, []),
    "brdf_hip_last_channels_kernel_us": (C.c_double, []),
    "brdf_hip_last_fit_stamps": (C.c_int, [C.POINTER(C.c_longlong)]),
    "brdf_hip_last_fit_trace": (C.c_int, [C.POINTER(C.c_longlong), C.c_int]),
    "brdf_hip_last_fit_stats": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                          D]),
}


def load() -> C.CDLL:
    # PyTorch-ROCm bundles its own libamdhip64; this package shares device memory and streams with
    # torch, so both must bind to ONE HIP runtime: import torch first, then our library resolves its
    # libamdhip64 dependency to the copy already in the process.  (A C/C++ host that does not use
    # torch simply links the system ROCm runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch is part of this image
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C brdf_amd/csrc`). brdf_amd has no fallback path without its HIP library.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    for name, (res, args) in ABI.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise ImportError(f"{LIB_PATH} does not export `{name}` declared in include/brdf_levmar.h") from exc
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def last_error() -> str:
    return (lib.brdf_hip_last_error() or b"").decode()
