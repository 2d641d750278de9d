"""Python host API over the C ABI: torch CUDA(=HIP) tensors provide the device memory and stream."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import D, ExtraData, last_error, lib

MODEL_PHONG, MODEL_BLINN_PHONG, MODEL_WARD = 0, 1, 2
METHOD_DIF, METHOD_BC_DIF, METHOD_BC_DER, METHOD_DER = 0, 1, 2, 3  # 2 / 3: dlevmar_bc_der / dlevmar_der with the analytic Jacobian


@dataclass
class FitResult:
    ret: int  # number of iterations, or -1 (LM_ERROR)
    p: np.ndarray  # fitted parameters [3]
    info: np.ndarray  # levmar info[10]
    covar: np.ndarray | None = None


def _require(cond: bool, what: str) -> None:
    """Argument check that survives `python -O` (an `assert` would not): a bad shape, dtype or index must never reach a kernel."""
    if not cond:
        raise ValueError(what)


def _dptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(D)


def _f64(v, size):
    if v is None:
        return None
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    _require(a.size == size, f"expected {size} values, got {a.size}")
    return a


def _stream_handle(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# brdf_hip_fit_dev once more, with plain addresses for its pointer arguments: ndarray.ctypes.data_as() costs ~3 us per pointer and a
# single-fit call passes seven of them -- more host time than the launch itself.  One scratch array per call, one base address.
_FIT_DEV = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(("brdf_hip_fit_dev", lib))
_OFF_P, _OFF_LB, _OFF_UB, _OFF_DS, _OFF_OPTS, _OFF_INFO, _OFF_COVAR, _SCRATCH = 0, 3, 6, 9, 12, 17, 27, 36


def fit_single(method: int, model: int, angles, x, p0, *, lb=None, ub=None, dscl=None, itmax=100, opts=None,
               want_covar=False) -> FitResult:
    """One fit over device-resident samples.  angles: CUDA float64 tensor [3,n] (or [3n]), x: [n].

    Mirrors a dlevmar_dif / dlevmar_bc_dif call (levmar.h:112-127) with the samples already in HBM.
    """
    import torch
    _require(angles.is_cuda and x.is_cuda and angles.dtype == torch.float64 and x.dtype == torch.float64, "angles, x: CUDA float64 tensors")
    if not angles.is_contiguous():
        angles = angles.contiguous()
    if not x.is_contiguous():
        x = x.contiguous()
    n = x.numel()
    _require(angles.numel() == 3 * n, "angles must hold 3 planes of x.numel() doubles")
    buf = np.zeros(_SCRATCH)  # p | lb | ub | dscl | opts | info | covar: the call's host arguments and results
    base = buf.ctypes.data

    def put(off, v, size):
        if v is None:
            return None
        a = np.asarray(v, dtype=np.float64).reshape(-1)
        _require(a.size == size, f"expected {size} values, got {a.size}")
        buf[off:off + size] = a
        return base + 8 * off

    p_ptr = put(_OFF_P, p0, 3)
    _require(p_ptr is not None, "p0 is required")
    args = (method, model, angles.data_ptr(), x.data_ptr(), n, p_ptr, put(_OFF_LB, lb, 3), put(_OFF_UB, ub, 3), put(_OFF_DS, dscl, 3), itmax,
            put(_OFF_OPTS, opts, 5), base + 8 * _OFF_INFO, (base + 8 * _OFF_COVAR) if want_covar else None)
    dev = x.device
    if torch.cuda.current_device() == (dev.index if dev.index is not None else torch.cuda.current_device()):
        ret = _FIT_DEV(*args, torch.cuda.current_stream().cuda_stream)
    else:
        with torch.cuda.device(dev):
            ret = _FIT_DEV(*args, torch.cuda.current_stream().cuda_stream)
    return FitResult(ret, buf[_OFF_P:_OFF_P + 3].copy(), buf[_OFF_INFO:_OFF_INFO + 10].copy(),
                     buf[_OFF_COVAR:_OFF_COVAR + 9].reshape(3, 3).copy() if want_covar else None)


def fit_channels(method: int, model: int, angles, x, p0, *, lb=None, ub=None, dscl=None, itmax=100, opts=None, want_covar=False):
    """K fits over ONE set of planes (brdf_hip_fit_channels_dev): the three colour channels of a capture, as the reference's
    callers fit them (brdfdata.cpp:1159-1181).  angles: CUDA float64 [3,n]; x: CUDA float64 [K,n]; p0: [K,3] (or [3]: the same
    start for every channel).  Returns a list of K FitResult; `last_channels_stats()` tells whether they shared one launch."""
    import torch
    _require(angles.is_cuda and x.is_cuda and angles.dtype == torch.float64 and x.dtype == torch.float64, "angles, x: CUDA float64 tensors")
    angles, x = angles.contiguous(), x.contiguous()
    _require(x.dim() == 2 and angles.numel() == 3 * x.shape[1], "angles [3,n], x [K,n]")
    K, n = int(x.shape[0]), int(x.shape[1])
    p = np.ascontiguousarray(np.broadcast_to(np.asarray(p0, dtype=np.float64).reshape(-1, 3), (K, 3))).copy()
    info = np.zeros((K, 10))
    covar = np.zeros((K, 9)) if want_covar else None
    lba, uba, dsa = _f64(lb, 3), _f64(ub, 3), _f64(dscl, 3)
    oa = _f64(opts, 5) if opts is not None else None
    with torch.cuda.device(angles.device):
        rc = lib.brdf_hip_fit_channels_dev(method, model, angles.data_ptr(), x.data_ptr(), n, n, K, _dptr(p), _dptr(lba), _dptr(uba),
                                           _dptr(dsa), itmax, _dptr(oa), _dptr(info), _dptr(covar), _stream_handle(torch))
    out = []
    for c in range(K):
        failed = info[c, 6] in (4, 7) or (rc != 0 and info[c, 6] == 0)  # levmar's LM_ERROR cases (lm_core.c:841); the ABI returns the worst channel
        out.append(FitResult(-1 if failed else int(info[c, 5]), p[c].copy(), info[c].copy(), None if covar is None else covar[c].reshape(3, 3).copy()))
    return out


def last_channels_stats(channels: int = 3):
    """per channel of the last fit_channels(): passes, jac_passes, device_us; and whether the channels shared one launch"""
    shared = C.c_int(0)
    per = []
    for c in range(channels):
        passes, jac, us = C.c_longlong(0), C.c_longlong(0), C.c_double(0.0)
        lib.brdf_hip_last_channels_stats(c, C.byref(shared), C.byref(passes), C.byref(jac), C.byref(us))
        per.append({"passes": passes.value, "jac_passes": jac.value, "device_us": us.value})
    return {"shared_launch": bool(shared.value), "channels": per, "kernel_us": lib.brdf_hip_last_channels_kernel_us()}


def fit_batch(method: int, model: int, angles, x, p0, *, lb=None, ub=None, itmax=100, opts=None):
    """S independent fits.  angles: CUDA float64 [S,3,n], x: [S,n], p0: CUDA float64 [S,3] (updated in place).

    Returns (p [S,3], info [S,10], ret [S] int32) as CUDA tensors; asynchronous on the current stream.
    """
    import torch
    _require(angles.is_cuda and x.is_cuda and p0.is_cuda and angles.device == x.device == p0.device, "bad argument: " 'angles.is_cuda and x.is_cuda and p0.is_cuda and angles.device == x.device == p0.device')
    _require(angles.dtype == torch.float64 and x.dtype == torch.float64 and p0.dtype == torch.float64, "bad argument: " 'angles.dtype == torch.float64 and x.dtype == torch.float64 and p0.dtype == torch.float64')  # the kernels read raw doubles
    S, n = x.shape
    _require(tuple(angles.shape) == (S, 3, n) and tuple(p0.shape) == (S, 3), "bad argument: " 'tuple(angles.shape) == (S, 3, n) and tuple(p0.shape) == (S, 3)')
    angles = angles.contiguous()
    x = x.contiguous()
    p = p0.contiguous()
    info = torch.zeros((S, 10), dtype=torch.float64, device=x.device)
    ret = torch.zeros((S,), dtype=torch.int32, device=x.device)
    lb_a, ub_a, op_a = _f64(lb, 3), _f64(ub, 3), _f64(opts, 5)
    with torch.cuda.device(x.device):
        rc = lib.brdf_hip_fit_batch_dev(method, model, angles.data_ptr(), x.data_ptr(), S, n, p.data_ptr(), _dptr(lb_a),
                                        _dptr(ub_a), itmax, _dptr(op_a), info.data_ptr(), ret.data_ptr(),
                                        _stream_handle(torch))
    if rc != 0:
        raise RuntimeError(f"brdf_hip_fit_batch_dev failed: {last_error()}")
    return p, info, ret


def model_eval(model: int, angles, p):
    """hx = model(p; samples) on the device (kernel K1 alone).  angles: CUDA float64 [3,n]."""
    import torch
    _require(angles.is_cuda and angles.dtype == torch.float64 and angles.numel() % 3 == 0, "bad argument: " 'angles.is_cuda and angles.dtype == torch.float64 and angles.numel() % 3 == 0')
    angles = angles.contiguous()
    n = angles.numel() // 3
    hx = torch.empty(n, dtype=torch.float64, device=angles.device)
    pa = _f64(p, 3)
    with torch.cuda.device(angles.device):
        rc = lib.brdf_hip_model_eval_dev(model, angles.data_ptr(), n, _dptr(pa), hx.data_ptr(), _stream_handle(torch))
    if rc != 0:
        raise RuntimeError(f"brdf_hip_model_eval_dev failed: {last_error()}")
    return hx


def led_table() -> np.ndarray:
    """CBRDFdata::InitLEDs (brdfdata.cpp:683-752): the capture rig's 16 LED positions, [16, 3]."""
    out = np.zeros(48)
    lib.brdf_hip_led_table(_dptr(out))
    return out.reshape(16, 3)


def _check_indices(idx, upper: int, what: str, lower: int = 0) -> None:
    """Range check of an index tensor.  The two reductions synchronise with the device; callers that keep a stream busy
    and vouch for their indices pass validate=False to the entry point instead."""
    if idx.numel() == 0:
        return
    lo, hi = int(idx.min()), int(idx.max())
    _require(lo >= lower and hi < upper, f"{what} (range [{lo}, {hi}], allowed [{lower}, {upper}))")


def cosines(vertices, faces, face_normals, leds, view_origin, *, surfels=None, rv_mode: int = 0, validate: bool = True):
    """vectors -> cosines on the device (GetCosLN / GetCosNH / GetCosRV, brdfdata.cpp:799-943) for a batch of surfels.
    vertices [nv,3] float64, faces [nf,3] int32, face_normals [nf,3] float64: CUDA tensors; surfels: CUDA int32 [S]
    (face index per surfel) or None (surfel s = face s); leds [L,3], view_origin [3]: host.  Returns CUDA float64
    [S, 3, L] -- the batched fitter's `angles` layout."""
    import torch
    vertices, faces, face_normals = vertices.contiguous(), faces.contiguous(), face_normals.contiguous()
    _require(vertices.is_cuda and faces.is_cuda and face_normals.is_cuda, "bad argument: " 'vertices.is_cuda and faces.is_cuda and face_normals.is_cuda')
    _require(vertices.dtype == torch.float64 and face_normals.dtype == torch.float64 and faces.dtype == torch.int32, "bad argument: " 'vertices.dtype == torch.float64 and face_normals.dtype == torch.float64 and faces.dtype == torch.int32')
    _require(vertices.shape[-1] == 3 and faces.shape[-1] == 3 and tuple(face_normals.shape) == (faces.shape[0], 3), "bad argument: " 'vertices.shape[-1] == 3 and faces.shape[-1] == 3 and tuple(face_normals.shape) == (faces.shape[0], 3)')
    if validate:
        _check_indices(faces, vertices.shape[0], "face index outside the vertex array")
    if surfels is not None:
        surfels = surfels.contiguous()
        _require(surfels.is_cuda and surfels.dtype == torch.int32, "bad argument: " 'surfels.is_cuda and surfels.dtype == torch.int32')
        if validate:
            _check_indices(surfels, faces.shape[0], "surfel index outside the face array")
    S = int(surfels.numel()) if surfels is not None else int(faces.shape[0])
    la = np.ascontiguousarray(leds, dtype=np.float64).reshape(-1, 3)
    L = la.shape[0]
    va = _f64(view_origin, 3)
    out = torch.empty((S, 3, L), dtype=torch.float64, device=vertices.device)
    with torch.cuda.device(vertices.device):
        rc = lib.brdf_hip_cosines_dev(vertices.data_ptr(), faces.data_ptr(), face_normals.data_ptr(),
                                      surfels.data_ptr() if surfels is not None else None, S, _dptr(la), L, _dptr(va), rv_mode,
                                      out.data_ptr(), _stream_handle(torch))
    if rc != 0:
        raise RuntimeError(f"brdf_hip_cosines_dev failed: {last_error()}")
    return out


def fit_capture(model: int, images, pixel_map, vertices, faces, face_normals, leds, view_origin, *, rv_mode: int = 0,
                p0=(0.5, 1.0, 1.0), lb=(0.0, 0.0, 0.0), ub=(100.0, 100.0, 100.0), itmax: int = 100, opts=None, brdf_surfaces=None,
                validate: bool = True):
    """The pixel loop of CBRDFdata::CalcBRDFEquation (brdfdata.cpp:1188-1227) on the device.  images: CUDA uint8
    [L,H,W,3] (BGR), pixel_map: CUDA int32 [H,W] (face index or -1), mesh as in cosines().  Returns (brdf_surfaces
    CUDA float64 [nf,3,3] = {kd,ks,n} per face and channel, avg[3], number of pixels that carried a face)."""
    import torch
    images, pixel_map = images.contiguous(), pixel_map.contiguous()
    vertices, faces, face_normals = vertices.contiguous(), faces.contiguous(), face_normals.contiguous()
    _require(images.is_cuda and pixel_map.is_cuda and vertices.is_cuda and faces.is_cuda and face_normals.is_cuda, "bad argument: " 'images.is_cuda and pixel_map.is_cuda and vertices.is_cuda and faces.is_cuda and face_normals.is_cuda')
    _require(images.dtype == torch.uint8 and pixel_map.dtype == torch.int32 and faces.dtype == torch.int32, "bad argument: " 'images.dtype == torch.uint8 and pixel_map.dtype == torch.int32 and faces.dtype == torch.int32')
    _require(vertices.dtype == torch.float64 and face_normals.dtype == torch.float64, "bad argument: " 'vertices.dtype == torch.float64 and face_normals.dtype == torch.float64')
    _require(images.dim() == 4 and images.shape[3] == 3 and tuple(pixel_map.shape) == tuple(images.shape[1:3]), "bad argument: " 'images.dim() == 4 and images.shape[3] == 3 and tuple(pixel_map.shape) == tuple(images.shape[1:3])')  # [L,H,W,3] BGR, [H,W]
    _require(tuple(face_normals.shape) == (faces.shape[0], 3) and vertices.shape[-1] == 3, "bad argument: " 'tuple(face_normals.shape) == (faces.shape[0], 3) and vertices.shape[-1] == 3')
    if validate:  # (-1 = no face under the pixel)
        _check_indices(pixel_map, faces.shape[0], "pixel map names a face that does not exist", lower=-1)
        _check_indices(faces, vertices.shape[0], "face index outside the vertex array")
    L, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
    nf = int(faces.shape[0])
    if brdf_surfaces is None:
        brdf_surfaces = torch.zeros((nf, 3, 3), dtype=torch.float64, device=images.device)
    la = np.ascontiguousarray(leds, dtype=np.float64).reshape(-1, 3)
    _require(la.shape[0] == L, "bad argument: " 'la.shape[0] == L')
    va, pa, lba, uba = _f64(view_origin, 3), _f64(p0, 3), _f64(lb, 3), _f64(ub, 3)
    oa = _f64(opts, 5) if opts is not None else None
    avg = np.zeros(3)
    npx = C.c_longlong(0)
    with torch.cuda.device(images.device):
        rc = lib.brdf_hip_fit_capture_dev(model, images.data_ptr(), L, H, W, pixel_map.data_ptr(), vertices.data_ptr(),
                                          faces.data_ptr(), face_normals.data_ptr(), nf, _dptr(la), _dptr(va), rv_mode,
                                          _dptr(pa), _dptr(lba), _dptr(uba), itmax, _dptr(oa) if oa is not None else None,
                                          brdf_surfaces.data_ptr(), _dptr(avg), C.byref(npx), _stream_handle(torch))
    if rc != 0:
        raise RuntimeError(f"brdf_hip_fit_capture_dev failed: {last_error()}")
    return brdf_surfaces, avg, npx.value


def fit_capture_single(model: int, images, pixel_map, vertices, faces, face_normals, leds, view_origin, *, rv_mode: int = 0,
                       p0=(0.0, 0.0, 0.0), lb=(0.0, 0.0, 0.0), ub=(100.0, 100.0, 100.0), itmax: int = 2000,
                       opts=(1e-3, 1e-15, 1e-10, 1e-50, 1.0), validate: bool = True):
    """CalcBRDFEquation_SingleBRDF (brdfdata.cpp:1138-1186) on the device: one {kd, ks, n} per colour channel for the
    whole object.  Defaults are the reference's call-site values (brdfdata.cpp:1002, :1046-1056).  Returns
    (single_brdf [3,3], info [3,10], faces used)."""
    import torch
    images, pixel_map = images.contiguous(), pixel_map.contiguous()
    vertices, faces, face_normals = vertices.contiguous(), faces.contiguous(), face_normals.contiguous()
    _require(images.is_cuda and pixel_map.is_cuda and vertices.is_cuda and faces.is_cuda and face_normals.is_cuda, "bad argument: " 'images.is_cuda and pixel_map.is_cuda and vertices.is_cuda and faces.is_cuda and face_normals.is_cuda')
    _require(images.dtype == torch.uint8 and pixel_map.dtype == torch.int32 and faces.dtype == torch.int32, "bad argument: " 'images.dtype == torch.uint8 and pixel_map.dtype == torch.int32 and faces.dtype == torch.int32')
    _require(vertices.dtype == torch.float64 and face_normals.dtype == torch.float64, "bad argument: " 'vertices.dtype == torch.float64 and face_normals.dtype == torch.float64')
    _require(images.dim() == 4 and images.shape[3] == 3 and tuple(pixel_map.shape) == tuple(images.shape[1:3]), "bad argument: " 'images.dim() == 4 and images.shape[3] == 3 and tuple(pixel_map.shape) == tuple(images.shape[1:3])')  # [L,H,W,3] BGR, [H,W]
    _require(tuple(face_normals.shape) == (faces.shape[0], 3) and vertices.shape[-1] == 3, "bad argument: " 'tuple(face_normals.shape) == (faces.shape[0], 3) and vertices.shape[-1] == 3')
    if validate:  # (-1 = no face under the pixel)
        _check_indices(pixel_map, faces.shape[0], "pixel map names a face that does not exist", lower=-1)
        _check_indices(faces, vertices.shape[0], "face index outside the vertex array")
    L, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
    nf = int(faces.shape[0])
    la = np.ascontiguousarray(leds, dtype=np.float64).reshape(-1, 3)
    _require(la.shape[0] == L, "bad argument: " 'la.shape[0] == L')
    va, pa, lba, uba, oa = _f64(view_origin, 3), _f64(p0, 3), _f64(lb, 3), _f64(ub, 3), _f64(opts, 5)
    out, info = np.zeros(9), np.zeros(30)
    nfu = C.c_longlong(0)
    with torch.cuda.device(images.device):
        rc = lib.brdf_hip_fit_capture_single_dev(model, images.data_ptr(), L, H, W, pixel_map.data_ptr(), vertices.data_ptr(),
                                                 faces.data_ptr(), face_normals.data_ptr(), nf, _dptr(la), _dptr(va), rv_mode,
                                                 _dptr(pa), _dptr(lba), _dptr(uba), itmax, _dptr(oa), _dptr(out), _dptr(info),
                                                 C.byref(nfu), _stream_handle(torch))
    if rc != 0:
        raise RuntimeError(f"brdf_hip_fit_capture_single_dev failed: {last_error()}")
    return out.reshape(3, 3), info.reshape(3, 10), nfu.value


def host_dlevmar(method: int, model: int, angles: np.ndarray, x: np.ndarray, p0, *, lb=None, ub=None, dscl=None,
                 itmax=100, opts=None, want_covar=False) -> FitResult:
    """The drop-in call exactly as brdfdata.cpp:1058/1119 makes it: HOST arrays, a BRDFFunc-style callback
    (here the library's own BRDFFunc_hip) and a struct extraData payload."""
    angles = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size
    p = _f64(p0, 3).copy()
    info = np.zeros(10)
    covar = np.zeros(9) if want_covar else None
    ed = ExtraData(_dptr(angles), model)
    func = C.cast(lib.BRDFFunc_hip, C.c_void_p)
    lb_a, ub_a, ds_a, op_a = _f64(lb, 3), _f64(ub, 3), _f64(dscl, 3), _f64(opts, 5)
    if method == METHOD_DIF:
        ret = lib.dlevmar_dif(func, _dptr(p), _dptr(x), 3, n, itmax, _dptr(op_a), _dptr(info), None, _dptr(covar),
                              C.byref(ed))
    elif method == METHOD_DER:  # dlevmar_der(BRDFFunc_hip, BRDFJac_hip, ...)
        ret = lib.dlevmar_der(func, C.cast(lib.BRDFJac_hip, C.c_void_p), _dptr(p), _dptr(x), 3, n, itmax, _dptr(op_a), _dptr(info),
                              None, _dptr(covar), C.byref(ed))
    elif method == METHOD_BC_DER:  # dlevmar_bc_der(BRDFFunc_hip, BRDFJac_hip, ...): the analytic Jacobian, all on the device
        ret = lib.dlevmar_bc_der(func, C.cast(lib.BRDFJac_hip, C.c_void_p), _dptr(p), _dptr(x), 3, n, _dptr(lb_a), _dptr(ub_a),
                                 _dptr(ds_a), itmax, _dptr(op_a), _dptr(info), None, _dptr(covar), C.byref(ed))
    else:
        ret = lib.dlevmar_bc_dif(func, _dptr(p), _dptr(x), 3, n, _dptr(lb_a), _dptr(ub_a), _dptr(ds_a), itmax,
                                 _dptr(op_a), _dptr(info), None, _dptr(covar), C.byref(ed))
    return FitResult(ret, p, info, None if covar is None else covar.reshape(3, 3))


def model_jacobian(model: int, angles: np.ndarray, p) -> np.ndarray:
    """BRDFJac_hip through host pointers: the analytic Jacobian [n, 3] of a built-in model."""
    a = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
    n = a.size // 3
    jac = np.zeros(3 * n)
    pa = _f64(p, 3).copy()
    ed = ExtraData(_dptr(a), model)
    lib.BRDFJac_hip(_dptr(pa), _dptr(jac), 3, n, C.byref(ed))
    return jac.reshape(n, 3)


def chkjac(model: int, angles: np.ndarray, p) -> np.ndarray:
    """dlevmar_chkjac (misc_core.c:250-321) on BRDFFunc_hip / BRDFJac_hip: err[n], ~1 where the Jacobian row is right."""
    a = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
    n = a.size // 3
    err = np.zeros(n)
    pa = _f64(p, 3).copy()
    ed = ExtraData(_dptr(a), model)
    lib.dlevmar_chkjac(C.cast(lib.BRDFFunc_hip, C.c_void_p), C.cast(lib.BRDFJac_hip, C.c_void_p), _dptr(pa), 3, n, C.byref(ed),
                       _dptr(err))
    return err


def last_fit_stats() -> dict:
    a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
    us = C.c_double(0.0)
    lib.brdf_hip_last_fit_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(us))
    return {"passes": a.value, "jac_passes": b.value, "eval_passes": c.value, "device_us": us.value,
            "launches": lib.brdf_hip_last_fit_launches(), "kernel_us": lib.brdf_hip_last_fit_kernel_us()}


def set_launch_timing(on: bool) -> None:
    """resident fits bracket their launch with a HIP event pair on the launch stream; last_fit_stats()["kernel_us"] /
    last_channels_stats()["kernel_us"] then hold the kernel's duration (-1 otherwise)"""
    lib.brdf_hip_set_launch_timing(1 if on else 0)
