"""ctypes access to the checkers: oracle/liboracle.so (our CPU restatement), oracle/_ref (the reference's
own levmar compiled from /root/reference, when it was built) and tests/cpp/libhost_machine.so (the product's
LM state machines driven on the host with reference-order sums).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = C.POINTER(C.c_double)

orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
_ref_path = os.path.join(ROOT, "oracle", "_ref", "liblevmar_ref.so")
ref = C.CDLL(_ref_path) if os.path.exists(_ref_path) else None
hm = C.CDLL(os.path.join(ROOT, "tests", "cpp", "libhost_machine.so"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(D)


def f64(v):
    return None if v is None else np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))


def brdf_fit(which: str, method: int, model: int, angles, x, p0, itmax=100, opts=None, lb=None, ub=None):
    """One BRDF fit through the oracle ('orc'), the compiled reference ('ref') or the host-driven product
    state machine ('hm').  Returns (ret, p[3], info[10])."""
    a = f64(angles)
    xx = f64(x)
    n = xx.size
    p = f64(p0).copy()
    info = np.zeros(10)
    o, l, u = f64(opts), f64(lb), f64(ub)
    if which == "orc":
        r = orc.orc_brdf_fit(method, model, ptr(a), ptr(xx), n, ptr(p), itmax, ptr(o), ptr(l), ptr(u), ptr(info))
    elif which == "ref":
        assert ref is not None, "oracle/_ref was not built (no /root/reference here and no prebuilt .so)"
        r = ref.ref_brdf_fit(method, model, ptr(a), ptr(xx), n, ptr(p), itmax, ptr(o), ptr(l), ptr(u), ptr(info))
    elif which in ("hm", "hm_fast"):
        fn = hm.hm_brdf_fit if which == "hm" else hm.hm_brdf_fit_fast
        bc = method in (1, 2)
        r = fn(method, model, ptr(a), ptr(xx), n, ptr(p), itmax, ptr(o), ptr(l) if bc else None,
               ptr(u) if bc else None, None, ptr(info), None, None)
    else:
        raise ValueError(which)
    return r, p, info


def model_values(model: int, angles, p):
    a = f64(angles)
    n = a.size // 3
    hx = np.zeros(n)

    class Extra(C.Structure):
        _fields_ = [("angles", D), ("modelInfo", C.c_int)]

    ed = Extra(ptr(a), model)
    pp = f64(p).copy()
    orc.orc_brdf_func(ptr(pp), ptr(hx), 3, n, C.byref(ed))
    return hx


def ref_model_values(model: int, angles, p):
    """oracle/_ref: model values through the REFERENCE's own BRDFFunc (brdfdata.cpp:962-989 compiled in place) for models
    0 / 1; Ward (not in the reference) through the restated callback"""
    assert ref is not None, "oracle/_ref was not built"
    a = f64(angles)
    n = a.size // 3
    hx = np.zeros(n)
    ref.ref_brdf_values(model, ptr(a), n, ptr(f64(p).copy()), ptr(hx))
    return hx


def rel_err(p, p_ref):
    p, p_ref = np.asarray(p), np.asarray(p_ref)
    return float(np.max(np.abs(p - p_ref) / np.maximum(np.abs(p_ref), 1e-12)))


def cosines(vertices, faces, normals, leds, view, surfels=None, rv_mode=0):
    """oracle/cosines_oracle.c: angles[S][3][L] for the surfels (face indices; None = every face in order)"""
    v, nr, ld, vw = f64(vertices), f64(normals), f64(leds), f64(view)
    fc = np.ascontiguousarray(np.asarray(faces, dtype=np.int32).reshape(-1))
    sf = None if surfels is None else np.ascontiguousarray(np.asarray(surfels, dtype=np.int32))
    S = fc.size // 3 if sf is None else sf.size
    L = ld.size // 3
    out = np.zeros((S, 3, L))
    IP = C.POINTER(C.c_int)
    orc.orc_cosines(ptr(v), fc.ctypes.data_as(IP), ptr(nr), None if sf is None else sf.ctypes.data_as(IP), C.c_longlong(S),
                    ptr(ld), L, ptr(vw), rv_mode, ptr(out))
    return out


def led_table():
    out = np.zeros(48)
    orc.orc_led_table(ptr(out))
    return out.reshape(16, 3)


def fit_capture(model, images, pixel_map, vertices, faces, normals, leds, view, rv_mode=0, p0=(0.5, 1.0, 1.0),
                lb=(0.0, 0.0, 0.0), ub=(100.0, 100.0, 100.0), itmax=100, opts=None):
    """oracle/cosines_oracle.c: orc_fit_capture -> (brdf_surfaces[nf,3,3], avg[3], pixels)"""
    img = np.ascontiguousarray(images, dtype=np.uint8)
    pm = np.ascontiguousarray(pixel_map, dtype=np.int32)
    Lc, H, W = img.shape[:3]
    fc = np.ascontiguousarray(np.asarray(faces, dtype=np.int32).reshape(-1))
    nf = fc.size // 3
    out = np.zeros((nf, 3, 3))
    avg = np.zeros(3)
    IP = C.POINTER(C.c_int)
    orc.orc_fit_capture.restype = C.c_longlong
    n = orc.orc_fit_capture(model, img.ctypes.data_as(C.c_void_p), Lc, H, W, pm.ctypes.data_as(IP), ptr(f64(vertices)),
                            fc.ctypes.data_as(IP), ptr(f64(normals)), nf, ptr(f64(leds)), ptr(f64(view)), rv_mode, ptr(f64(p0)),
                            ptr(f64(lb)), ptr(f64(ub)), itmax, ptr(f64(opts)), ptr(out), ptr(avg))
    return out, avg, n


def fit_capture_single(model, images, pixel_map, vertices, faces, normals, leds, view, rv_mode=0, p0=(0.0, 0.0, 0.0),
                       lb=(0.0, 0.0, 0.0), ub=(100.0, 100.0, 100.0), itmax=2000, opts=(1e-3, 1e-15, 1e-10, 1e-50, 1.0)):
    """oracle/cosines_oracle.c: orc_fit_capture_single -> (single_brdf[3,3], info[3,10], faces used)"""
    img = np.ascontiguousarray(images, dtype=np.uint8)
    pm = np.ascontiguousarray(pixel_map, dtype=np.int32)
    Lc, H, W = img.shape[:3]
    fc = np.ascontiguousarray(np.asarray(faces, dtype=np.int32).reshape(-1))
    nf = fc.size // 3
    out, info = np.zeros(9), np.zeros(30)
    work = np.zeros(6 * Lc * nf)
    last = np.zeros(nf, dtype=np.int64)
    IP = C.POINTER(C.c_int)
    orc.orc_fit_capture_single.restype = C.c_longlong
    F = orc.orc_fit_capture_single(model, img.ctypes.data_as(C.c_void_p), Lc, H, W, pm.ctypes.data_as(IP), ptr(f64(vertices)),
                                   fc.ctypes.data_as(IP), ptr(f64(normals)), nf, ptr(f64(leds)), ptr(f64(view)), rv_mode,
                                   ptr(f64(p0)), ptr(f64(lb)), ptr(f64(ub)), itmax, ptr(f64(opts)), ptr(out), ptr(info), ptr(work),
                                   last.ctypes.data_as(C.POINTER(C.c_longlong)))
    return out.reshape(3, 3), info.reshape(3, 10), F
