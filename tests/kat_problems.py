"""The known-answer problems of the reference's lmdemo.c (starting points, data and bounds as set up in
lmdemo.c:860-1111; the problem FUNCTIONS themselves are not restated -- they are called by address from the
compiled reference object in oracle/_ref, which exports them as ordinary symbols)."""
import ctypes as C

import numpy as np

DM = 1.7976931348623157e308
OPTS = (1e-3, 1e-15, 1e-15, 1e-20, 1e-6)  # lmdemo.c:816-817
MEYER_X = [34.780, 28.610, 23.650, 19.630, 16.370, 13.720, 11.540, 9.744, 8.261, 7.030, 6.005, 5.147, 4.427, 3.820,
           3.307, 2.872]

OSBORNE_X = [8.44E-1, 9.08E-1, 9.32E-1, 9.36E-1, 9.25E-1, 9.08E-1, 8.81E-1, 8.5E-1, 8.18E-1, 7.84E-1, 7.51E-1, 7.18E-1,
             6.85E-1, 6.58E-1, 6.28E-1, 6.03E-1, 5.8E-1, 5.58E-1, 5.38E-1, 5.22E-1, 5.06E-1, 4.9E-1, 4.78E-1, 4.67E-1,
             4.57E-1, 4.48E-1, 4.38E-1, 4.31E-1, 4.24E-1, 4.2E-1, 4.14E-1, 4.11E-1, 4.06E-1]

PROBLEMS = {
    0: dict(kind="der", f="ros", j="jacros", p=[-1.2, 1.0], x=[0, 0], itmax=1000),
    1: dict(kind="der", f="modros", j="jacmodros", p=[-1.2, 1.0], x=[0, 0, 0], itmax=1000),
    2: dict(kind="der", f="powell", j="jacpowell", p=[3.0, 1.0], x=[0, 0], itmax=1000),
    5: dict(kind="der", f="osborne", j="jacosborne", p=[0.5, 1.5, -1.0, 1.0E-2, 2.0E-2], x=OSBORNE_X, itmax=1000),
    6: dict(kind="der", f="helval", j="jachelval", p=[-1.0, 0.0, 0.0], x=[0, 0, 0], itmax=1000),
    3: dict(kind="dif", f="wood", p=[-3, -1, -3, -1], x=[0] * 6, itmax=1000),
    4: dict(kind="dif", f="meyer", p=[8.85, 4.0, 2.5], x=MEYER_X, itmax=1000, covar=True),
    11: dict(kind="bc_der", f="hs01", j="jachs01", p=[-2, 1], x=[0, 0], lb=[-DM, -1.5], ub=[DM, DM], itmax=1000),
    12: dict(kind="bc_der", f="hs21", j="jachs21", p=[-1, -1], x=[0, 0], lb=[2, -50], ub=[50, 50], itmax=1000),
    13: dict(kind="bc_der", f="hatfldb", j="jachatfldb", p=[.1] * 4, x=[0] * 4, lb=[0] * 4, ub=[DM, 0.8, DM, DM],
             itmax=1000),
    14: dict(kind="bc_der", f="hatfldc", j="jachatfldc", p=[.9] * 4, x=[0] * 4, lb=[0] * 4, ub=[10] * 4, itmax=1000),
    15: dict(kind="bc_der", f="combust", j="jaccombust", p=[1e-4] * 5, x=[0] * 5, lb=[1e-4] * 5, ub=[100] * 5,
             itmax=5000),
}
# the same bc problems through the finite-difference shim (not in lmdemo's table; reference-vs-oracle only)
for _k in (11, 12, 13, 14, 15):
    _p = dict(PROBLEMS[_k])
    _p["kind"] = "bc_dif"
    PROBLEMS[100 + _k] = _p

# what the reference's lmdemo prints ("%.7g" solution; iters, reason, nfev, njev, nlss)
SURVEY_TABLE = {
    0: ("0.9432309 0.8893884", (1000, 3, 1290, 1000, 1289)),
    1: ("0.9999992 0.9999984", (14, 2, 25, 14, 25)),
    2: ("-9.00243e-11 -6.719414e-05", (198, 6, 208, 198, 207)),
    5: ("0.3754101 1.935847 -1.464687 0.01286753 0.0221227", (34, 2, 45, 34, 45)),
    6: ("1 3.691042e-13 5.857861e-13", (9, 6, 10, 9, 9)),
    3: ("1 1 1 1", (113, 6, 158, 11, 113)),
    4: ("2.481778 6.181346 3.502236", (209, 2, 273, 21, 210)),
    11: ("1 1", (14, 6, 23, 14, 14)),
    12: ("2 -4.688186e-19", (5, 1, 10, 6, 5)),
    13: ("0.9472136 0.8 0.64 0.4096", (939, 2, 3186, 939, 939)),
    14: ("1 1 1 1", (4, 6, 5, 4, 4)),
    15: ("0.00343023 31.3265 0.0683504 0.859529 0.03696244", (68, 6, 87, 68, 68)),
}


def run_problem(lib, prefix, pr, ref_lib=None):
    """Run one problem through `lib` (entry points prefix+dlevmar_*), callbacks taken from ref_lib (default lib)."""
    from tests.oracle_libs import f64, ptr
    src = ref_lib or lib
    fp = lambda name: C.cast(getattr(src, name), C.c_void_p)  # noqa: E731
    p = f64(pr["p"]).copy()
    x = f64(pr["x"])
    m, n = p.size, x.size
    info = np.zeros(10)
    opts = f64(OPTS)
    covar = np.zeros(m * m) if pr.get("covar") else None
    if pr["kind"] == "dif":
        fn = getattr(lib, prefix + "dlevmar_dif")
        r = fn(fp(pr["f"]), ptr(p), ptr(x), m, n, pr["itmax"], ptr(opts), ptr(info), None, ptr(covar), None)
    elif pr["kind"] == "der":
        fn = getattr(lib, prefix + "dlevmar_der")
        r = fn(fp(pr["f"]), fp(pr["j"]), ptr(p), ptr(x), m, n, pr["itmax"], ptr(opts), ptr(info), None, ptr(covar), None)
    else:
        lb, ub = f64(pr["lb"]), f64(pr["ub"])
        if pr["kind"] == "bc_der":
            fn = getattr(lib, prefix + "dlevmar_bc_der")
            r = fn(fp(pr["f"]), fp(pr["j"]), ptr(p), ptr(x), m, n, ptr(lb), ptr(ub), None, pr["itmax"], ptr(opts),
                   ptr(info), None, None, None)
        else:
            fn = getattr(lib, prefix + "dlevmar_bc_dif")
            r = fn(fp(pr["f"]), ptr(p), ptr(x), m, n, ptr(lb), ptr(ub), None, pr["itmax"], ptr(opts), ptr(info), None,
                   None, None)
    return r, p, info, covar


# ---- single precision (slevmar_*, levmar.h:208-310): float problems of oracle/ref_shim.c (ours), starting points and
# boxes chosen here; the known answers are what the compiled REFERENCE's slevmar_* returns (tests/golden/slevmar_kat.json)
FM = 3.4028234663852886e38
SOPTS = (1e-3, 1e-7, 1e-7, 1e-12, 1e-4)
SPROBLEMS = {
    "ros_der": dict(kind="der", f="sp_rosenbrock", j="sp_rosenbrock_jac", p=[-1.2, 1.0], x=[0, 0], itmax=200),
    "ros_dif": dict(kind="dif", f="sp_rosenbrock", p=[-1.2, 1.0], x=[0, 0], itmax=200),
    "wood_dif": dict(kind="dif", f="sp_wood", p=[-3, -1, -3, -1], x=[0] * 6, itmax=500),
    "meyer_dif": dict(kind="dif", f="sp_meyer", p=[8.85, 4.0, 2.5], x=MEYER_X, itmax=500, covar=True),
    "helval_der": dict(kind="der", f="sp_helval", j="sp_helval_jac", p=[-1.0, 0.0, 0.0], x=[0, 0, 0], itmax=200, covar=True),
    "ros_bc_der": dict(kind="bc_der", f="sp_rosenbrock", j="sp_rosenbrock_jac", p=[-2, 1], x=[0, 0], lb=[-FM, -1.5], ub=[FM, FM], itmax=200),
    "ros_bc_dif": dict(kind="bc_dif", f="sp_rosenbrock", p=[-2, 1], x=[0, 0], lb=[-FM, -1.5], ub=[0.5, FM], itmax=200),
    "hatfldb_bc_der": dict(kind="bc_der", f="sp_hatfldb", j="sp_hatfldb_jac", p=[.1] * 4, x=[0] * 4, lb=[0] * 4, ub=[FM, 0.8, FM, FM], itmax=500),
    "hatfldb_bc_dif": dict(kind="bc_dif", f="sp_hatfldb", p=[.1] * 4, x=[0] * 4, lb=[0] * 4, ub=[FM, 0.8, FM, FM], itmax=500, covar=True),
}


def run_sproblem(lib, pr, ref_lib=None):
    """one float problem through `lib`'s slevmar_* (callbacks by address from ref_lib, default lib) -> ret, p, info, covar (float32)"""
    src = ref_lib or lib
    F = C.POINTER(C.c_float)
    fp = lambda name: C.cast(getattr(src, name), C.c_void_p)  # noqa: E731
    f32 = lambda v: None if v is None else np.ascontiguousarray(np.asarray(v, dtype=np.float32))  # noqa: E731
    ptr = lambda a: None if a is None else a.ctypes.data_as(F)  # noqa: E731
    p, x = f32(pr["p"]).copy(), f32(pr["x"])
    m, n = p.size, x.size
    info, opts = np.zeros(10, dtype=np.float32), f32(SOPTS)
    covar = np.zeros(m * m, dtype=np.float32) if pr.get("covar") else None
    if pr["kind"] == "dif":
        r = lib.slevmar_dif(fp(pr["f"]), ptr(p), ptr(x), m, n, pr["itmax"], ptr(opts), ptr(info), None, ptr(covar), None)
    elif pr["kind"] == "der":
        r = lib.slevmar_der(fp(pr["f"]), fp(pr["j"]), ptr(p), ptr(x), m, n, pr["itmax"], ptr(opts), ptr(info), None, ptr(covar), None)
    else:
        lb, ub = f32(pr["lb"]), f32(pr["ub"])
        if pr["kind"] == "bc_der":
            r = lib.slevmar_bc_der(fp(pr["f"]), fp(pr["j"]), ptr(p), ptr(x), m, n, ptr(lb), ptr(ub), None, pr["itmax"], ptr(opts),
                                   ptr(info), None, ptr(covar), None)
        else:
            r = lib.slevmar_bc_dif(fp(pr["f"]), ptr(p), ptr(x), m, n, ptr(lb), ptr(ub), None, pr["itmax"], ptr(opts), ptr(info), None,
                                   ptr(covar), None)
    return r, p, info, covar
