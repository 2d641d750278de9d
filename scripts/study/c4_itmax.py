"""configs[3] dlevmar_bc_dif ends at itmax on the device far more often than in the reference.  Same machine, same
model arithmetic, summation order swapped (scripts/study/sum_order.cpp): how often does each order end at itmax, and what does
surfel 330 do?  usage: python scripts/study/c4_itmax.py [surfels=2048]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from brdf_amd import synth
lib = C.CDLL(os.path.join(ROOT, "scripts", "study", "libsum_order.so"))
D = C.POINTER(C.c_double)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = 4096
lb, ub = (np.array(v, dtype=np.float64) for v in synth.bounds(2))
o = np.array(synth.OPTS)
def fit(order, a, xs, trace=0):
    p = np.array(synth.P0[2]); info = np.zeros(10)
    r = lib.study_bc_fit(order, a.ctypes.data_as(D), xs.ctypes.data_as(D), n, p.ctypes.data_as(D), synth.ITMAX, o.ctypes.data_as(D),
                         lb.ctypes.data_as(D), ub.ctypes.data_as(D), info.ctypes.data_as(D), trace)
    return r, p, info
names = {0: "reference order", 1: "kernel order, fma", 2: "kernel order, mul+add"}
itmax = {k: 0 for k in names}; nfev = {k: 0.0 for k in names}; iters = {k: 0.0 for k in names}
for first in range(0, S, 256):
    angles, x, _ = synth.make_surfels(2, n, first=first, count=min(256, S - first))
    for s_ in range(angles.shape[0]):
        a = np.ascontiguousarray(angles[s_].reshape(-1)); xs = np.ascontiguousarray(x[s_])
        for order in names:
            r, p, info = fit(order, a, xs)
            itmax[order] += int(info[6] == 3); nfev[order] += info[7]; iters[order] += info[5]
for order, nm in names.items():
    print(f"{nm:24s}: {itmax[order]:4d} of {S} fits end at itmax, mean iterations {iters[order] / S:.2f}, mean nfev {nfev[order] / S:.1f}", flush=True)
if len(sys.argv) > 2:
    s_ = int(sys.argv[2])
    angles, x, _ = synth.make_surfels(2, n, first=s_, count=1)
    a = np.ascontiguousarray(angles[0].reshape(-1)); xs = np.ascontiguousarray(x[0])
    for order, nm in names.items():
        print(f"surfel {s_}, {nm}:")
        r, p, info = fit(order, a, xs, 1)
        print("   ->", r, info[5:8])
