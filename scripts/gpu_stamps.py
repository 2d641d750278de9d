import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
names = ["load", "fold", "step", "uniforms", "persist", "sweep", "reduce"]           # launch chain (stream_fit.hip)
names_res = ["-", "sweep+reduce", "gather", "fold", "step+build", "-", "-"]   # resident regime, control wave view (resident_fit.hip)
for model, n in [(2, 4096), (2, 1_000_000), (1, 1_000_000)]:
    angles, x, _ = synth.make_single(model, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for method in (0, 1):
        r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        st = brdf_amd.last_fit_stats()
        out = (C.c_longlong * 8)(); brdf_amd.lib.brdf_hip_last_fit_stamps(out)
        P = max(1, st['passes'] - 1)
        print(model, n, method, r.ret, 'us/pass %.2f' % (st['device_us'] / st['passes']),
              'launches', st['launches'], ' '.join(f"{nm}={out[k]/P:.0f}" for k, nm in enumerate(names_res if st['launches'] == 1 else names)), 'cycles/pass', flush=True)
