"""diagnostic (stamps build): cycles per pass of the resident regime's sections against the number of workgroups"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
names_res = ["-", "sweep+reduce", "level1", "level2+fold", "step+build"]
for n in [1024, 8192, 32768, 65536, 131072, 262144, 1_000_000]:
    angles, x, _ = synth.make_single(2, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for method in (0, 1):
        r = brdf_amd.fit_single(method, 2, a, xd, synth.P0[2], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        st = brdf_amd.last_fit_stats()
        out = (C.c_longlong * 8)(); brdf_amd.lib.brdf_hip_last_fit_stamps(out)
        P = max(1, st['passes'])
        print(n, 'G', min(256, (n + 1023) // 1024), 'method', method, 'us/pass %.2f' % (st['device_us'] / st['passes']),
              ' '.join(f"{nm}={out[k]/P:.0f}" for k, nm in enumerate(names_res)), flush=True)
