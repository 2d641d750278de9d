"""dlevmar_bc_der with the analytic device Jacobian against dlevmar_bc_dif: 1M-sample fits"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
for model in (2, 1):
    angles, x, _ = synth.make_single(model, 1_000_000)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    lb, ub = synth.bounds(model)
    for method in (1, 2):
        for _ in range(3):
            r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=lb, ub=ub, itmax=100, opts=synth.OPTS)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N = 10
        for _ in range(N):
            r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=lb, ub=ub, itmax=100, opts=synth.OPTS)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
        st = brdf_amd.last_fit_stats()
        print(f"model {model} {'bc_dif' if method == 1 else 'bc_der'}: {dt*1e3:.3f} ms/fit, iterations {r.ret}, nfev {r.info[7]:.0f}, njev {r.info[8]:.0f}, "
              f"passes {st['passes']}, p = {r.p}", flush=True)
