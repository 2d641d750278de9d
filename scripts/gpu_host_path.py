"""The drop-in entry points with HOST pointers (what brdfdata.cpp:1058 / :1119 call): per-call latency, PCIe and staging
included -- 10^6-sample fits and the application's literal 16-sample call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import brdf_amd
from brdf_amd import synth
for model, n, reps in ((2, 1_000_000, 8), (1, 1_000_000, 8), (1, 16, 400)):
    angles, x, _ = synth.make_single(model, n)
    for method, name in ((0, "dlevmar_dif"), (1, "dlevmar_bc_dif")):
        for _ in range(3):
            r = brdf_amd.host_dlevmar(method, model, angles, x, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        t0 = time.perf_counter()
        for _ in range(reps):
            r = brdf_amd.host_dlevmar(method, model, angles, x, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        dt = (time.perf_counter() - t0) / reps
        print(f"host-pointer {name} model {model} n={n}: {dt*1e6:.1f} us per call (ret {r.ret}, nfev {r.info[7]:.0f}), "
              f"{r.info[7]*n/dt:.3e} residual-evals/s, {1/dt:.0f} calls/s", flush=True)
