"""diagnostic: wall time per single fit against the kernel's own clock (host-side overhead of one fit)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
angles, x, _ = synth.make_single(2, 1_000_000)
a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
for method in (0, 1):
    for _ in range(3):
        brdf_amd.fit_single(method, 2, a, xd, synth.P0[2], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
    torch.cuda.synchronize()
    N = 30
    dev_us = 0.0
    t0 = time.perf_counter()
    for _ in range(N):
        brdf_amd.fit_single(method, 2, a, xd, synth.P0[2], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        dev_us += brdf_amd.last_fit_stats()["device_us"]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N * 1e6
    print(f"method {method}: wall {wall:.1f} us/fit, device clock first pass -> result {dev_us / N:.1f} us/fit, outside {wall - dev_us / N:.1f} us", flush=True)
