import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
cases = [(2, 4096), (2, 1_000_000), (1, 1_000_000)]
for model, n in cases:
    angles, x, _ = synth.make_single(model, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for method in (0, 1):
        for rep in range(2):
            r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
            st = brdf_amd.last_fit_stats()
            print(model, n, method, r.ret, st, flush=True)
torch.cuda.synchronize()
