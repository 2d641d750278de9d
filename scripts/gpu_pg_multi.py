import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
for model, n in [(0, 1_000_000), (1, 1_000_000), (2, 1_000_000), (1, 100_003), (2, 100_003), (1, 4096), (2, 4096)]:
    angles, x, _ = synth.make_single(model, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for K in ("1", "8"):
        os.environ["BRDF_HIP_PG_MULTI"] = K
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = brdf_amd.fit_single(1, model, a, xd, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = brdf_amd.last_fit_stats()
        print(f"model={model} n={n} K={K}: ret={r.ret} nfev={r.info[7]:.0f} passes={st['passes']} wall={dt*1e3:.3f} ms p={r.p}", flush=True)
