import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
from brdf_amd._lib import lib
dev = torch.device("cuda:0")
def gen(model, S, n):
    truth = torch.from_numpy(synth.surfel_truth(model, 0, S)).to(dev)
    a = torch.empty((S, 3, n), dtype=torch.float64, device=dev); x = torch.empty((S, n), dtype=torch.float64, device=dev)
    assert lib.brdf_hip_synth_dev(model, synth.SEED, 0, S, n, truth.data_ptr(), a.data_ptr(), x.data_ptr(), None) == 0
    torch.cuda.synchronize(); return a, x
for model in (1, 2):
  for (S, n) in [(4096, 4096), (65536, 256), (65536, 16)]:
    a, x = gen(model, S, n)
    for method in (0, 1):
        for rep in range(2):
            p0 = torch.from_numpy(np.tile(np.array(synth.P0[model]), (S, 1))).to(dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lb, ub = synth.bounds(model)
            p, info, ret = brdf_amd.fit_batch(method, model, a, x, p0, lb=lb, ub=ub, itmax=100, opts=synth.OPTS)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nfev = info[:, 7].sum().item(); it = info[:, 5].mean().item(); bad = (ret < 0).sum().item()
        print(f"model={model} S={S} n={n} method={method}: {dt*1e3:.1f} ms, fits/s={S/dt:.3e}, evals/s={nfev*n/dt:.3e}, mean iters={it:.1f}, mean nfev={nfev/S:.0f}, failed={bad}", flush=True)
