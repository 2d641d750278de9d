"""Lane-per-fit kernel (n = 16, dlevmar_bc_dif): time against the number of fits, and the distribution of the fits' lengths.
A launch ends with its longest fits running alone, so T(S) = tail + S / rate.  usage: python scripts/gpu_lane_scaling.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
from brdf_amd._lib import lib

dev = torch.device("cuda:0")
n = 16
for model in (1, 2):
    pts = []
    for lg in (16, 18, 20, 21, 22):
        S = 1 << lg
        truth = torch.from_numpy(synth.surfel_truth(model, 0, S)).to(dev)
        a = torch.empty((S, 3, n), dtype=torch.float64, device=dev)
        x = torch.empty((S, n), dtype=torch.float64, device=dev)
        assert lib.brdf_hip_synth_dev(model, synth.SEED, 0, S, n, truth.data_ptr(), a.data_ptr(), x.data_ptr(), None) == 0
        lb, ub = synth.bounds(model)
        best = 1e9
        for _ in range(3):
            p0 = torch.from_numpy(np.tile(np.array(synth.P0[model]), (S, 1))).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p, info, ret = brdf_amd.fit_batch(1, model, a, x, p0, lb=lb, ub=ub, itmax=100, opts=synth.OPTS)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        nf = info[:, 7].cpu().numpy()
        it = info[:, 5].cpu().numpy()
        pts.append((S, best))
        print(f"model {model} S 2^{lg}: {best * 1e3:8.2f} ms  {S / best:.3e} fits/s | iterations mean {it.mean():.1f}, at itmax {np.mean(it >= 100) * 100:.1f} %; "
              f"nfev mean {nf.mean():.0f}, 99.9th percentile {np.percentile(nf, 99.9):.0f}, max {nf.max():.0f}", flush=True)
        del a, x, truth, p, info, ret
    (s1, t1), (s2, t2) = pts[-3], pts[-1]
    b = (t2 - t1) / (s2 - s1)
    print(f"model {model}: T(S) = {1e3 * (t1 - b * s1):.1f} ms + S / {1.0 / b:.3e} fits/s  (from S = 2^20 and 2^22)")
