"""stress: many single fits of random size / model / entry point, resident regime against the launch chain.
Every resident fit must finish in ONE launch (no silent fallback) and agree with the chain's result."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
N = int(os.environ.get("STRESS_FITS", "200"))
worst = 0.0
t0 = time.time()
for it in range(N):
    model = int(rng.integers(0, 3)); method = int(rng.integers(0, 3))
    n = int(2 ** rng.uniform(np.log2(200), np.log2(1_048_576)))
    angles, x, _ = synth.make_single(model, n, seed=synth.SEED + it)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    lb, ub = synth.bounds(model)
    res = {}
    for env in ("1", "0"):
        os.environ["BRDF_HIP_RESIDENT"] = env
        r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=lb, ub=ub, itmax=100, opts=synth.OPTS)
        st = brdf_amd.last_fit_stats()
        assert r.ret >= 0, (it, model, method, n, env, brdf_amd.last_error())
        assert (st["launches"] == 1) == (env == "1"), (it, model, method, n, env, st)
        res[env] = r
    rel = float(np.max(np.abs(res["1"].p - res["0"].p) / np.maximum(np.abs(res["0"].p), 1e-12)))
    e_rel = abs(res["1"].info[1] - res["0"].info[1]) / res["0"].info[1]
    worst = max(worst, rel)
    assert e_rel <= 1e-8, (it, model, method, n, res["1"].info[1], res["0"].info[1])
    assert rel <= 1e-5, (it, model, method, n, res["1"].p, res["0"].p)
    if it % 25 == 24:
        print(f"{it + 1} fits, worst rel diff of p {worst:.2e}, {time.time() - t0:.1f} s", flush=True)
print("stress ok:", N, "fits, worst rel diff", worst)
