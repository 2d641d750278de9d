import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
model, n = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
angles, x, _ = synth.make_single(model, n)
a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
for method in (1, 2):
    lb, ub = synth.bounds(model) if method == 2 else (synth.LB, synth.UB)
    for sj in ("1", "0"):
        os.environ["BRDF_HIP_SPEC_JAC"] = sj
        r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=lb, ub=ub, itmax=100, opts=synth.OPTS, want_covar=True)
        print(method, sj, r.ret, [v.hex() for v in r.p], [float(v).hex() for v in r.info[2:5]], [float(v).hex() for v in r.covar.reshape(-1)[:4]], brdf_amd.last_fit_stats()["passes"], flush=True)
