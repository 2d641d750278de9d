"""First GPU contact: smoke, then timing of single 1M-sample fits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.smoke()
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
for model in (2, 1):
    n = 1_000_000
    angles, x, _ = synth.make_single(model, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for method in (0, 1):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            st = brdf_amd.last_fit_stats()
            print(f"model={model} method={method} ret={r.ret} p={r.p} nfev={r.info[7]:.0f} wall={dt*1e3:.3f} ms "
                  f"dev={st['device_us']:.1f} us passes={st['passes']} us/pass={st['device_us']/max(1,st['passes']):.2f} "
                  f"evals/s={r.info[7]*n/dt:.3e}", flush=True)

# PCIe-inclusive: the drop-in entry point with HOST pointers (allocation + 32 MB upload + fit), Ward 1M
angles, x, _ = synth.make_single(2, 1_000_000)
for method in (0, 1):
    for rep in range(3):
        t0 = time.perf_counter()
        r = brdf_amd.host_dlevmar(method, 2, angles, x, synth.P0[2], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        dt = time.perf_counter() - t0
    print(f"host-pointer drop-in method={method}: {dt*1e3:.3f} ms per call, {r.info[7]*1e6/dt:.3e} evals/s (PCIe + allocation inclusive)", flush=True)
