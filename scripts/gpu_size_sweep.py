"""time per single Ward fit against the number of samples (resident regime up to 2^20 samples, launch chain beyond)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
print("| n | entry | regime | passes | ms per fit | µs per pass | evals/s |")
print("|---|---|---|---|---|---|---|")
for n in (1000, 10_000, 100_000, 262_144, 1_000_000, 4_000_000):
    angles, x, _ = synth.make_single(2, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for method, name in ((0, "dif"), (1, "bc_dif")):
        for _ in range(2):
            r = brdf_amd.fit_single(method, 2, a, xd, synth.P0[2], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        torch.cuda.synchronize(); N = 10; t0 = time.perf_counter()
        for _ in range(N):
            r = brdf_amd.fit_single(method, 2, a, xd, synth.P0[2], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
        st = brdf_amd.last_fit_stats()
        print(f"| {n:,} | {name} | {'resident' if st['launches'] == 1 else 'chain'} | {st['passes']} | {dt*1e3:.3f} | {st['device_us']/st['passes']:.1f} | {r.info[7]*n/dt:.2e} |", flush=True)
