import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
path = os.environ["BRDF_HIP_STEP_DUMP"]
if os.path.exists(path): os.remove(path)
for model, n in [(2, 100000), (1, 100000)]:
    angles, x, _ = synth.make_single(model, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for method in (0, 1):
        brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
agg = collections.defaultdict(list)
for line in open(path):
    m, cyc, kinds, phases = map(int, line.split())
    agg[(m, kinds, phases)].append(cyc)
names = {0:'DONE',1:'EVAL',2:'SCALED',3:'JAC',4:'DIF_INIT',5:'DIF_JAC',6:'DIF_TRIAL',7:'DIF_UPDATE',8:'EVAL_MULTI'}
for (m, kinds, phases), v in sorted(agg.items(), key=lambda kv: -len(kv[1])):
    print(f"method={m} {names[kinds//100]:>9s}->{names[kinds%100]:<9s} phase {phases//100:2d}->{phases%100:2d}  n={len(v):5d}  mean={np.mean(v):7.0f}  min={min(v):6d} max={max(v):6d} cycles")
