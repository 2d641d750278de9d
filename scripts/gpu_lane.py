"""n = 16 dlevmar_bc_dif (the application's call, brdfdata.cpp:1119): the lane-per-fit kernel (lane_fit.hip) against
the wave-per-fit / four-fits-per-wave kernels it replaces, and its agreement with the CPU oracle fit by fit.
usage: python scripts/gpu_lane.py [S_log2=20]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
from brdf_amd._lib import lib
from tests import oracle_libs as L
from concurrent.futures import ThreadPoolExecutor

dev = torch.device("cuda:0")
S = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
n = 16


def gen(model, S, n):
    truth = torch.from_numpy(synth.surfel_truth(model, 0, S)).to(dev)
    a = torch.empty((S, 3, n), dtype=torch.float64, device=dev)
    x = torch.empty((S, n), dtype=torch.float64, device=dev)
    assert lib.brdf_hip_synth_dev(model, synth.SEED, 0, S, n, truth.data_ptr(), a.data_ptr(), x.data_ptr(), None) == 0
    torch.cuda.synchronize()
    return a, x


def run(model, method, a, x, reps=2):
    lb, ub = synth.bounds(model)
    best = 1e9
    for _ in range(reps):
        p0 = torch.from_numpy(np.tile(np.array(synth.P0[model]), (a.shape[0], 1))).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p, info, ret = brdf_amd.fit_batch(method, model, a, x, p0, lb=lb, ub=ub, itmax=100, opts=synth.OPTS)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, p, info, ret


out = {}
for model in (1, 2):
    a, x = gen(model, S, n)
    lb, ub = synth.bounds(model)
    # oracle on the first K fits
    K = 2048
    ah, xh = a[:K].cpu().numpy(), x[:K].cpu().numpy()

    def one(k):
        return L.brdf_fit("orc", 1, model, ah[k], xh[k], synth.P0[model], synth.ITMAX, synth.OPTS, lb, ub)

    with ThreadPoolExecutor(8) as ex:
        ref = list(ex.map(one, range(K)))
    variants = [("wave_per_fit", {"BRDF_HIP_LANE": "0"})]
    for w in ("1", "2", "4"):
        variants.append((f"lane_w{w}", {"BRDF_HIP_LANE": "1", "BRDF_HIP_LANE_WAVES": w}))
    for q, mw in ((1, 0), (8, 3), (16, 6), (24, 6), (32, 6), (32, 12), (40, 8)):
        variants.append((f"lane_w1_q{q}_mw{mw}", {"BRDF_HIP_LANE": "1", "BRDF_HIP_LANE_WAVES": "1", "BRDF_HIP_LANE_QUORUM": str(q),
                                                    "BRDF_HIP_LANE_MAXWAIT": str(mw)}))
    if model != 2:
        variants.append(("lane_w1_exact_pow", {"BRDF_HIP_LANE": "1", "BRDF_HIP_LANE_WAVES": "1", "BRDF_HIP_EXACT_POW": "1"}))
    for name, env in variants:
        for k in ("BRDF_HIP_LANE", "BRDF_HIP_LANE_WAVES", "BRDF_HIP_EXACT_POW", "BRDF_HIP_LANE_QUORUM", "BRDF_HIP_LANE_MAXWAIT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        dt, p, info, ret = run(model, 1, a, x)
        ph, ih, rh = p[:K].cpu().numpy(), info[:K].cpu().numpy(), ret[:K].cpu().numpy()
        same_traj = close_p = obj_ok = both = 0
        worst_obj = 0.0
        for k, (r, p_ref, info_ref) in enumerate(ref):
            if r < 0 or rh[k] < 0:
                continue
            both += 1
            same_traj += int(np.array_equal(ih[k, 5:8], info_ref[5:8]))
            close_p += int(L.rel_err(ph[k], p_ref) <= 1e-5)
            rel = (ih[k, 1] - info_ref[1]) / max(info_ref[1], 1e-300)
            obj_ok += int(rel <= 1e-6)
            worst_obj = max(worst_obj, rel)
        res = {"seconds": dt, "fits_per_s": S / dt, "mean_nfev": float(info[:, 7].mean().item()), "mean_iters": float(info[:, 5].mean().item()),
               "failed": int((ret < 0).sum().item()), "oracle_subset": both, "same_iters_reason_nfev": same_traj, "params_within_1e-5": close_p,
               "objective_not_worse_1e-6": obj_ok, "worst_objective_excess": worst_obj}
        out[f"model{model}_{name}"] = res
        print(f"model={model} {name}: {json.dumps(res)}", flush=True)
    del a, x
    torch.cuda.empty_cache()
print(json.dumps(out))
