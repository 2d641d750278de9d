"""EVERY surfel of a multi-surfel configuration against the CPU oracle ("... plus all surfels
against the build's CPU restatement"): the device fits all S surfels in one call, the planes are copied back and the
oracle (C restatement, bit-identical to the compiled reference on the fixtures) refits every one of them on the host
cores.  usage: gpu_all_surfels.py <c4|c5> <dif|bc_dif> [surfels]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import brdf_amd
from brdf_amd import synth, dist as bdist
from tests import oracle_libs as L

cfg, entry = sys.argv[1], sys.argv[2]
S, n = {"c4": (65536, 4096), "c5": (1 << 20, 256)}[cfg]
if len(sys.argv) > 3:
    S = int(sys.argv[3])
method, model = (0 if entry == "dif" else 1), 2
dev = torch.device("cuda:0")
angles, x, p0 = bdist.gpu_make_shard(model, n, dev)(0, S)
lb, ub = synth.bounds(model)
t0 = time.perf_counter()
p, info, ret = brdf_amd.fit_batch(method, model, angles, x, p0.clone(), lb=lb, ub=ub, itmax=synth.ITMAX, opts=synth.OPTS)
torch.cuda.synchronize()
gpu_s = time.perf_counter() - t0
p, info, ret = p.cpu().numpy(), info.cpu().numpy(), ret.cpu().numpy()
a_h, x_h = angles.cpu().numpy(), x.cpu().numpy()
del angles, x
workers = min(len(os.sched_getaffinity(0)), 16)
p_ref = np.tile(np.array(synth.P0[model]), (S, 1))
info_ref = np.zeros((S, 10))
ret_ref = np.zeros(S, dtype=np.int32)
opts, lba, uba = np.array(synth.OPTS), np.array(lb), np.array(ub)
fn = L.orc.orc_brdf_fit_batch
fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
bounds = np.linspace(0, S, 8 * workers + 1).astype(np.int64)

def run(k):
    a, b = int(bounds[k]), int(bounds[k + 1])
    fn(method, model, a_h[a:b].ctypes.data, x_h[a:b].ctypes.data, b - a, n, p_ref[a:b].ctypes.data, synth.ITMAX, opts.ctypes.data,
       lba.ctypes.data, uba.ctypes.data, info_ref[a:b].ctypes.data, ret_ref[a:b].ctypes.data)

t0 = time.perf_counter()
with ThreadPoolExecutor(workers) as ex:
    list(ex.map(run, range(len(bounds) - 1)))
cpu_s = time.perf_counter() - t0
pg, pr = p.copy(), p_ref.copy()
if method == 0:  # Ward depends on alpha^2: +alpha and -alpha are the same minimiser for the unconstrained fit
    pg[:, 2], pr[:, 2] = np.abs(pg[:, 2]), np.abs(pr[:, 2])
rel_p = np.max(np.abs(pg - pr) / np.maximum(np.abs(pr), 1e-12), axis=1)
rel_e = np.abs(info[:, 1] - info_ref[:, 1]) / info_ref[:, 1]
both_conv = (info[:, 6] != 3) & (info_ref[:, 6] != 3) & (ret >= 0) & (ret_ref >= 0)
ok = both_conv & (rel_p <= 1e-5) & (rel_e <= 1e-8)
out = {"config": cfg, "entry": entry, "surfels": S, "samples": n, "gpu_seconds": gpu_s, "oracle_seconds": cpu_s, "oracle_threads": workers,
       "failed_gpu": int((ret < 0).sum()), "failed_oracle": int((ret_ref < 0).sum()),
       "itmax_gpu": int((info[:, 6] == 3).sum()), "itmax_oracle": int((info_ref[:, 6] == 3).sum()),
       "converged_on_both": int(both_conv.sum()), "within_1e-5_and_1e-8": int(ok.sum()),
       "converged_on_both_but_outside": int((both_conv & ~ok).sum()),
       "max_rel_err_params_converged": float(rel_p[both_conv].max()), "max_rel_err_sumsq_converged": float(rel_e[both_conv].max()),
       "objective_no_worse_than_oracle_1e-6": int((info[:, 1] <= info_ref[:, 1] * (1 + 1e-6)).sum()),
       "worst_objective_excess": float(((info[:, 1] - info_ref[:, 1]) / info_ref[:, 1]).max()),
       "oracle_worse_than_gpu_1e-6": int((info_ref[:, 1] > info[:, 1] * (1 + 1e-6)).sum()),
       "worst_objective_excess_of_the_oracle": float(((info_ref[:, 1] - info[:, 1]) / info[:, 1]).max()),
       "identical_iteration_count": int((info[:, 5] == info_ref[:, 5]).sum())}
out["mean_iterations"] = [float(info[:, 5].mean()), float(info_ref[:, 5].mean())]
out["mean_nfev"] = [float(info[:, 7].mean()), float(info_ref[:, 7].mean())]
gi = np.nonzero((info[:, 6] == 3) & (info_ref[:, 6] != 3))[0]
out["gpu_itmax_only"] = {"count": int(gi.size), "oracle_iterations_mean": float(info_ref[gi, 5].mean()) if gi.size else 0.0,
                         "oracle_reasons": np.unique(info_ref[gi, 6], return_counts=True)[1].tolist() if gi.size else [],
                         "gpu_nfev_mean": float(info[gi, 7].mean()) if gi.size else 0.0, "oracle_nfev_mean": float(info_ref[gi, 7].mean()) if gi.size else 0.0,
                         "examples": [{"surfel": int(k), "p": p[k].tolist(), "p_oracle": p_ref[k].tolist(), "info": info[k].tolist(), "info_oracle": info_ref[k].tolist()} for k in gi[:3]]}
w = int(np.argmax((info[:, 1] - info_ref[:, 1]) / info_ref[:, 1]))
out["worst_surfel"] = {"surfel": w, "p": p[w].tolist(), "p_oracle": p_ref[w].tolist(), "info": info[w].tolist(), "info_oracle": info_ref[w].tolist(),
                       "truth": synth.surfel_truth(model, w, 1)[0].tolist()}
bad = np.nonzero(both_conv & ~ok)[0][:5]
out["examples_outside"] = [{"surfel": int(s), "p": p[s].tolist(), "p_oracle": p_ref[s].tolist(), "sumsq": info[s, 1], "sumsq_oracle": info_ref[s, 1],
                            "reason": info[s, 6], "reason_oracle": info_ref[s, 6]} for s in bad]
print(json.dumps(out))
