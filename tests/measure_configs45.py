"""BASELINE.json configs 4 and 5 at FULL size on one GPU (batched regime): S x n synthetic surfels generated on the
device, fitted, parity-checked on a fixed subset against the CPU oracle, timed.  Used under rocprofv3 for the
HBM-read evidence (samples are read from HBM exactly once per fit)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth, dist as bdist
dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")); D = C.POINTER(C.c_double)
which = sys.argv[1:] or ["c4", "c5"]
cfg = {"c4": (65536, 4096), "c5": (1 << 20, 256)}
model = 2
out = {}
for name in which:
    S, n = cfg[name]
    make = bdist.gpu_make_shard(model, n, dev)
    t0 = time.perf_counter(); angles, x, p0 = make(0, S); torch.cuda.synchronize(); gen_s = time.perf_counter() - t0
    lb, ub = synth.bounds(model)
    res = {}
    for method, mname in ((0, "dif"), (1, "bc_dif")):
        p_in = p0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p, info, ret = brdf_amd.fit_batch(method, model, angles, x, p_in, lb=lb, ub=ub, itmax=synth.ITMAX, opts=synth.OPTS)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nfev = info[:, 7].sum().item()
        # parity on a fixed subset of surfels against the CPU oracle
        idx = np.linspace(0, S - 1, 64).astype(int)
        worst_p = worst_e = 0.0; checked = 0
        opts = np.array(synth.OPTS); lba, uba = np.array(lb), np.array(ub)
        for s in idx:
            a_h = np.ascontiguousarray(angles[s].cpu().numpy().reshape(-1)); x_h = np.ascontiguousarray(x[s].cpu().numpy())
            pp = np.array(synth.P0[model]); ii = np.zeros(10)
            r = orc.orc_brdf_fit(method, model, a_h.ctypes.data_as(D), x_h.ctypes.data_as(D), n, pp.ctypes.data_as(D), synth.ITMAX,
                                 opts.ctypes.data_as(D), lba.ctypes.data_as(D), uba.ctypes.data_as(D), ii.ctypes.data_as(D))
            if r < 0 or ii[6] == 3 or ret[s].item() < 0: continue
            checked += 1
            worst_p = max(worst_p, float(np.max(np.abs(p[s].cpu().numpy() - pp) / np.maximum(np.abs(pp), 1e-12))))
            worst_e = max(worst_e, abs(info[s, 1].item() - ii[1]) / ii[1])
        res[mname] = {"seconds": dt, "fits_per_s": S / dt, "residual_evals_per_s": nfev * n / dt, "mean_nfev": nfev / S,
                      "failed_fits": int((ret < 0).sum().item()), "parity_subset": checked, "max_rel_err_params": worst_p,
                      "max_rel_err_sumsq": worst_e, "min_traffic_bytes": S * n * 32 + S * 104}
        print(name, mname, json.dumps(res[mname]), flush=True)
    out[name] = {"S": S, "n": n, "model": "ward", "generate_seconds": gen_s, **res}
    del angles, x, p0
    torch.cuda.empty_cache()
print(json.dumps(out))
