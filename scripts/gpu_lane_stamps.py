"""Where a wave of the lane-per-fit kernel spends its cycles (diagnostic build: make -C brdf_amd/csrc lane_variant
SUFFIX=_lst EXTRA=-DBRDF_LANE_STAMPS; run with BRDF_HIP_LIB=brdf_amd/libbrdf_hip_lst.so).  usage: gpu_lane_stamps.py [S_log2=20]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
from brdf_amd._lib import lib

dev = torch.device("cuda:0")
S = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
n = 16
raw = C.CDLL(brdf_amd.LIB_PATH)
names = ["bookkeeping", "refill", "sweeps(light)", "sweeps(heavy)", "steps(light)", "steps(heavy)", "-", "light rounds", "heavy rounds",
         "evaluating lanes in light rounds", "evaluating lanes in heavy rounds", "lanes wanting heavy", "total cycles"]
for model in (1, 2):
    truth = torch.from_numpy(synth.surfel_truth(model, 0, S)).to(dev)
    a = torch.empty((S, 3, n), dtype=torch.float64, device=dev)
    x = torch.empty((S, n), dtype=torch.float64, device=dev)
    assert lib.brdf_hip_synth_dev(model, synth.SEED, 0, S, n, truth.data_ptr(), a.data_ptr(), x.data_ptr(), None) == 0
    lb, ub = synth.bounds(model)
    for rep in range(2):
        p0 = torch.from_numpy(np.tile(np.array(synth.P0[model]), (S, 1))).to(dev)
        torch.cuda.synchronize()
        out = (C.c_longlong * 48)()
        raw.brdf_hip_lane_stamps(out)  # clear
        t0 = time.perf_counter()
        p, info, ret = brdf_amd.fit_batch(1, model, a, x, p0, lb=lb, ub=ub, itmax=100, opts=synth.OPTS)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    raw.brdf_hip_lane_stamps(out)
    v = list(out)
    tot = v[12]
    print(f"model {model}: {S / dt:.3e} fits/s, {dt * 1e3:.1f} ms, mean nfev {info[:, 7].mean().item():.0f}, mean iters {info[:, 5].mean().item():.1f}")
    for i in (0, 1, 2, 3, 4, 5):
        print(f"   {names[i]:16s} {100.0 * v[i] / tot:5.1f} % of wave cycles")
    lr, hr = max(v[7], 1), max(v[8], 1)
    print(f"   rounds: {v[7]} light ({v[2] / lr:.0f} sweep + {v[4] / lr:.0f} step cycles each, {v[9] / lr:.1f} lanes evaluating), "
          f"{v[8]} heavy ({v[1] / hr:.0f} refill + {v[3] / hr:.0f} sweep + {v[5] / hr:.0f} step cycles each, {v[10] / hr:.1f} lanes evaluating)")
    print(f"   lanes wanting a heavy round, average over all rounds: {v[11] / (lr + hr):.1f}")
    ph = ["outside", "B_INIT_EVAL", "B_ITER_TOP", "B_AFTER_JAC", "B_SOLVE", "B_AFTER_LM_EVAL", "B_AFTER_LM_NORM", "B_LM_JUDGE", "B_LS_PROLOGUE",
          "B_LS_ISSUE", "B_LS_EVAL", "B_PG_BEGIN", "B_PG_ISSUE", "B_PG_EVAL", "B_PG_NORM", "B_PG_JUDGE", "B_PG_MULTI", "B_COMMIT", "B_END_ITER",
          "B_FINISH", "B_DONE"] + ["-"] * 8 + ["dispatch trip start", "step entry (heavy round)", "step entry (light round)"]
    pv = v[16:48]
    print("   phases of BcMachine::run, % of wave cycles: " + ", ".join(f"{ph[i]} {100.0 * pv[i] / tot:.1f}" for i in range(32) if pv[i] > 0.002 * tot))
