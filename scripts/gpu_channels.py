"""three colour channels over one set of planes: ONE shared resident launch (brdf_hip_fit_channels_dev) against three
single-fit launches, HIP events on the launch stream.  usage: python scripts/gpu_channels.py [reps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
from tests import oracle_libs as L
from tests.test_gpu_parity import _channel_measurements
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for model, n, balanced in ((2, 1_000_000, True), (1, 1_000_000, True), (2, 1_000_000, False), (1, 402_928, False))[:int(os.environ.get('CASES', '4'))]:
    angles, x0, _ = synth.make_single(model, n)
    if balanced:  # the benchmark's generator: one truth, three noise draws
        rng = np.random.default_rng(5)
        clean = L.model_values(model, angles, synth.TRUTH[model])
        xs = np.stack([x0] + [clean + 0.01 * (rng.random(n) - 0.5) for _ in range(2)])
    else:
        xs = _channel_measurements(model, angles, n)
    a = torch.from_numpy(np.ascontiguousarray(angles)).to(dev)
    xd = torch.from_numpy(np.ascontiguousarray(xs)).to(dev)
    kw = dict(lb=synth.LB, ub=synth.UB, itmax=synth.ITMAX, opts=synth.OPTS)
    for method in (1,):
        def shared():
            return brdf_amd.fit_channels(method, model, a, xd, synth.P0[model], **kw)
        def serial():
            return [brdf_amd.fit_single(method, model, a, xd[c], synth.P0[model], **kw) for c in range(3)]
        out = {}
        for name, fn in (("shared", shared), ("serial", serial)):
            for _ in range(3):
                r = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            for _ in range(reps):
                r = fn()
            e1.record(); torch.cuda.synchronize()
            out[name] = (e0.elapsed_time(e1) * 1e3 / reps, (time.perf_counter() - t0) * 1e6 / reps, r)
        st = brdf_amd.last_fit_stats()
        r = shared()
        cst = brdf_amd.last_channels_stats(3)["channels"]
        passes = [int(cst[c]["passes"]) for c in range(3)]
        per = [round(cst[c]["device_us"] / max(1, cst[c]["passes"]), 2) for c in range(3)]
        nfev = [float(r.info[7]) for r in out["shared"][2]]
        if os.environ.get("BRDF_HIP_LIB", "").find("stamps") >= 0:
            import ctypes as C
            for c in range(3):
                o8 = (C.c_longlong * 8)(); brdf_amd.lib.brdf_hip_last_channels_stamps(c, o8)
                P = max(1, passes[c])
                if c < 2:
                    print(f"    sweeping wave: {['sweep', 'served'][c]} {o8[6]} {['stage1', 'lifetime'][c]} {o8[7]}")
                print(f"    channel {c}: cycles/pass wait_workers {o8[1]/P:.0f} stage2 {o8[2]/P:.0f} exchange {o8[3]/P:.0f} step {o8[4]/P:.0f} uniforms {o8[5]/P:.0f}")
        print(f"model {model} n {n} method {method}: shared {out['shared'][0]:.1f} us (wall {out['shared'][1]:.1f}) serial {out['serial'][0]:.1f} us "
              f"(wall {out['serial'][1]:.1f}) speedup {out['serial'][0] / out['shared'][0]:.2f}x  nfev {nfev} passes {passes} us/pass per channel {per}", flush=True)
