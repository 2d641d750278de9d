import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
dev = torch.device("cuda:0")
names = ["load", "fold", "step", "uniforms", "persist", "sweep", "reduce"]           # launch chain (stream_fit.hip)
names_res = ["-", "stage2+X2", "publish+level1", "level2+fold", "build+B", "own_sweep", "stage1+X1wait", "step"]   # resident regime, control wave view (resident_fit.hip)
cases = [(2, 4096), (2, 1_000_000)] if os.environ.get("BRDF_STAMPS_WARD_ONLY") else [(2, 4096), (2, 1_000_000), (1, 1_000_000)]
for model, n in cases:
    angles, x, _ = synth.make_single(model, n)
    a = torch.from_numpy(angles).to(dev); xd = torch.from_numpy(x).to(dev)
    for method in (0, 1):
        r = brdf_amd.fit_single(method, model, a, xd, synth.P0[model], lb=synth.LB, ub=synth.UB, itmax=100, opts=synth.OPTS)
        st = brdf_amd.last_fit_stats()
        out = (C.c_longlong * 8)(); brdf_amd.lib.brdf_hip_last_fit_stamps(out)
        P = max(1, st['passes'] - 1)
        print(model, n, method, r.ret, 'us/pass %.2f' % (st['device_us'] / st['passes']),
              'launches', st['launches'], ' '.join(f"{nm}={out[k]/P:.0f}" for k, nm in enumerate(names_res if st['launches'] == 1 else names + ["-"])), 'cycles/pass', flush=True)
        if st['launches'] == 1 and n > 100000:  # one epoch's timeline of every workgroup (s_memrealtime, 10 ns ticks)
            tr = (C.c_longlong * (8 * 257))(); rows = brdf_amd.lib.brdf_hip_last_fit_trace(tr, 257)
            t = np.array(tr[:], dtype=np.int64).reshape(257, 8)
            print('    LM step sections (cycles/pass, LM_STAMP 1..7):', ' '.join(f'{v / P:.0f}' for v in t[256, 1:]), flush=True)
            t = t[:256]
            if t[:, 0].min() > 0:
                t0 = t[:, 0].min()
                lead = t[:, 6] > 0
                for k, nm in enumerate(["pass start", "own sweep done", "reduced (X2)", "published (+level 1 for leaders)", "level 2 gathered", "stepped"]):
                    v = (t[:, k] - t0) * 0.01
                    print(f"    trace {nm:34s} us after the first workgroup's pass start: min {v.min():6.2f} median {np.median(v):6.2f} max {v.max():6.2f}"
                          + (f" | leaders: min {v[lead].min():6.2f} max {v[lead].max():6.2f}" if k == 3 and lead.any() else ""), flush=True)
                if os.environ.get("BRDF_TRACE_WORKERS"):
                    for k, nm in ((6, "wave 4 (SIMD of the control wave) sweep done"), (7, "wave 7 sweep done")):
                        v = (t[:, k] - t0) * 0.01
                        print(f"    trace {nm:34s} us after the first workgroup's pass start: min {v.min():6.2f} median {np.median(v):6.2f} max {v.max():6.2f}", flush=True)
                    continue
                print(f"    trace polls: level 1 (leaders) {sorted(t[lead, 6].tolist())}  level 2 min {t[:, 7].min()} median {int(np.median(t[:, 7]))} max {t[:, 7].max()}", flush=True)
