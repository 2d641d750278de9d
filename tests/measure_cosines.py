"""measurement of the vectors -> cosines kernel: S surfels x 16 lights, HIP events on the launch
stream; algorithmic bytes per surfel = 4 (surfel index) + 12 (face) + 72 (three gathered vertices) + 24 (normal) in,
3 x 16 x 8 = 384 out = 496 B.  The CPU figure is the C restatement (oracle/cosines_oracle.c) on one core."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from tests import oracle_libs as L
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
nv, nf, S = 250_000, 500_000, 1 << 20
vertices = rng.uniform(-80, 80, size=(nv, 3))
faces = rng.integers(0, nv, size=(nf, 3)).astype(np.int32)
nrm = rng.normal(size=(nf, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
surfels = rng.integers(0, nf, size=S).astype(np.int32)
view = np.array([310.0, -75.0, 700.0]); leds = brdf_amd.led_table()
tv, tf, tn, ts = (torch.from_numpy(a).to(dev) for a in (vertices, faces, nrm, surfels))
for _ in range(3):
    out = brdf_amd.cosines(tv, tf, tn, leds, view, surfels=ts, validate=False)
torch.cuda.synchronize()
K = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    out = brdf_amd.cosines(tv, tf, tn, leds, view, surfels=ts, validate=False)
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / K
# the same surfels in face order (a real pixel map is spatially coherent: neighbouring pixels see the same or adjacent faces;
# the random draw above makes every surfel touch ~5 cold 128-byte lines for its 112 gathered bytes)
ts_sorted = torch.sort(ts).values
for _ in range(3):
    out_sorted = brdf_amd.cosines(tv, tf, tn, leds, view, surfels=ts_sorted, validate=False)
torch.cuda.synchronize()
e0.record()
for _ in range(K):
    out_sorted = brdf_amd.cosines(tv, tf, tn, leds, view, surfels=ts_sorted, validate=False)
e1.record(); torch.cuda.synchronize()
us_sorted = 1e3 * e0.elapsed_time(e1) / K
sub = 1 << 16
t0 = time.perf_counter(); ref = L.cosines(vertices, faces, nrm, leds, view, surfels=surfels[:sub]); cpu_s = time.perf_counter() - t0
assert np.array_equal(out[:sub].cpu().numpy(), ref)
print(json.dumps({"kernel": "cosines_rows_kernel", "surfels": S, "lights": 16, "us_per_launch": us, "surfels_per_s": S / (us * 1e-6),
                  "roofline": {"bound": "hbm", "achieved": 496 * S / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": 496 * S / (us * 1e-6) / 1e9 / 8000.0, "algorithmic_bytes_per_surfel": 496},
                  "cpu_baseline": {"value": sub / cpu_s, "unit": "surfels/s", "cores": 1, "kind": "port", "sample": f"{sub} surfels"},
                  "surfels_in_face_order": {"us_per_launch": us_sorted, "achieved": 496 * S / (us_sorted * 1e-6) / 1e9, "frac": 496 * S / (us_sorted * 1e-6) / 1e9 / 8000.0},
                  "bit_exact_vs_oracle_on_sample": True}))
