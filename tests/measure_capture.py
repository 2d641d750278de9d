"""measurement of the capture loop: a synthetic 1024 x 1024 capture of 16 images, ~70 % of the
pixels on the mesh, three dlevmar_bc_dif fits of 16 samples per pixel; CPU figure: the C restatement on a sample."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
from brdf_amd import synth
from tests import oracle_libs as L
dev = torch.device("cuda:0")
rng = np.random.default_rng(2)
H = W = 1024; nv, nf = 20000, 40000
vertices = rng.uniform(-80, 80, size=(nv, 3)) + np.array([0.0, -80.0, 60.0])
faces = np.stack([rng.integers(0, nv, size=nf), rng.integers(0, nv, size=nf), rng.integers(0, nv, size=nf)], axis=1).astype(np.int32)
e1 = vertices[faces[:, 1]] - vertices[faces[:, 0]]; e2 = vertices[faces[:, 2]] - vertices[faces[:, 0]]
nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
leds = brdf_amd.led_table(); view = np.array([310.0, -75.0, 700.0])
c = vertices[faces].sum(axis=1) / 3.0
nrm[((leds.mean(axis=0)[None, :] - c) * nrm).sum(axis=1) < 0] *= -1.0
pixel_map = rng.integers(0, nf, size=(H, W)).astype(np.int32); pixel_map[rng.random((H, W)) < 0.3] = -1
# images: per-pixel Blinn-Phong radiance through the device cosines, quantised to 8 bit
tv, tf, tn = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (vertices, faces, nrm))
ang = torch.abs(brdf_amd.cosines(tv, tf, tn, leds, view, rv_mode=1))                     # [nf,3,16]
pm = torch.from_numpy(pixel_map).to(dev)
a_px = ang[pm.clamp(min=0).long()]                                                       # [H,W,3,16]
kd, ks, n = synth.TRUTH[1]
val = kd * a_px[:, :, 0, :] + ks * torch.pow(a_px[:, :, 1, :], n)                        # [H,W,16]
img = torch.zeros((16, H, W, 3), dtype=torch.uint8, device=dev)
for ch in range(3):
    q = torch.clamp(torch.round(val * (0.6 + 0.2 * ch) * 127.0), 0, 255).to(torch.uint8)  # [H,W,16]
    img[:, :, :, ch] = torch.flip(q.permute(2, 0, 1), dims=[1])                          # image row = H-1-y
opts = (1e-3, 1e-15, 1e-15, 1e-20, 1e-6)
surf, avg, npx = brdf_amd.fit_capture(1, img, pm, tv, tf, tn, leds, view, rv_mode=1, opts=opts)   # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
surf, avg, npx = brdf_amd.fit_capture(1, img, pm, tv, tf, tn, leds, view, rv_mode=1, opts=opts)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
# CPU: the restatement on a 48 x 48 crop
crop = 48
t0 = time.perf_counter()
_, _, npx_c = L.fit_capture(1, img[:, H - crop:, :crop, :].cpu().numpy(), pixel_map[:crop, :crop], vertices, faces, nrm, leds, view, rv_mode=1, opts=opts)
cpu_s = time.perf_counter() - t0
print(json.dumps({"call": "brdf_hip_fit_capture_dev", "image": [H, W], "lights": 16, "pixels_fitted": npx, "fits": 3 * npx,
                  "seconds": dt, "fits_per_s": 3 * npx / dt, "avg": [float(v) for v in avg],
                  "cpu_baseline": {"value": 3 * npx_c / cpu_s, "unit": "fits/s", "cores": 1, "kind": "port", "sample": f"{crop}x{crop} crop, {npx_c} pixels"}}))
