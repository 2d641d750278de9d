"""Known-byte-count launches with the pass kernels' own access pattern (8 B per lane, coalesced, fp64 planes):
calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for this pattern (MI355X_MICROARCH.md section HBM says
FETCH_SIZE reads 1/2 of the bytes of a 16 B/lane stream and that other widths must be calibrated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brdf_amd
dev = torch.device("cuda:0")
n = 64 * 1024 * 1024  # 3 planes x 512 MiB, far beyond the 256 MiB Infinity Cache
angles = torch.rand((3, n), dtype=torch.float64, device=dev) * 0.9 + 0.05
for model in (1, 2):  # Blinn-Phong reads 2 planes (1 GiB) + writes 512 MiB; Ward reads 3 planes (1.5 GiB) + writes 512 MiB
    for _ in range(3):
        hx = brdf_amd.model_eval(model, angles, (0.35, 0.6, 24.0) if model == 1 else (0.35, 0.25, 0.15))
torch.cuda.synchronize()
print("calibration launches done: model_eval_kernel<1> reads", 2 * n * 8, "B writes", n * 8, "B; <2> reads", 3 * n * 8, "B")
