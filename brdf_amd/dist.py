"""Surfel-sharded multi-GPU driver: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

Fits are independent (brdfdata.cpp:1195-1220 carries nothing between iterations of the pixel loop), so the
data path has NO collective: rank r generates/loads and fits only its own contiguous surfel range.  The one
collective of a job is the final gather of the fitted parameters + info[] to rank 0 (S x 14 doubles in total:
65,536 surfels -> 6.8 MB, i.e. < 1 MB per peer over seven point-to-point xGMI links; a ring would be per-link
bound and is unnecessary).
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous range [first, first+count) of rank `rank`: ceil(total/world) per rank, last ranks may be short
    (or empty).  SURVEY.md section 8e."""
    per = -(-total // world)
    first = min(total, rank * per)
    return first, min(per, total - first)


def gather_results(local: "torch.Tensor", total: int, group=None, force: bool = False):
    """Gather per-rank result rows [count_r, K] to rank 0 as one [total, K] tensor (None on other ranks).

    Shards may be ragged (the last rank short/empty): every rank pads to ceil(total/world) rows, one
    `dist.gather` moves them, rank 0 drops the padding.  `force`: take the collective path on a ONE-rank communicator
    too (bench.py's rehearsal of the RCCL calls on a one-GPU box); by default a single rank returns its rows untouched.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-total // world)
    k = local.shape[1]
    padded = torch.zeros((per, k), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    bucket = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, bucket, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat(bucket, dim=0)[:total]


def fit_sharded(method: int, model: int, total: int, n: int, make_shard, fit_shard, group=None):
    """Generic driver: `make_shard(first, count)` -> (angles, x, p0) for this rank's surfels,
    `fit_shard(angles, x, p0)` -> (p [count,3], info [count,10], ret [count]) tensors.  Returns on rank 0 the
    gathered (p, info, ret) for all `total` surfels in surfel order; None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = shard_range(total, rank, world)
    if count > 0:
        angles, x, p0 = make_shard(first, count)
        p, info, ret = fit_shard(angles, x, p0)
        rows = torch.cat([p, info, ret.to(p.dtype)[:, None]], dim=1)
    else:
        dev = "cuda" if (dist.is_initialized() and dist.get_backend(group) == "nccl") else "cpu"
        rows = torch.zeros((0, 14), dtype=torch.float64, device=dev)
    out = gather_results(rows, total, group)
    if out is None:
        return None
    return out[:, :3], out[:, 3:13], out[:, 13].to(torch.int32)


def gpu_make_shard(model: int, n: int, device, seed: int | None = None):
    """make_shard for the synthetic configurations: planes/measurements are generated on the device by
    brdf_hip_synth_dev for exactly this rank's surfels (counter-based stream: no rank ever touches another
    rank's data)."""
    import torch
    from . import synth
    from ._lib import lib, last_error
    seed = synth.SEED if seed is None else seed

    def make(first: int, count: int):
        truth = torch.from_numpy(np.ascontiguousarray(synth.surfel_truth(model, first, count, seed))).to(device)
        angles = torch.empty((count, 3, n), dtype=torch.float64, device=device)
        x = torch.empty((count, n), dtype=torch.float64, device=device)
        with torch.cuda.device(device):
            rc = lib.brdf_hip_synth_dev(model, seed, first, count, n, truth.data_ptr(), angles.data_ptr(), x.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(last_error())
        p0 = torch.from_numpy(np.tile(np.array(synth.P0[model]), (count, 1))).to(device)
        return angles, x, p0

    return make
