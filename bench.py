#!/usr/bin/env python3
"""bench.py -- BRDF residual-evals/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

Workload (config.workload): BASELINE.json configs[1] -- a single-material Ward 3-parameter fit over
1,000,000 synthetic (cos theta_i, cos theta_h, cos theta_o, measured) samples, fitted with the
reference's dlevmar_dif entry point (levmar.h:112-115); one *step* = one complete fit with the samples
already resident in HBM.  The same fit through dlevmar_bc_dif (the call the application makes,
brdfdata.cpp:1119) is reported next to it under "bc_dif".

residual-evals = info[7] * n, levmar's own nfev accounting.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank fits its own material (different
seed) -- fits are independent, so there is no data-path collective; the fitted parameters + info[] of all
steps are gathered to rank 0 with ONE RCCL gather at the end of the timed region.  scaling = "weak".
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MODEL, N_SAMPLES = 2, 1_000_000  # Ward, BASELINE.json configs[1]
BYTES_PER_SAMPLE_PASS = {0: 24, 1: 24, 2: 32}  # 2 (Ward: 3) planes + measurement, fp64


def cpu_baseline(method, angles, x, p0, opts, lb, ub, itmax, budget_s=12.0):
    """The reference's own CPU levmar path (oracle/_ref, compiled from /root/reference in the dev container)
    or, if that library is absent, our CPU restatement -- timed on ONE host core on the same inputs."""
    ref_path = os.path.join(ROOT, "oracle", "_ref", "liblevmar_ref.so")
    if os.path.exists(ref_path):
        lib, fn, kind = C.CDLL(ref_path), "ref_brdf_fit", "reference"
    else:
        lib, fn, kind = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_brdf_fit", "port"
    D = C.POINTER(C.c_double)
    flat = np.ascontiguousarray(angles.reshape(-1))
    o, l, u = (np.array(v, dtype=np.float64) for v in (opts, lb, ub))
    reps, evals, secs, p, info = 0, 0.0, 0.0, None, None
    while reps < 1 or (secs < budget_s and reps < 8):
        p = np.array(p0, dtype=np.float64)
        info = np.zeros(10)
        t0 = time.perf_counter()
        getattr(lib, fn)(method, MODEL, flat.ctypes.data_as(D), x.ctypes.data_as(D), x.size, p.ctypes.data_as(D), itmax,
                         o.ctypes.data_as(D), l.ctypes.data_as(D), u.ctypes.data_as(D), info.ctypes.data_as(D))
        secs += time.perf_counter() - t0
        evals += info[7] * x.size
        reps += 1
    return {"value": evals / secs, "unit": "residual-evals/s", "cores": 1, "kind": kind,
            "sample": f"{reps} x the full {x.size}-sample Ward {'dlevmar_dif' if method == 0 else 'dlevmar_bc_dif'} fit "
                      f"({secs:.1f} s of CPU time, gcc -O2)"}, p, info


def cpu_baseline_batched(method, n, count, budget_s=10.0):
    """the reference's CPU levmar (oracle/_ref, or our restatement) on the first `count` surfels of the multi-surfel
    workload, one after the other on ONE host core, as the reference's pixel loop does (brdfdata.cpp:1195-1220)"""
    from brdf_amd import synth
    ref_path = os.path.join(ROOT, "oracle", "_ref", "liblevmar_ref.so")
    if os.path.exists(ref_path):
        lib, fn, kind = C.CDLL(ref_path), "ref_brdf_fit", "reference"
    else:
        lib, fn, kind = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_brdf_fit", "port"
    D = C.POINTER(C.c_double)
    angles, x, _ = synth.make_surfels(MODEL, n, first=0, count=count)
    lb, ub = synth.bounds(MODEL)
    o, l, u = (np.array(v, dtype=np.float64) for v in (synth.OPTS, lb, ub))
    fits, evals, secs = 0, 0.0, 0.0
    for s_ in range(count):
        p = np.array(synth.P0[MODEL], dtype=np.float64)
        info = np.zeros(10)
        a = np.ascontiguousarray(angles[s_].reshape(-1))
        xs = np.ascontiguousarray(x[s_])
        t0 = time.perf_counter()
        getattr(lib, fn)(method, MODEL, a.ctypes.data_as(D), xs.ctypes.data_as(D), n, p.ctypes.data_as(D), synth.ITMAX,
                         o.ctypes.data_as(D), l.ctypes.data_as(D), u.ctypes.data_as(D), info.ctypes.data_as(D))
        secs += time.perf_counter() - t0
        evals += info[7] * n
        fits += 1
        if secs > budget_s:
            break
    return {"value": evals / secs, "unit": "residual-evals/s", "cores": 1, "kind": kind, "fits_per_s": fits / secs,
            "sample": f"the first {fits} surfels of the workload, one fit after the other ({secs:.1f} s of CPU time, gcc -O2)"}


def main_batched(args):
    """BASELINE.json configs[3] (65,536 surfels x 4,096 samples) / configs[4] (2^20 x 256), Ward, batched regime:
    every rank generates (on the device) and fits only its own contiguous surfel range; ONE RCCL gather of the fitted
    parameters + info[] + return codes at the end of each step (brdf_amd/dist.py).  Total work is fixed: strong scaling."""
    import torch
    import torch.distributed as dist

    import brdf_amd
    from brdf_amd import dist as bdist
    from brdf_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("BRDF_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("BRDF_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    S, n = {"c4": (65536, 4096), "c5": (1 << 20, 256)}[args.workload]
    model, method = 2, (0 if args.entry == "dif" else 1)
    lb, ub = synth.bounds(model)
    first, count = bdist.shard_range(S, rank, world)
    angles, x, p0 = bdist.gpu_make_shard(model, n, dev)(first, count)

    def fit_shard(a, xx, pp):
        return brdf_amd.fit_batch(method, model, a, xx, pp.clone(), lb=lb, ub=ub, itmax=synth.ITMAX, opts=synth.OPTS)

    def one_step():
        p, info, ret = fit_shard(angles, x, p0)
        rows = torch.cat([p, info, ret.to(p.dtype)[:, None]], dim=1)
        if backend != "nccl":
            rows = rows.cpu()
        return bdist.gather_results(rows, S)

    for _ in range(args.warmup):
        one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wt = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    if rank == 0:
        wall = float(wt.item())
        nfev = float(out[:, 3 + 7].sum().item())
        failed = int((out[:, 13] < 0).sum().item())
        min_traffic = S * n * 32 + S * 104
        line = {"metric": "BRDF residual-evals/sec (Ward 3-param, multi-surfel), whole job", "value": nfev * n * args.steps / wall,
                "unit": "residual-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"BASELINE.json configs[{3 if args.workload == 'c4' else 4}]: {S} independent surfels x {n} samples, "
                                       f"Ward fit, dlevmar_{args.entry}, surfels sharded contiguously over the ranks, one RCCL gather per step",
                           "surfels": S, "samples_per_surfel": n, "entry_point": "dlevmar_" + args.entry, "fits_per_s": S * args.steps / wall,
                           "mean_nfev": nfev / S, "failed_fits": failed},
                "roofline": {"bound": "hbm", "achieved": min_traffic * args.steps / wall / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": min_traffic * args.steps / wall / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "note": "batched regime: the samples are read from HBM once per fit (min_traffic = 32 B x n x S + 104 B per fit) "
                                     "and the LM iterations run out of registers; the kernel is bound by fp64 VALU issue and the serial LM step, "
                                     "not by HBM (DESIGN.md section 4)"}}
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline_batched(method, n, 768 if n > 1024 else 4096)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 (default, the benchmarked configuration): BASELINE.json configs[1]; c4 / c5: configs[3] / [4], "
                         "the multi-surfel configurations, surfels sharded over the ranks (strong scaling)")
    ap.add_argument("--entry", default="dif", choices=["dif", "bc_dif"], help="entry point for c4 / c5")
    args = ap.parse_args()
    if args.workload != "c2":
        return main_batched(args)

    import torch
    import torch.distributed as dist

    import brdf_amd
    from brdf_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal knobs (not used by the driver): run N ranks on ONE GPU over gloo to exercise the N>1 code path
    backend = os.environ.get("BRDF_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("BRDF_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    # each rank owns one material: same generator, different seed -> different planes and noise
    angles, x, truth = synth.make_single(MODEL, N_SAMPLES, seed=synth.SEED + 7919 * rank)
    a_dev = torch.from_numpy(angles).to(dev)
    x_dev = torch.from_numpy(x).to(dev)
    p0, opts, lb, ub, itmax = synth.P0[MODEL], synth.OPTS, synth.LB, synth.UB, synth.ITMAX

    def run(method, steps, warmup):
        for _ in range(warmup):
            brdf_amd.fit_single(method, MODEL, a_dev, x_dev, p0, lb=lb, ub=ub, itmax=itmax, opts=opts)
        res_np = np.zeros((steps, 13))  # filled inside the timed loop (numpy: ~1 us per row; torch CPU indexing costs ~20 us)
        passes = jac = launches = 0
        dev_us = 0.0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # one more event pair around every fit: the kernel launches of ONE fit without the host gap between two fits
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        ev0.record()  # same (current) stream the C ABI launches the pass kernels on
        for s in range(steps):
            evs[s][0].record()
            r = brdf_amd.fit_single(method, MODEL, a_dev, x_dev, p0, lb=lb, ub=ub, itmax=itmax, opts=opts)
            evs[s][1].record()
            if r.ret < 0:
                raise RuntimeError(f"fit failed: {brdf_amd.last_error()}")
            res_np[s, :3] = r.p
            res_np[s, 3:] = r.info
            st = brdf_amd.last_fit_stats()
            passes += st["passes"]
            launches += st["launches"]
            jac += st["jac_passes"]
            dev_us += st["device_us"]
        ev1.record()
        results = torch.from_numpy(res_np)
        gathered = None
        if world > 1:  # the one collective of the job: fitted parameters + info[] of every step -> rank 0
            res_dev = results.to(coll_dev)
            gathered = [torch.empty_like(res_dev) for _ in range(world)] if rank == 0 else None
            dist.gather(res_dev, gathered, dst=0)
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
        fit_ms = sum(a.elapsed_time(b) for a, b in evs)
        stat = torch.tensor([wall, float(results[:, 10].sum().item()) * N_SAMPLES, ev_ms, float(passes), float(jac), dev_us,
                             float(launches), fit_ms],
                            dtype=torch.float64, device=coll_dev)
        if world > 1:
            allstat = [torch.empty_like(stat) for _ in range(world)]
            dist.all_gather(allstat, stat)
            allstat = torch.stack(allstat).cpu()
        else:
            allstat = stat.cpu()[None, :]
        if rank == 0 and gathered is not None:
            results = torch.stack([g.cpu() for g in gathered])  # [world, steps, 13]
        else:
            results = results[None]
        return allstat, results

    out = {}
    for method, name in ((0, "dif"), (1, "bc_dif")):
        allstat, results = run(method, args.steps, args.warmup)
        wall = float(allstat[:, 0].max())  # max over ranks
        evals = float(allstat[:, 1].sum())  # whole job
        r0 = results[0]
        out[name] = {
            "value": evals / wall, "ms_per_step": 1e3 * wall / args.steps, "evals_per_step_rank0": float(r0[0, 10]) * N_SAMPLES,
            "nfev": float(r0[0, 10]), "iters": float(r0[0, 8]), "passes_per_step": float(allstat[0, 3]) / args.steps,
            "event_ms_rank0": float(allstat[0, 2]), "device_us_per_step": float(allstat[0, 5]) / args.steps,
            "p": [float(v) for v in r0[0, :3]], "sumsq": float(r0[0, 4]),
            # HIP-event time of the fits (one event pair per fit, summed over the timed region) / ALL launches of the
            # pass kernel in them (launch chain: the passes plus the few run-ahead launches per fit that find it finished):
            # the population rocprofv3 averages over.  region_us_per_launch = the whole timed region / launches, i.e.
            # with the host's gaps between two fits
            "avg_launch_us": 1e3 * float(allstat[0, 7]) / max(1.0, float(allstat[0, 6])),
            "region_us_per_launch": 1e3 * float(allstat[0, 2]) / max(1.0, float(allstat[0, 6])),
            "launches_per_step": float(allstat[0, 6]) / args.steps,
            # device clock (s_memrealtime) from the first to the finishing pass / passes: sweeping launches only
            "avg_sweeping_launch_us": float(allstat[0, 5]) / max(1.0, float(allstat[0, 3])),
        }

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    head = out["dif"]
    # The dominant kernel.  Resident regime (default for dlevmar_dif when the fit fits the chip): ONE launch per fit, the
    # launch performs passes_per_fit sweeps over samples it read from HBM once.  Launch chain: one launch per sweep.
    resident = head["launches_per_step"] < 1.5
    kernel = "resident_fit_kernel<2, 0, true, false>" if resident else "stream_pass<2, 0, true>"
    # HBM traffic per launch: measured with rocprofv3 PMC counters in separate profiling passes of this very
    # command (scripts/profile_round.sh), corrected as MI355X_MICROARCH.md prescribes; committed under profiles/
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        traffic = tj["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    sweeps_per_launch = head["passes_per_step"] if resident else 1.0
    bytes_per_launch = BYTES_PER_SAMPLE_PASS[MODEL] * N_SAMPLES * sweeps_per_launch
    achieved = bytes_per_launch / (head["avg_launch_us"] * 1e-6) / 1e9
    if resident:
        kernel_desc = ("brdf::resident_fit_kernel<2,0,true,false> (one launch per fit: samples + secant Jacobian resident in registers/LDS, "
                       "model eval + residual + Broyden + JtJ/Jte fused per LM evaluation, in-launch all-gather between evaluations)")
        note = ("algorithmic bytes = 32 B per sample per LM evaluation (Ward: 3 planes + measurement, fp64) x 1e6 samples x the "
                "launch's evaluations (sweeps_per_launch); avg launch = HIP-event time around each fit's launch (its 4 KB upload "
                "included), averaged over the timed region's fits -- in-launch exchanges and serial LM steps are part of the launch; "
                "region_us_per_launch adds the host's gap between two fits; avg_sweeping_launch_us = device clock "
                "per evaluation.  traffic = HBM-side bytes per launch from rocprofv3 FETCH_SIZE (x2 gfx950 correction, calibrated) + "
                "WRITE_SIZE, profiles/r01_traffic.json: the samples are read ONCE per fit, so the measured traffic is ~1/sweeps of the "
                "algorithmic bytes -- the launch is bound by the exchange + LM step latency and fp64 issue, not by HBM (DESIGN.md)")
    else:
        kernel_desc = "brdf::stream_pass<2,0,true> (fused model eval + residual + Broyden + JtJ/Jte sweep, one launch per LM evaluation)"
        note = ("avg launch = HIP-event time of the fits / launches of the pass kernel in them (passes + run-ahead "
                "launches that return at once; includes inter-launch gaps, the per-fit upload and the in-kernel LM step); "
                "avg_sweeping_launch_us = device clock over the passes only; 32 B per sample-pass for Ward (3 planes + measurement, fp64); traffic = "
                "HBM-side bytes per launch from rocprofv3 FETCH_SIZE (x2 gfx950 correction, calibrated) + WRITE_SIZE, "
                "profiles/r01_traffic.json: the dif trial pass really moves 80 B/sample (the secant Jacobian is "
                "read and rewritten), the bc_dif pass moves exactly the algorithmic 32 B/sample")
    line = {
        "metric": "BRDF residual-evals/sec (1 M samples, Ward 3-param), whole job; rel-err vs CPU levmar in `parity`", "value": head["value"], "unit": "residual-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: single-material Ward 3-param fit, 1,000,000 synthetic samples, "
                               "dlevmar_dif (FD Jacobian + Broyden), p0={0.5,0.5,0.3}, opts={1e-3,1e-15,1e-15,1e-20,1e-6}, itmax=100; "
                               "one step = one complete fit, samples resident in HBM when the timed region starts; one material per GPU",
                   "n_samples": N_SAMPLES, "brdf": "ward", "entry_point": "dlevmar_dif", "fits_per_step_per_gpu": 1,
                   "nfev_per_fit": head["nfev"], "lm_iterations": head["iters"], "passes_per_fit": head["passes_per_step"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "kernel": kernel_desc, "regime": "resident" if resident else "launch chain",
                     "algorithmic_bytes_per_launch": bytes_per_launch, "sweeps_per_launch": sweeps_per_launch,
                     "avg_launch_us": head["avg_launch_us"], "region_us_per_launch": head["region_us_per_launch"],
                     "avg_sweeping_launch_us": head["avg_sweeping_launch_us"],
                     "launches_per_step": head["launches_per_step"], "note": note},
        "fitted_params": head["p"], "sumsq": head["sumsq"],
        "bc_dif": {k: out["bc_dif"][k] for k in ("value", "ms_per_step", "nfev", "iters", "passes_per_step", "launches_per_step", "avg_launch_us",
                                                    "avg_sweeping_launch_us", "p")},
    }
    if not args.no_cpu:
        base, p_cpu, info_cpu = cpu_baseline(0, angles, x, p0, opts, lb, ub, itmax)
        line["cpu_baseline"] = base
        p_gpu = np.array(head["p"])
        line["parity"] = {"max_rel_err_params_vs_cpu_levmar": float(np.max(np.abs(p_gpu - p_cpu) / np.maximum(np.abs(p_cpu), 1e-12))),
                          "rel_err_sumsq": float(abs(head["sumsq"] - info_cpu[1]) / info_cpu[1]), "tolerance": 1e-5}
        base_bc, p_cpu_bc, _ = cpu_baseline(1, angles, x, p0, opts, lb, ub, itmax, budget_s=6.0)
        line["bc_dif"]["cpu_baseline"] = base_bc
        line["bc_dif"]["max_rel_err_params_vs_cpu_levmar"] = float(
            np.max(np.abs(np.array(out["bc_dif"]["p"]) - p_cpu_bc) / np.maximum(np.abs(p_cpu_bc), 1e-12)))
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
