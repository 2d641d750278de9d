#!/usr/bin/env python3
"""bench.py -- BRDF residual-evals/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

Workloads (config.workload):
  c2 (default, the benchmarked configuration)  BASELINE.json configs[1] -- a single-material Ward 3-parameter fit over
       1,000,000 synthetic (cos theta_i, cos theta_h, cos theta_o, measured) samples, fitted with the reference's
       dlevmar_dif entry point (levmar.h:112-115); one *step* = one complete fit with the samples already resident in
       HBM.  The same fit through dlevmar_bc_dif (the call the application makes, brdfdata.cpp:1119) is reported next
       to it under "bc_dif".
  c3   configs[2] -- the same with the reference's own Blinn-Phong model (brdfdata.cpp:983-987), 24 B per sample-pass.
  c4 / c5   configs[3] / [4] -- 65,536 surfels x 4,096 samples / 2^20 surfels x 256 samples (Ward), batched regime,
       surfels sharded contiguously over the ranks, ONE RCCL gather per step.

residual-evals = info[7] * n, levmar's own nfev accounting.

`python bench.py --gpus N` with N > 1 starts N ranks itself (torch.distributed.run on 127.0.0.1, one rank per GPU)
BEFORE anything touches the GPU and relays rank 0's line; under an external launcher (RANK / WORLD_SIZE set) it is one
of the ranks.  c2 / c3 with N > 1: every rank fits its own material (different seed) -- fits are independent, so there
is no data-path collective; the fitted parameters + info[] of all steps are gathered to rank 0 with ONE RCCL gather
at the end of the timed region ("weak" scaling).  c4 / c5: fixed total work ("strong").
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SAMPLES = 1_000_000  # BASELINE.json configs[1] / [2]
BYTES_PER_SAMPLE_PASS = {0: 24, 1: 24, 2: 32}  # 2 (Ward: 3) planes + measurement, fp64
MODEL_NAME = {0: "phong", 1: "blinn-phong", 2: "ward"}
P_TOL, E_TOL = 1e-5, 1e-8  # BASELINE.json north_star / SURVEY.md section 8d
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02_traffic.json")


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline legs (the reference's own levmar, oracle/_ref, or our restatement): never inside a timed GPU region
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_lib():
    ref_path = os.path.join(ROOT, "oracle", "_ref", "liblevmar_ref.so")
    if os.path.exists(ref_path):
        return C.CDLL(ref_path), "ref_brdf_fit", "reference"
    return C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_brdf_fit", "port"


def cpu_baseline(method, model, angles, x, p0, opts, lb, ub, itmax, budget_s=12.0):
    """The reference's own CPU levmar path (oracle/_ref, compiled from /root/reference in the dev container) or, if
    that library is absent, our CPU restatement -- timed on ONE host core (levmar is serial) on the same inputs."""
    lib, fn, kind = _cpu_lib()
    D = C.POINTER(C.c_double)
    flat = np.ascontiguousarray(angles.reshape(-1))
    o, l, u = (np.array(v, dtype=np.float64) for v in (opts, lb, ub))
    reps, evals, secs, p, info = 0, 0.0, 0.0, None, None
    while reps < 1 or (secs < budget_s and reps < 8):
        p = np.array(p0, dtype=np.float64)
        info = np.zeros(10)
        t0 = time.perf_counter()
        getattr(lib, fn)(method, model, flat.ctypes.data_as(D), x.ctypes.data_as(D), x.size, p.ctypes.data_as(D), itmax,
                         o.ctypes.data_as(D), l.ctypes.data_as(D), u.ctypes.data_as(D), info.ctypes.data_as(D))
        secs += time.perf_counter() - t0
        evals += info[7] * x.size
        reps += 1
    name = "dlevmar_dif" if method == 0 else "dlevmar_bc_dif"
    return {"value": evals / secs, "unit": "residual-evals/s", "cores": 1, "kind": kind,
            "sample": f"{reps} x the full {x.size}-sample {MODEL_NAME[model]} {name} fit ({secs:.1f} s of CPU time, gcc -O2)"}, p, info


def cpu_worker_main(argv):
    """one worker PROCESS of the all-core batched baseline (levmar is not re-entrant as shipped: static LU buffer,
    Axb_core.c:1142-1143 -- processes, not threads): fits surfels first, first+stride, ... until its time budget is
    spent; numpy + ctypes only, never the GPU.  Prints one JSON line."""
    method, model, n, first, stride, limit, budget = (int(argv[0]), int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4]),
                                                        int(argv[5]), float(argv[6]))
    from brdf_amd import synth  # (brdf_amd/__init__ is not imported with -c below: this pulls synth only via sys.path)
    lib, fn, kind = _cpu_lib()
    D = C.POINTER(C.c_double)
    lb, ub = synth.bounds(model)
    o, l, u = (np.array(v, dtype=np.float64) for v in (synth.OPTS, lb, ub))
    fits, evals, secs = 0, 0.0, 0.0
    s_ = first
    while s_ < limit and secs < budget:
        angles, x, _ = synth.make_surfels(model, n, first=s_, count=1)
        p = np.array(synth.P0[model], dtype=np.float64)
        info = np.zeros(10)
        a = np.ascontiguousarray(angles[0].reshape(-1))
        xs = np.ascontiguousarray(x[0])
        t0 = time.perf_counter()
        getattr(lib, fn)(method, model, a.ctypes.data_as(D), xs.ctypes.data_as(D), n, p.ctypes.data_as(D), synth.ITMAX,
                         o.ctypes.data_as(D), l.ctypes.data_as(D), u.ctypes.data_as(D), info.ctypes.data_as(D))
        secs += time.perf_counter() - t0
        evals += info[7] * n
        fits += 1
        s_ += stride
    print(json.dumps({"fits": fits, "evals": evals, "secs": secs, "kind": kind}), flush=True)


def cpu_baseline_batched(method, model, n, total, budget_s=10.0):
    """SURVEY.md section 8d / BASELINE.md section 3: the reference's CPU levmar on ALL host cores, one worker process
    per core, worker w fitting surfels w, w + cores, ... of the same workload for `budget_s` seconds each (a bounded
    sample of the S surfels).  Rate = fitted residual-evals / the slowest worker's busy time."""
    cores = host_cores()
    # the helper module synth.py is imported by file path so that the workers never load the HIP library or torch
    code = ("import sys, importlib.util, types; sys.path.insert(0, %r); "
            "pkg = types.ModuleType('brdf_amd'); pkg.__path__ = [%r]; sys.modules['brdf_amd'] = pkg; "
            "import bench; bench.cpu_worker_main(sys.argv[1:])") % (ROOT, os.path.join(ROOT, "brdf_amd"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(method), str(model), str(n), str(w), str(cores), str(total), str(budget_s)],
                              stdout=subprocess.PIPE, text=True, cwd=ROOT) for w in range(cores)]
    outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    fits = sum(o["fits"] for o in outs)
    evals = sum(o["evals"] for o in outs)
    busy = max(o["secs"] for o in outs)
    return {"value": evals / busy, "unit": "residual-evals/s", "cores": cores, "kind": outs[0]["kind"], "fits_per_s": fits / busy,
            "sample": f"{fits} of the workload's {total} surfels (surfel w, w+{cores}, ... on worker w), {cores} worker processes x "
                      f"{budget_s:.0f} s, one fit after the other per worker as in the reference's pixel loop (brdfdata.cpp:1195-1220), gcc -O2"}


# ---------------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------------
def host_cores():
    """worker processes of the all-core CPU baseline = the host cores this job may really use: the scheduler affinity, cut
    down by a cgroup CPU quota if there is one, by BRDF_BENCH_CPU_WORKERS if set, and by 16 per visible GPU (the CPU
    share that comes with one GPU of a shared node)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    if os.environ.get("BRDF_BENCH_CPU_WORKERS"):
        return max(1, min(cores, int(os.environ["BRDF_BENCH_CPU_WORKERS"])))
    try:
        import torch
        gpus = max(1, torch.cuda.device_count())  # (counting devices does not initialise the GPU)
    except Exception:  # noqa: BLE001
        gpus = 1
    return max(1, min(cores, 16 * gpus))


def source_hash():
    """sha256 over the kernel sources the library is built from: ties a profiled traffic figure to the code it was
    measured on (profiles/r02_traffic.json carries the hash of the sources that were profiled)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "brdf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary -- only if it was measured on the
    sources this library was built from; otherwise None (a stale constant is worse than no number)"""
    try:
        tj = json.load(open(TRAFFIC_FILE))
        if tj.get("source_hash") != source_hash():
            return None, f"{os.path.relpath(TRAFFIC_FILE, ROOT)} was measured on sources {tj.get('source_hash')}, this build is {source_hash()}"
        return tj["kernels"][kernel]["hbm_bytes_per_launch"], f"{os.path.relpath(TRAFFIC_FILE, ROOT)} ({tj.get('tag')}), sources {tj.get('source_hash')}"
    except (OSError, KeyError, ValueError) as exc:
        return None, f"no profiled traffic: {exc}"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) as CHILD processes before this process
    has touched the GPU (it never does) and pass their output through; rank 0 prints the JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, cwd=ROOT).returncode


FORCE_COLL = False


def dist_setup(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # rehearsal knobs (not used by the driver): BRDF_BENCH_BACKEND=gloo runs the N ranks' collectives on the CPU,
    # BRDF_BENCH_DEVICE=k puts every rank on GPU k, BRDF_BENCH_STUB=1 replaces the fit by a stub (no GPU at all: the
    # CPU test of this launcher)
    backend = os.environ.get("BRDF_BENCH_BACKEND", "nccl")
    stub = os.environ.get("BRDF_BENCH_STUB") == "1"
    local = int(os.environ.get("BRDF_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    # BRDF_BENCH_COLLECTIVES=1 (rehearsal): run every collective of the N > 1 path on a ONE-rank communicator as well, so
    # that the RCCL calls themselves (init, barrier, all_reduce, gather on device tensors) execute on a one-GPU box
    global FORCE_COLL
    FORCE_COLL = os.environ.get("BRDF_BENCH_COLLECTIVES") == "1"
    if world > 1 or FORCE_COLL:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout is for rank 0's ONE JSON line,
        # so file descriptor 1 points at stderr while the communicator is created (init + a first barrier)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()
            if not stub:
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return torch, dist, rank, world, backend, dev, stub


# ---------------------------------------------------------------------------------------------------------------------
# c4 / c5: multi-surfel workloads
# ---------------------------------------------------------------------------------------------------------------------
def main_batched(args):
    from brdf_amd import synth
    S, n = {"c4": (65536, 4096), "c5": (1 << 20, 256)}[args.workload]
    if args.surfels:
        S = args.surfels
    if args.samples:
        n = args.samples
    model, method = 2, (0 if args.entry == "dif" else 1)
    lb, ub = synth.bounds(model)
    cpu_base = None
    if (not args.no_cpu and int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("BRDF_BENCH_STUB") != "1"):
        # before this process touches the GPU: the workers are plain child processes (numpy + ctypes, no torch, no HIP)
        cpu_base = cpu_baseline_batched(method, model, n, S)
    torch, dist, rank, world, backend, dev, stub = dist_setup(args)
    from brdf_amd import dist as bdist
    first, count = bdist.shard_range(S, rank, world)
    if stub:
        a_np, x_np, _ = synth.make_surfels(model, n, first=first, count=count)
        angles, x = torch.from_numpy(a_np), torch.from_numpy(x_np)
        p0 = torch.from_numpy(np.tile(np.array(synth.P0[model]), (count, 1)))

        def fit_shard(a, xx, pp):  # plumbing stand-in: "fits" every surfel to a value derived from its data, nfev = 1
            info = torch.zeros((a.shape[0], 10), dtype=torch.float64)
            info[:, 7] = 1.0
            return pp + xx.mean(dim=1, keepdim=True), info, torch.zeros(a.shape[0], dtype=torch.int32)
    else:
        import brdf_amd
        angles, x, p0 = bdist.gpu_make_shard(model, n, dev)(first, count)

        def fit_shard(a, xx, pp):
            return brdf_amd.fit_batch(method, model, a, xx, pp.clone(), lb=lb, ub=ub, itmax=synth.ITMAX, opts=synth.OPTS)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def one_step():
        p, info, ret = fit_shard(angles, x, p0)
        rows = torch.cat([p, info, ret.to(p.dtype)[:, None]], dim=1)
        if backend != "nccl":
            rows = rows.cpu()
        return bdist.gather_results(rows, S, force=FORCE_COLL)

    for _ in range(args.warmup):
        one_step()
    if world > 1 or FORCE_COLL:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    if world > 1 or FORCE_COLL:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    wt = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1 or FORCE_COLL:
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
        # the gathered rows (p, info, ret of every surfel, in surfel order) as one digest: runs on 1, 2, 4, 8 ranks must agree bit for bit
        line["result_sha256"] = hashlib.sha256(out.cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
        if stub:
            line["stub_checksum"] = float(out[:, :3].sum().item())
        if cpu_base is not None:
            line["cpu_baseline"] = cpu_base
        print(json.dumps(line), flush=True)
    if world > 1 or FORCE_COLL:
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# c2 / c3: one large fit per GPU
# ---------------------------------------------------------------------------------------------------------------------
def main_single(args, model):
    from brdf_amd import synth
    rank0_cpu = {}
    rank_env = int(os.environ.get("RANK", "0"))
    p0, opts, lb, ub, itmax = synth.P0[model], synth.OPTS, synth.LB, synth.UB, synth.ITMAX
    angles, x, truth = synth.make_single(model, N_SAMPLES, seed=synth.SEED + 7919 * rank_env)
    if not args.no_cpu and rank_env == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # CPU legs first, before this process has touched the GPU (they are never inside a timed region either way)
        rank0_cpu["dif"] = cpu_baseline(0, model, angles, x, p0, opts, lb, ub, itmax)
        rank0_cpu["bc_dif"] = cpu_baseline(1, model, angles, x, p0, opts, lb, ub, itmax, budget_s=6.0)

    torch, dist, rank, world, backend, dev, stub = dist_setup(args)
    import brdf_amd
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    # each rank owns one material: same generator, different seed -> different planes and noise
    a_dev = torch.from_numpy(angles).to(dev)
    x_dev = torch.from_numpy(x).to(dev)

    def run(method, steps, warmup):
        for _ in range(warmup):
            brdf_amd.fit_single(method, model, a_dev, x_dev, p0, lb=lb, ub=ub, itmax=itmax, opts=opts)
        res_np = np.zeros((steps, 13))  # filled inside the timed loop (numpy: ~1 us per row; torch CPU indexing costs ~20 us)
        passes = jac = launches = 0
        dev_us = 0.0
        if world > 1 or FORCE_COLL:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # one more event pair around every fit: the kernel launches of ONE fit without the host gap between two fits
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        ev0.record()  # same (current) stream the C ABI launches the pass kernels on
        for s in range(steps):
            evs[s][0].record()
            r = brdf_amd.fit_single(method, model, a_dev, x_dev, p0, lb=lb, ub=ub, itmax=itmax, opts=opts)
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
        if world > 1 or FORCE_COLL:  # the one collective of the job: fitted parameters + info[] of every step -> rank 0
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
        if world > 1 or FORCE_COLL:
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
        if world > 1 or FORCE_COLL:
            dist.destroy_process_group()
        return 0

    bpsp = BYTES_PER_SAMPLE_PASS[model]

    def roofline_of(head, method):
        # The dominant kernel.  Resident regime (default when the fit fits the chip): ONE launch per fit, the launch performs
        # passes_per_fit sweeps over samples it read from HBM once.  Launch chain: one launch per sweep.
        resident = head["launches_per_step"] < 1.5
        kernel = f"resident_fit_kernel<{model}, {method}, true, false>" if resident else f"stream_pass<{model}, {method}, true>"
        traffic, traffic_src = profiled_traffic(kernel)
        sweeps_per_launch = head["passes_per_step"] if resident else 1.0
        bytes_per_launch = bpsp * N_SAMPLES * sweeps_per_launch
        achieved = bytes_per_launch / (head["avg_launch_us"] * 1e-6) / 1e9
        r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "traffic": traffic, "traffic_source": traffic_src, "kernel": "brdf::" + kernel, "regime": "resident" if resident else "launch chain",
             "algorithmic_bytes_per_launch": bytes_per_launch, "sweeps_per_launch": sweeps_per_launch,
             "avg_launch_us": head["avg_launch_us"], "region_us_per_launch": head["region_us_per_launch"],
             "avg_sweeping_launch_us": head["avg_sweeping_launch_us"], "launches_per_step": head["launches_per_step"]}
        if traffic is not None:  # what really crosses the HBM interface: the samples are read ONCE per resident fit
            r["hbm_measured_gbs"] = traffic / (head["avg_launch_us"] * 1e-6) / 1e9
        return r

    head = out["dif"]
    roof = roofline_of(head, 0)
    roof["note"] = (
        f"achieved = SURVEY.md section 8d's algorithmic bytes ({bpsp} B per sample per LM evaluation: "
        f"{'3' if model == 2 else '2'} planes + measurement, fp64) x 1e6 samples x the launch's evaluations (sweeps_per_launch) / the "
        "average HIP-event time of a fit's launch (one event pair per fit on the launch stream); in the "
        "resident regime one launch is a whole fit: the in-launch exchanges and serial LM steps are part of it.  region_us_per_launch "
        "adds the host's gap between two fits; avg_sweeping_launch_us = device clock per evaluation.  traffic = HBM-side bytes per "
        "launch from rocprofv3 FETCH_SIZE (x2 gfx950 correction, calibrated) + WRITE_SIZE, valid only for the sources named in "
        "traffic_source: the samples are read ONCE per fit, so the measured traffic (hbm_measured_gbs) is ~1/sweeps of the algorithmic "
        "figure -- the launch is bound by the exchange + LM step latency and by fp64 issue, not by HBM bandwidth (DESIGN.md section 4)")
    line = {
        "metric": f"BRDF residual-evals/sec (1 M samples, {MODEL_NAME[model].title()} 3-param), whole job; rel-err vs CPU levmar in `parity`",
        "value": head["value"], "unit": "residual-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{1 if model == 2 else 2}]: single-material {MODEL_NAME[model]} 3-param fit, 1,000,000 synthetic samples, "
                               f"dlevmar_dif (FD Jacobian + Broyden), p0={{{', '.join(str(v) for v in p0)}}}, opts={{1e-3,1e-15,1e-15,1e-20,1e-6}}, itmax=100; "
                               "one step = one complete fit, samples resident in HBM when the timed region starts; one material per GPU",
                   "n_samples": N_SAMPLES, "brdf": MODEL_NAME[model], "entry_point": "dlevmar_dif", "fits_per_step_per_gpu": 1,
                   "nfev_per_fit": head["nfev"], "lm_iterations": head["iters"], "passes_per_fit": head["passes_per_step"]},
        "roofline": roof,
        "fitted_params": head["p"], "sumsq": head["sumsq"],
        "bc_dif": {k: out["bc_dif"][k] for k in ("value", "ms_per_step", "nfev", "iters", "passes_per_step", "launches_per_step", "avg_launch_us",
                                                    "avg_sweeping_launch_us", "p")},
    }
    line["bc_dif"]["roofline"] = {k: v for k, v in roofline_of(out["bc_dif"], 1).items()
                                  if k in ("achieved", "frac", "traffic", "traffic_source", "kernel", "algorithmic_bytes_per_launch", "avg_launch_us")}
    rc = 0
    if rank0_cpu:
        base, p_cpu, info_cpu = rank0_cpu["dif"]
        line["cpu_baseline"] = base
        p_gpu = np.array(head["p"])
        line["parity"] = {"max_rel_err_params_vs_cpu_levmar": float(np.max(np.abs(p_gpu - p_cpu) / np.maximum(np.abs(p_cpu), 1e-12))),
                          "rel_err_sumsq": float(abs(head["sumsq"] - info_cpu[1]) / info_cpu[1]), "tolerance": P_TOL, "tolerance_sumsq": E_TOL}
        base_bc, p_cpu_bc, _ = rank0_cpu["bc_dif"]
        line["bc_dif"]["cpu_baseline"] = base_bc
        line["bc_dif"]["max_rel_err_params_vs_cpu_levmar"] = float(
            np.max(np.abs(np.array(out["bc_dif"]["p"]) - p_cpu_bc) / np.maximum(np.abs(p_cpu_bc), 1e-12)))
        ok = (line["parity"]["max_rel_err_params_vs_cpu_levmar"] <= P_TOL and line["parity"]["rel_err_sumsq"] <= E_TOL
              and line["bc_dif"]["max_rel_err_params_vs_cpu_levmar"] <= P_TOL)
        line["parity"]["ok"] = bool(ok)
        if not ok:  # a fast number with a wrong answer is not a result
            print("bench.py: PARITY FAILURE against the CPU levmar path", file=sys.stderr)
            rc = 3
    print(json.dumps(line), flush=True)
    if world > 1 or FORCE_COLL:
        dist.destroy_process_group()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default, the benchmarked configuration): BASELINE.json configs[1] (Ward); c3: configs[2] (Blinn-Phong); "
                         "c4 / c5: configs[3] / [4], the multi-surfel configurations, surfels sharded over the ranks (strong scaling)")
    ap.add_argument("--entry", default="dif", choices=["dif", "bc_dif"], help="entry point for c4 / c5")
    ap.add_argument("--surfels", type=int, default=0, help=argparse.SUPPRESS)  # rehearsal / tests: shrink c4 / c5
    ap.add_argument("--samples", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    if args.workload in ("c4", "c5"):
        return main_batched(args)
    return main_single(args, 2 if args.workload == "c2" else 1)


if __name__ == "__main__":
    sys.exit(main())
