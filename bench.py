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
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_traffic.json")  # rocprofv3 PMC records of scripts/profile_round.sh (traffic and VALU instructions)
VALU_PEAK_LANE_INSTR_S = 256 * 4 * 16 * 2.4e9  # fp64 VALU issue: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz = 3.93e13 lane-instructions/s


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
    return {"value": evals / busy, "unit": "residual-evals/s", "cores": cores, "nproc": os.cpu_count(), "kind": outs[0]["kind"], "fits_per_s": fits / busy,
            "cores_note": "worker processes = min(cores this job may use, 16 per visible GPU: the CPU share of one GPU of a shared node); "
                          "BRDF_BENCH_CPU_WORKERS=all lifts the cap to every usable core, BRDF_BENCH_CPU_WORKERS=k sets it",
            "sample": f"{fits} of the workload's {total} surfels (surfel w, w+{cores}, ... on worker w), {cores} worker processes x "
                      f"{budget_s:.0f} s, one fit after the other per worker as in the reference's pixel loop (brdfdata.cpp:1195-1220), gcc -O2"}


# ---------------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------------
def host_cores():
    """worker processes of the all-core CPU baseline = the host cores this job may really use: the scheduler affinity, cut
    down by a cgroup CPU quota if there is one, by BRDF_BENCH_CPU_WORKERS if set (`all` = no further cap), and otherwise by
    16 per visible GPU (the CPU share that comes with one GPU of a shared node)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    want = os.environ.get("BRDF_BENCH_CPU_WORKERS")
    if want == "all":  # BASELINE.md section 3's literal "all host cores": every core the job may use, no per-GPU share
        return max(1, cores)
    if want:
        return max(1, min(cores, int(want)))
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


def alu_roofline(kernel, evals_per_s, evals_per_launch):
    """the roof the resident / batched kernels really sit under: fp64 VALU issue.  valu_lane_instr_per_eval = rocprofv3
    SQ_INSTS_VALU of one launch of `kernel` (wave instructions, x 64 lanes; every VALU instruction of the launch: sweeps,
    reductions, the LM steps) / the residual evaluations of that launch, measured on the sources this library was built
    from (else None); achieved = evals/s x that; peak = 3.93e13 lane-instr/s, the issue rate of fp64 instructions (32-bit
    VALU instructions issue twice as fast, so the fraction is an upper bound of the issue time in use)"""
    out = {"valu_lane_instr_per_eval": None, "achieved_instr_per_s": None, "peak": VALU_PEAK_LANE_INSTR_S, "unit": "lane-instr/s", "frac": None}
    try:
        tj = json.load(open(TRAFFIC_FILE))
        if tj.get("source_hash") != source_hash():
            out["source"] = f"{os.path.relpath(TRAFFIC_FILE, ROOT)} was measured on sources {tj.get('source_hash')}, this build is {source_hash()}"
            return out
        per_launch = tj["kernels"][kernel]["valu_wave_instr_per_launch"] * 64.0
        out["valu_lane_instr_per_eval"] = per_launch / evals_per_launch
        out["achieved_instr_per_s"] = evals_per_s * out["valu_lane_instr_per_eval"]
        out["frac"] = out["achieved_instr_per_s"] / VALU_PEAK_LANE_INSTR_S
        out["source"] = f"rocprofv3 --pmc SQ_INSTS_VALU, {os.path.relpath(TRAFFIC_FILE, ROOT)} ({tj.get('tag')}), sources {tj.get('source_hash')}"
    except (OSError, KeyError, ValueError, TypeError) as exc:
        out["source"] = f"no profiled instruction count: {exc}"
    return out


def batched_kernel_name(model, method, n):
    """the kernel brdf_hip_fit_batch_dev runs for fits of n samples (batch_fit.hip: geometry by fit size), as rocprofv3 names it"""
    m = {0: 0, 1: 1, 2: 1, 3: 2}[method]
    if n <= 16:
        return f"lane_fit_kernel<{model}, true, 1>" if method == 1 else f"rows_fit_kernel<{model}, {m}, true>"
    if n <= 256:
        return f"batch_fit_kernel<{model}, {m}, true, 64, 4>"
    if n <= 1024:
        return f"batch_fit_kernel<{model}, {m}, true, 256, 4>"
    return f"resident_fit_kernel<{model}, {m}, true, true>"


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
def batched_cpu_subset(workload, entry, count=64):
    """parity leg of a multi-surfel line: the CPU oracle's fits of the first `count` surfels (numpy + ctypes, before the
    process touches the GPU); returns (p[count,3], info[count,10], ret[count])"""
    from brdf_amd import synth
    S, n = {"c4": (65536, 4096), "c5": (1 << 20, 256)}[workload]
    model, method = 2, (0 if entry == "dif" else 1)
    lib, fn, kind = _cpu_lib()
    D = C.POINTER(C.c_double)
    lb, ub = synth.bounds(model)
    o, l, u = (np.array(v, dtype=np.float64) for v in (synth.OPTS, lb, ub))
    angles, x, _ = synth.make_surfels(model, n, first=0, count=count)
    ps, infos, rets = np.zeros((count, 3)), np.zeros((count, 10)), np.zeros(count, dtype=np.int64)
    for s_ in range(count):
        p = np.array(synth.P0[model], dtype=np.float64)
        a = np.ascontiguousarray(angles[s_].reshape(-1))
        xs = np.ascontiguousarray(x[s_])
        rets[s_] = getattr(lib, fn)(method, model, a.ctypes.data_as(D), xs.ctypes.data_as(D), n, p.ctypes.data_as(D), synth.ITMAX,
                                    o.ctypes.data_as(D), l.ctypes.data_as(D), u.ctypes.data_as(D), infos[s_].ctypes.data_as(D))
        ps[s_] = p
    return {"p": ps, "info": infos, "ret": rets, "kind": kind}


def batched_line(args, workload, entry, ctx, steps, warmup, cpu_base=None, cpu_subset=None):
    """one multi-surfel workload (c4 / c5) on the ranks of `ctx`; returns rank 0's line (None elsewhere)"""
    from brdf_amd import synth
    torch, dist, rank, world, backend, dev, stub = ctx
    S, n = {"c4": (65536, 4096), "c5": (1 << 20, 256)}[workload]
    if args.surfels:
        S = args.surfels
    if args.samples:
        n = args.samples
    model, method = 2, (0 if entry == "dif" else 1)
    lb, ub = synth.bounds(model)
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

    for _ in range(warmup):
        one_step()
    if world > 1 or FORCE_COLL:
        dist.barrier()
    sync()
    ev = None
    if not stub:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step()
    if ev:
        ev[1].record()
    if world > 1 or FORCE_COLL:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    wt = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1 or FORCE_COLL:
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    line = None
    if rank == 0:
        wall = float(wt.item())
        nfev = float(out[:, 3 + 7].sum().item())
        failed = int((out[:, 13] < 0).sum().item())
        min_traffic = S * n * 32 + S * 104
        evals_per_s = nfev * n * steps / wall
        kernel = batched_kernel_name(model, method, n)
        line = {"metric": "BRDF residual-evals/sec (Ward 3-param, multi-surfel), whole job", "value": evals_per_s,
                "unit": "residual-evals/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": 1e3 * wall / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"BASELINE.json configs[{3 if workload == 'c4' else 4}]: {S} independent surfels x {n} samples, "
                                       f"Ward fit, dlevmar_{entry}, surfels sharded contiguously over the ranks, one RCCL gather per step",
                           "surfels": S, "samples_per_surfel": n, "entry_point": "dlevmar_" + entry, "fits_per_s": S * steps / wall,
                           "mean_nfev": nfev / S, "failed_fits": failed},
                "roofline": {"bound": "fp64-valu", "kernel": "brdf::" + kernel,
                             "alu": alu_roofline(kernel, evals_per_s / world, nfev * n / world),
                             "hbm": {"achieved": min_traffic * steps / wall / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": min_traffic * steps / wall / 1e9 / HBM_PEAK_GBS, "min_traffic_bytes_per_step": min_traffic},
                             "traffic": None,
                             "note": "batched regime: the samples are read from HBM once per fit (hbm.min_traffic = 32 B x n x S + 104 B per fit) "
                                     "and the LM iterations run out of registers, so the HBM fraction is small by construction; the binding roof is "
                                     "fp64 VALU issue (alu: lane-instructions per second against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz) and the "
                                     "serial LM step of one wave per fit (DESIGN.md section 4)"}}
        r = line["roofline"]
        r["achieved"], r["peak"], r["unit"], r["frac"] = (r["alu"]["achieved_instr_per_s"], r["alu"]["peak"], "lane-instr/s", r["alu"]["frac"]) \
            if r["alu"]["frac"] is not None else (r["hbm"]["achieved"], HBM_PEAK_GBS, "GB/s", r["hbm"]["frac"])
        if r["alu"]["frac"] is None:
            r["bound"] = "hbm"
        if ev:
            line["event_ms_per_step"] = ev[0].elapsed_time(ev[1]) / steps
        # the gathered rows (p, info, ret of every surfel, in surfel order) as one digest: runs on 1, 2, 4, 8 ranks must agree bit for bit
        line["result_sha256"] = hashlib.sha256(out.cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
        if stub:
            line["stub_checksum"] = float(out[:, :3].sum().item())
        if cpu_base is not None:
            line["cpu_baseline"] = cpu_base
        if cpu_subset is not None:  # parity on the surfels the CPU leg fitted: SURVEY.md section 8d's bar on those that converge on both sides
            k = cpu_subset["p"].shape[0]
            got = out[:k].cpu().numpy()
            both = (cpu_subset["ret"] >= 0) & (got[:, 13] >= 0) & (cpu_subset["info"][:, 6] != 3) & (got[:, 3 + 6] != 3)
            pc, pg = cpu_subset["p"][both], got[both, :3]
            if method == 0:  # the unconstrained fit may land on -alpha (the model depends on alpha^2): compare |alpha|
                pc, pg = np.abs(pc), np.abs(pg)
            rel = np.max(np.abs(pg - pc) / np.maximum(np.abs(pc), 1e-12), axis=1)
            rel_e = np.abs(got[both, 3 + 1] - cpu_subset["info"][both, 1]) / cpu_subset["info"][both, 1]
            line["parity"] = {"surfels_compared": int(both.sum()), "of": int(k), "cpu": cpu_subset["kind"],
                              "max_rel_err_params_vs_cpu_levmar": float(rel.max()) if rel.size else None,
                              "max_rel_err_sumsq": float(rel_e.max()) if rel_e.size else None, "tolerance": P_TOL, "tolerance_sumsq": E_TOL,
                              "ok": bool(rel.size > 0 and rel.max() <= P_TOL and rel_e.max() <= E_TOL)}
    del angles, x, p0
    if not stub:
        torch.cuda.empty_cache()
    return line


def main_batched(args):
    cpu_base = None
    method = 0 if args.entry == "dif" else 1
    S, n = {"c4": (65536, 4096), "c5": (1 << 20, 256)}[args.workload]
    cpu_subset = None
    if (not args.no_cpu and int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("BRDF_BENCH_STUB") != "1"):
        # before this process touches the GPU: the workers are plain child processes (numpy + ctypes, no torch, no HIP)
        cpu_base = cpu_baseline_batched(method, 2, args.samples or n, args.surfels or S)
        if not args.surfels and not args.samples:
            cpu_subset = batched_cpu_subset(args.workload, args.entry)
    ctx = dist_setup(args)
    torch, dist, rank, world = ctx[0], ctx[1], ctx[2], ctx[3]
    line = batched_line(args, args.workload, args.entry, ctx, args.steps, args.warmup, cpu_base, cpu_subset)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1 or FORCE_COLL:
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# c2 / c3: one large fit per GPU
# ---------------------------------------------------------------------------------------------------------------------
def single_cpu_refs(model, angles, x, timed=True):
    """CPU legs of a single-fit line (the reference's levmar on one host core), before the process touches the GPU.  timed:
    the cpu_baseline of the headline (10-20 s); otherwise one fit per entry point, for the parity figures only"""
    from brdf_amd import synth
    p0, opts, lb, ub, itmax = synth.P0[model], synth.OPTS, synth.LB, synth.UB, synth.ITMAX
    return {"dif": cpu_baseline(0, model, angles, x, p0, opts, lb, ub, itmax, budget_s=12.0 if timed else 0.0),
            "bc_dif": cpu_baseline(1, model, angles, x, p0, opts, lb, ub, itmax, budget_s=6.0 if timed else 0.0)}


def single_line(args, model, ctx, steps, warmup, angles, x, rank0_cpu):
    """one single-material workload (c2 / c3): every rank fits its own material `steps` times through both entry points;
    returns (rank 0's line or None, exit code)"""
    from brdf_amd import synth
    torch, dist, rank, world, backend, dev, stub = ctx
    p0, opts, lb, ub, itmax = synth.P0[model], synth.OPTS, synth.LB, synth.UB, synth.ITMAX
    import brdf_amd
    brdf_amd.set_launch_timing(True)  # (the roofline's kernel duration: an event pair around every resident launch)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    # each rank owns one material: same generator, different seed -> different planes and noise
    a_dev = torch.from_numpy(angles).to(dev)
    x_dev = torch.from_numpy(x).to(dev)

    def run(method, steps, warmup):
        for _ in range(warmup):
            brdf_amd.fit_single(method, model, a_dev, x_dev, p0, lb=lb, ub=ub, itmax=itmax, opts=opts)
        res_np = np.zeros((steps, 13))  # filled inside the timed loop (numpy: ~1 us per row; torch CPU indexing costs ~20 us)
        passes = jac = launches = 0
        dev_us = kern_us = 0.0
        kern_n = 0
        if world > 1 or FORCE_COLL:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # The kernel launches of ONE fit without the host gap between two fits: in the resident regime the library's own event
        # pair around the launch (launch timing, read back through last_fit_stats()); for a fit that runs as a chain of launches,
        # an event pair around the call
        lib_timer = warmup > 0 and brdf_amd.last_fit_stats()["kernel_us"] >= 0.0
        evs = [] if lib_timer else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        ev0.record()  # same (current) stream the C ABI launches the pass kernels on
        for s in range(steps):
            if not lib_timer:
                evs[s][0].record()
            r = brdf_amd.fit_single(method, model, a_dev, x_dev, p0, lb=lb, ub=ub, itmax=itmax, opts=opts)
            if not lib_timer:
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
            if st["kernel_us"] >= 0.0:  # resident regime: the event pair the library recorded around this fit's launch
                kern_us += st["kernel_us"]
                kern_n += 1
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
        fit_ms = sum(a.elapsed_time(b) for a, b in evs) if evs else ev_ms
        stat = torch.tensor([wall, float(results[:, 10].sum().item()) * N_SAMPLES, ev_ms, float(passes), float(jac), dev_us,
                             float(launches), fit_ms, kern_us, float(kern_n)],
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
        allstat, results = run(method, steps, warmup)
        wall = float(allstat[:, 0].max())  # max over ranks
        evals = float(allstat[:, 1].sum())  # whole job
        r0 = results[0]
        nfev, njev = float(r0[0, 10]), float(r0[0, 11])
        # the streaming visits of the samples the REFERENCE makes for this fit (N_jac_passes + N_trial_passes):
        # one per Jacobian (levmar counts it as m, bc: m + 1, evaluations in nfev) and one per other evaluation
        ref_passes = njev + (nfev - (3.0 if method == 0 else 4.0) * njev)
        out[name] = {
            "value": evals / wall, "ms_per_step": 1e3 * wall / steps, "evals_per_step_rank0": nfev * N_SAMPLES,
            "nfev": nfev, "njev": njev, "iters": float(r0[0, 8]), "passes_per_step": float(allstat[0, 3]) / steps,
            "reference_passes_per_step": ref_passes,
            "event_ms_rank0": float(allstat[0, 2]), "device_us_per_step": float(allstat[0, 5]) / steps,
            "p": [float(v) for v in r0[0, :3]], "sumsq": float(r0[0, 4]),
            # HIP-event time of the fits (one event pair per fit, summed over the timed region) / ALL launches of the
            # pass kernel in them (launch chain: the passes plus the few run-ahead launches per fit that find it finished):
            # the population rocprofv3 averages over.  region_us_per_launch = the whole timed region / launches, i.e.
            # with the host's gaps between two fits
            # Resident regime (one launch per fit): avg_launch_us = the HIP event pair the LIBRARY records on the launch stream
            # right in front of and right behind the kernel launch (brdf_hip_set_launch_timing) -- the kernel's duration, what
            # rocprofv3 --kernel-trace reports for it; call_us_per_launch = the pair recorded here around the whole synchronous
            # call (the kernel plus the host's launch and completion latency, 10-20 us)
            "avg_launch_us": (float(allstat[0, 8]) / float(allstat[0, 9])) if float(allstat[0, 9]) == float(allstat[0, 6]) and float(allstat[0, 9]) > 0
                             else 1e3 * float(allstat[0, 7]) / max(1.0, float(allstat[0, 6])),
            "launch_timer": "library event pair around the kernel launch" if float(allstat[0, 9]) == float(allstat[0, 6]) and float(allstat[0, 9]) > 0
                            else "event pair around the call",
            "call_us_per_launch": 1e3 * float(allstat[0, 7]) / max(1.0, float(allstat[0, 6])),  # (library timer: = region_us_per_launch)
            "region_us_per_launch": 1e3 * float(allstat[0, 2]) / max(1.0, float(allstat[0, 6])),
            "launches_per_step": float(allstat[0, 6]) / steps,
            # device clock (s_memrealtime) from the first to the finishing pass / passes: sweeping launches only
            "avg_sweeping_launch_us": float(allstat[0, 5]) / max(1.0, float(allstat[0, 3])),
        }
    del a_dev, x_dev
    if rank != 0:
        return None, 0

    bpsp = BYTES_PER_SAMPLE_PASS[model]

    def roofline_of(head, method):
        # The dominant kernel.  Resident regime (default when the fit fits the chip): ONE launch per fit.  Launch chain: one launch
        # per sweep.  Algorithmic bytes by SURVEY.md section 8d: bytes per sample-pass x n x the passes of the REFERENCE algorithm
        # (reference_passes: one per Jacobian, one per other evaluation; identical to levmar's own counts in info[7], info[8]).
        # The launch serves them with FEWER sweeps over its resident samples (a chain of rejected trial points shares one sweep; a
        # candidate that is taken brings the next iteration's Jacobian with it): sweeps_per_launch and the fraction by executed
        # sweeps are reported next to it.
        resident = head["launches_per_step"] < 1.5
        kernel = f"resident_fit_kernel<{model}, {method}, true, false>" if resident else f"stream_pass<{model}, {method}, true>"
        traffic, traffic_src = profiled_traffic(kernel)
        sweeps_per_launch = head["passes_per_step"] if resident else 1.0
        ref_per_launch = head["reference_passes_per_step"] if resident else head["reference_passes_per_step"] / max(1.0, head["passes_per_step"])
        bytes_per_launch = bpsp * N_SAMPLES * ref_per_launch
        achieved = bytes_per_launch / (head["avg_launch_us"] * 1e-6) / 1e9
        by_sweeps = bpsp * N_SAMPLES * sweeps_per_launch / (head["avg_launch_us"] * 1e-6) / 1e9
        evals_per_launch = head["evals_per_step_rank0"] / max(1.0, head["launches_per_step"])
        r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "traffic": traffic, "traffic_source": traffic_src, "kernel": "brdf::" + kernel, "regime": "resident" if resident else "launch chain",
             "algorithmic_bytes_per_launch": bytes_per_launch, "reference_passes_per_launch": ref_per_launch,
             "sweeps_per_launch": sweeps_per_launch, "achieved_by_executed_sweeps": by_sweeps, "frac_by_executed_sweeps": by_sweeps / HBM_PEAK_GBS,
             "avg_launch_us": head["avg_launch_us"], "launch_timer": head["launch_timer"], "call_us_per_launch": head["call_us_per_launch"],
             "region_us_per_launch": head["region_us_per_launch"],
             "avg_sweeping_launch_us": head["avg_sweeping_launch_us"], "launches_per_step": head["launches_per_step"],
             "alu": alu_roofline(kernel, head["evals_per_step_rank0"] / (head["avg_launch_us"] * 1e-6 * max(1.0, head["launches_per_step"])),
                                 evals_per_launch)}
        if traffic is not None:  # what really crosses the HBM interface: the samples are read ONCE per resident fit
            r["hbm_measured_gbs"] = traffic / (head["avg_launch_us"] * 1e-6) / 1e9
        return r

    head = out["dif"]
    roof = roofline_of(head, 0)
    roof["note"] = (
        f"achieved = SURVEY.md section 8d's algorithmic bytes ({bpsp} B per sample-pass: "
        f"{'3' if model == 2 else '2'} planes + measurement, fp64) x 1e6 samples x the passes of the reference algorithm for this fit "
        "(reference_passes_per_launch = one per Jacobian + one per other evaluation, from info[7] / info[8]) / the "
        "average HIP-event time of a fit's launch (launch_timer: in the resident regime the event pair the library records on the launch stream right in front of and right behind the kernel launch, brdf_hip_set_launch_timing; call_us_per_launch = the pair around the whole synchronous call, i.e. with the host's launch and completion latency); in the "
        "resident regime one launch is a whole fit: the in-launch exchanges and serial LM steps are part of it.  The launch makes "
        "sweeps_per_launch sweeps over its resident samples for them (shared sweeps: DESIGN.md section 2); frac_by_executed_sweeps counts "
        "those instead.  region_us_per_launch adds the host's gap between two fits; avg_sweeping_launch_us = device clock per sweep.  "
        "traffic = HBM-side bytes per launch from rocprofv3 FETCH_SIZE (x2 gfx950 correction, calibrated) + WRITE_SIZE, valid only for "
        "the sources named in traffic_source: the samples are read ONCE per fit, so the measured traffic (hbm_measured_gbs) is a small "
        "fraction of the algorithmic figure -- the launch is bound by the exchange + LM step latency and by fp64 issue (alu), not by "
        "HBM bandwidth (DESIGN.md section 4)")
    line = {
        "metric": f"BRDF residual-evals/sec (1 M samples, {MODEL_NAME[model].title()} 3-param), whole job; rel-err vs CPU levmar in `parity`",
        "value": head["value"], "unit": "residual-evals/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{1 if model == 2 else 2}]: single-material {MODEL_NAME[model]} 3-param fit, 1,000,000 synthetic samples, "
                               f"dlevmar_dif (FD Jacobian + Broyden), p0={{{', '.join(str(v) for v in p0)}}}, opts={{1e-3,1e-15,1e-15,1e-20,1e-6}}, itmax=100; "
                               "one step = one complete fit, samples resident in HBM when the timed region starts; one material per GPU",
                   "n_samples": N_SAMPLES, "brdf": MODEL_NAME[model], "entry_point": "dlevmar_dif", "fits_per_step_per_gpu": 1,
                   "nfev_per_fit": head["nfev"], "lm_iterations": head["iters"], "reference_passes_per_fit": head["reference_passes_per_step"],
                   "passes_per_fit": head["passes_per_step"]},
        "roofline": roof,
        "fitted_params": head["p"], "sumsq": head["sumsq"],
        "bc_dif": {k: out["bc_dif"][k] for k in ("value", "ms_per_step", "nfev", "njev", "iters", "reference_passes_per_step", "passes_per_step",
                                                    "launches_per_step", "avg_launch_us", "launch_timer", "call_us_per_launch", "avg_sweeping_launch_us", "p")},
    }
    line["bc_dif"]["roofline"] = {k: v for k, v in roofline_of(out["bc_dif"], 1).items()
                                  if k in ("achieved", "frac", "traffic", "traffic_source", "kernel", "algorithmic_bytes_per_launch", "avg_launch_us",
                                           "reference_passes_per_launch", "sweeps_per_launch", "frac_by_executed_sweeps", "alu")}
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
    return line, rc


def channels_line(args, model, ctx, steps, warmup, angles, x, K=3):
    """K colour channels over ONE set of planes (the reference's callers: brdfdata.cpp:1159-1181): one truth, K noise draws;
    dlevmar_bc_dif through brdf_hip_fit_channels_dev (ONE shared resident launch) against K single-fit launches one after the
    other.  Rank 0's line."""
    from brdf_amd import synth
    torch, dist, rank, world, backend, dev, stub = ctx
    import brdf_amd
    lb, ub, itmax, opts, p0 = synth.LB, synth.UB, synth.ITMAX, synth.OPTS, synth.P0[model]
    a_dev = torch.from_numpy(angles).to(dev)
    clean = brdf_amd.model_eval(model, a_dev, synth.TRUTH[model])  # (the product's K1 kernel: f(truth) on the planes)
    gen = torch.Generator(device="cpu").manual_seed(20240 + model)
    xs = [torch.from_numpy(x).to(dev)] + [clean + (0.01 * (torch.rand(x.size, generator=gen, dtype=torch.float64) - 0.5)).to(dev) for _ in range(K - 1)]
    xd = torch.stack(xs).contiguous()
    kw = dict(lb=lb, ub=ub, itmax=itmax, opts=opts)

    brdf_amd.set_launch_timing(True)

    def timed(fn):
        """(last result, wall, us per step by the library's event pairs around the kernel launches -- by the event pair around
        the call where a step did not run as resident launches --, us per step by the event pair around the call)"""
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        kern = []
        t0 = time.perf_counter()
        for s_ in range(steps):
            evs[s_][0].record()
            r, k_us = fn()
            evs[s_][1].record()
            kern.append(k_us)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        call_us = sum(a.elapsed_time(b) for a, b in evs) * 1e3 / steps
        return r, wall, (sum(kern) / steps if min(kern) >= 0.0 else call_us), call_us

    def shared_step():
        r = brdf_amd.fit_channels(1, model, a_dev, xd, p0, **kw)
        return r, brdf_amd.last_channels_stats(K)["kernel_us"]

    def serial_step():
        out, us = [], 0.0
        for c in range(K):
            out.append(brdf_amd.fit_single(1, model, a_dev, xd[c], p0, **kw))
            k_us = brdf_amd.last_fit_stats()["kernel_us"]
            us = us + k_us if (k_us >= 0.0 and us >= 0.0) else -1.0
        return out, us

    shared, wall_sh, us_sh, call_sh = timed(shared_step)
    st = brdf_amd.last_channels_stats(K)
    serial, wall_se, us_se, call_se = timed(serial_step)
    nfev = [float(r.info[7]) for r in shared]
    njev = [float(r.info[8]) for r in shared]
    ref_passes = sum(nj + (nf - 4.0 * nj) for nf, nj in zip(nfev, njev))  # the reference's streaming visits, all channels
    evals = sum(nfev) * N_SAMPLES
    bpsp = BYTES_PER_SAMPLE_PASS[model]
    planes_b = bpsp - 8  # the planes' share of a sample-pass; 8 B = the measurement
    per_fit = bpsp * N_SAMPLES * ref_passes / (us_sh * 1e-6) / 1e9               # every channel pass reads planes + its measurement
    shared_planes = (planes_b + 8 * K) / K * N_SAMPLES * ref_passes / (us_sh * 1e-6) / 1e9  # a round of K passes reads the planes once
    identical = all(np.array_equal(a.p, b.p) and np.array_equal(a.info, b.info) for a, b in zip(shared, serial))
    kernel = f"channels_fit_kernel<{model}, true>"
    traffic, traffic_src = profiled_traffic(kernel)
    return {
        "workload": f"{K} channels over one set of planes (one truth, {K} noise draws), {MODEL_NAME[model]}, 1,000,000 samples each, dlevmar_bc_dif, "
                    "brdf_hip_fit_channels_dev: one shared resident launch; one step = the K fits",
        "channels": K, "entry_point": "dlevmar_bc_dif", "steps": steps, "shared_launch": bool(st["shared_launch"]),
        "value": evals * steps / wall_sh, "unit": "residual-evals/s", "ms_per_step": 1e3 * wall_sh / steps, "launch_us": us_sh, "call_us": call_sh,
        "one_after_the_other": {"ms_per_step": 1e3 * wall_se / steps, "launches_us": us_se, "calls_us": call_se, "speedup_of_the_shared_launch": us_se / us_sh},
        "nfev": nfev, "passes": [c["passes"] for c in st["channels"]], "bit_identical_to_single_fits": bool(identical),
        "fitted_params": [[float(v) for v in r.p] for r in shared],
        "roofline": {"bound": "hbm", "kernel": "brdf::" + kernel, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "achieved": per_fit, "frac": per_fit / HBM_PEAK_GBS,
                     "achieved_shared_planes": shared_planes, "frac_shared_planes": shared_planes / HBM_PEAK_GBS,
                     "reference_passes_per_launch": ref_passes, "traffic": traffic, "traffic_source": traffic_src,
                     "alu": alu_roofline(kernel, evals / (us_sh * 1e-6), evals),
                     "note": f"achieved: SURVEY.md section 8d's {bpsp} B per sample-pass per fit x the passes of the reference algorithm over all {K} "
                             f"channels / the launch; achieved_shared_planes: ({planes_b} + 8 K) / K B per sample-pass per fit -- the {K} channels' "
                             "passes of one round share one read of the planes"}}


def main_single(args, model):
    """c2 / c3.  The default run (`python bench.py`: c2 on one GPU) also carries the other BASELINE.json configurations as
    short sub-lines under `configs`: c3 through both entry points, c4 and c5 through dlevmar_dif -- a few steps each, parity
    against single CPU fits (c3) or the CPU fits of the first 64 surfels (c4 / c5) instead of the timed CPU baselines, which
    the explicit `--workload` runs keep."""
    from brdf_amd import synth
    rank_env = int(os.environ.get("RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    angles, x, truth = synth.make_single(model, N_SAMPLES, seed=synth.SEED + 7919 * rank_env)
    with_cpu = not args.no_cpu and rank_env == 0 and world_env == 1
    subs = args.workload == "c2" and args.gpus == 1 and world_env == 1 and not args.headline_only and os.environ.get("BRDF_BENCH_STUB") != "1"
    # CPU legs first, before this process has touched the GPU (they are never inside a timed region either way)
    rank0_cpu = single_cpu_refs(model, angles, x) if with_cpu else {}
    sub_cpu = {}
    if subs:
        a3, x3, _ = synth.make_single(1, N_SAMPLES, seed=synth.SEED)
        sub_cpu["c3"] = (a3, x3, single_cpu_refs(1, a3, x3, timed=False) if with_cpu else {})
        for wl in ("c4", "c5"):
            sub_cpu[wl] = batched_cpu_subset(wl, "dif") if with_cpu else None
    ctx = dist_setup(args)
    torch, dist, rank, world = ctx[0], ctx[1], ctx[2], ctx[3]
    line, rc = single_line(args, model, ctx, args.steps, args.warmup, angles, x, rank0_cpu)
    if rank == 0 and args.channels > 1 and os.environ.get("BRDF_BENCH_STUB") != "1":
        line["channels"] = channels_line(args, model, ctx, args.steps, args.warmup, angles, x, args.channels)
    if rank == 0 and subs:
        configs = {}
        try:
            a3, x3, refs3 = sub_cpu["c3"]
            l3, rc3 = single_line(args, 1, ctx, 10, 2, a3, x3, refs3)
            keep = ("value", "ms_per_step", "steps", "config", "roofline", "fitted_params", "sumsq", "bc_dif", "parity")
            configs["c3"] = {k: l3[k] for k in keep if k in l3}
            for part in (configs["c3"], configs["c3"].get("bc_dif", {})):
                part.pop("cpu_baseline", None)  # (one untimed CPU fit per entry point: a parity reference, not a baseline)
            rc = rc or rc3
            if args.channels <= 1:
                configs["c2_three_channels"] = channels_line(args, model, ctx, 10, 2, angles, x, 3)
                if not configs["c2_three_channels"]["bit_identical_to_single_fits"]:
                    print("bench.py: the shared launch's channels differ from single fits", file=sys.stderr)
                    rc = rc or 3
            for wl in ("c4", "c5"):
                lb_ = batched_line(args, wl, "dif", ctx, 2, 1, None, sub_cpu[wl])
                configs[wl + "_dif"] = {k: lb_[k] for k in ("value", "ms_per_step", "steps", "config", "roofline", "result_sha256", "parity",
                                                           "event_ms_per_step") if k in lb_}
                if "parity" in lb_ and not lb_["parity"]["ok"]:
                    print(f"bench.py: PARITY FAILURE in the {wl} sub-line", file=sys.stderr)
                    rc = rc or 3
        except Exception as exc:  # noqa: BLE001  (a sub-line must never cost the headline)
            configs["error"] = f"{type(exc).__name__}: {exc}"
            rc = rc or 4
        line["configs"] = configs
    if rank == 0:
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
    ap.add_argument("--headline-only", action="store_true", help="c2 on one GPU without the configs[2..4] sub-lines")
    ap.add_argument("--channels", type=int, default=1, help="c2 / c3: add a line for K colour channels over one set of planes (dlevmar_bc_dif, "
                    "brdf_hip_fit_channels_dev: one shared resident launch) next to the headline")
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
