"""Condenses the rocprofv3 CSVs of scripts/profile_round.sh into one text summary (per-kernel duration stats,
PMC bytes per launch, calibration factors)."""
import csv, glob, os, statistics as st, sys, collections

root = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.replace("void brdf::", "").split("(")[0]


def trace_stats(path):
    rows = list(csv.DictReader(open(path)))
    d = collections.defaultdict(list)
    for r in rows:
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


def pmc_stats(path, counter):
    rows = list(csv.DictReader(open(path)))
    d = collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name") == counter:
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d


t = find("trace", "*kernel_trace.csv")
if t:
    print("== kernel durations (rocprofv3 --kernel-trace), ns ==")
    for k, v in sorted(trace_stats(t).items(), key=lambda kv: -sum(kv[1])):
        real = [x for x in v if x > 5000]  # launches that did a sweep (run-ahead launches after the fit ended return at once)
        print(f"{k:45s} calls {len(v):6d} total {sum(v)/1e6:9.3f} ms  avg {st.mean(v):9.0f}  median {st.median(v):9.0f}  "
              f"min {min(v):7d} max {max(v):8d} | sweeping launches: {len(real)} avg {st.mean(real) if real else 0:9.0f}")
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE"),
                     ("pmc_valu", "SQ_INSTS_VALU"), ("pmc_valu", "SQ_ACTIVE_INST_VALU"), ("pmc_valu", "SQ_WAVE_CYCLES"), ("pmc_valu", "SQ_BUSY_CYCLES")):
    p = find(sub, "*counter_collection.csv")
    if not p:
        continue
    print(f"== {counter} per launch, as reported ({sub}; FETCH/WRITE_SIZE in KiB) ==")
    for k, v in sorted(pmc_stats(p, counter).items(), key=lambda kv: -sum(kv[1])):
        big = [x for x in v if x > 0.05 * max(v)] if max(v) > 0 else v
        print(f"{k:45s} launches {len(v):6d} mean {st.mean(v):12.1f} median {st.median(v):12.1f} max {max(v):12.1f} | non-trivial launches: {len(big)} mean {st.mean(big):12.1f}")


# ---- traffic record for bench.py (roofline.traffic): HBM bytes per launch of the fit kernels, tied to the sources profiled
if len(sys.argv) > 3 and sys.argv[2] == "--traffic-json":
    import hashlib, json
    out_path, tag = sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "r02")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(here, "brdf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    fetch = pmc_stats(find("pmc_fetch", "*counter_collection.csv"), "FETCH_SIZE")
    write = pmc_stats(find("pmc_write", "*counter_collection.csv"), "WRITE_SIZE")
    cal_f = pmc_stats(find("cal_fetch", "*counter_collection.csv"), "FETCH_SIZE")
    cal_w = pmc_stats(find("cal_write", "*counter_collection.csv"), "WRITE_SIZE")
    # calibration: model_eval_kernel<2> reads 3 x 512 MiB and writes 512 MiB in the pass kernels' own 8 B/lane pattern
    kf = next((v for k, v in cal_f.items() if "model_eval_kernel<2>" in k), None)
    kw = next((v for k, v in cal_w.items() if "model_eval_kernel<2>" in k), None)
    fcorr = (3 * 512 * 1024) / st.mean(kf) if kf else 2.0
    wcorr = (512 * 1024) / st.mean(kw) if kw else 1.0
    pv = find("pmc_valu", "*counter_collection.csv")
    valu = pmc_stats(pv, "SQ_INSTS_VALU") if pv else {}
    valu_busy = pmc_stats(pv, "SQ_ACTIVE_INST_VALU") if pv else {}
    wave_cyc = pmc_stats(pv, "SQ_WAVE_CYCLES") if pv else {}
    kernels = {}
    for k, v in fetch.items():
        if "fit_kernel" not in k and "stream_pass" not in k:
            continue
        big = [x for x in v if x > 0.05 * max(v)] if max(v) > 0 else v
        w = write.get(k, [0.0])
        wbig = [x for x in w if x > 0.05 * max(w)] if max(w) > 0 else w
        kernels[k] = {"fetch_kib_reported": st.mean(big), "fetch_correction": fcorr, "write_kib_reported": st.mean(wbig), "write_correction": wcorr,
                      "hbm_bytes_per_launch": int(1024 * (st.mean(big) * fcorr + st.mean(wbig) * wcorr))}
        if k in valu:  # VALU wave-instructions per (sweeping) launch: bench.py's roofline.alu
            iv = valu[k]
            ibig = [x for x in iv if x > 0.05 * max(iv)] if max(iv) > 0 else iv
            kernels[k]["valu_wave_instr_per_launch"] = st.mean(ibig)
            if k in valu_busy and k in wave_cyc:
                kernels[k]["sq_active_inst_valu_per_launch"] = st.mean([x for x in valu_busy[k] if x > 0.05 * max(valu_busy[k])])
                kernels[k]["sq_wave_cycles_per_launch"] = st.mean([x for x in wave_cyc[k] if x > 0.05 * max(wave_cyc[k])])
    json.dump({"tag": tag, "source_hash": h.hexdigest()[:16],
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU ... in separate passes over `python3 bench.py --steps 10 --warmup 2 --no-cpu` (scripts/profile_round.sh)",
               "calibration": f"scripts/calib_traffic.py: model_eval_kernel<2> reads 3 x 512 MiB / writes 512 MiB, 8 B per lane coalesced: FETCH_SIZE x {fcorr:.4f}, WRITE_SIZE x {wcorr:.4f}",
               "kernels": kernels}, open(out_path, "w"), indent=1)
    print("wrote", out_path)
