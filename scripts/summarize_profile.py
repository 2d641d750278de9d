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
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
    p = find(sub, "*counter_collection.csv")
    if not p:
        continue
    print(f"== {counter} per launch, KiB as reported ({sub}) ==")
    for k, v in sorted(pmc_stats(p, counter).items(), key=lambda kv: -sum(kv[1])):
        big = [x for x in v if x > 0.05 * max(v)] if max(v) > 0 else v
        print(f"{k:45s} launches {len(v):6d} mean {st.mean(v):12.1f} median {st.median(v):12.1f} max {max(v):12.1f} | non-trivial launches: {len(big)} mean {st.mean(big):12.1f}")
