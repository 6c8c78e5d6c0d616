"""Condense rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/.

    python tools/make_profiles.py r01

expects  gpurun_out/prof_bench/*/*kernel_stats.csv      rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline
         gpurun_out/pmc_fetch/*/*counter_collection.csv  rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
         gpurun_out/pmc_write/*/*counter_collection.csv  rocprofv3 --pmc WRITE_SIZE -- (same)
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

def first(pattern):
    g = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    return g[-1] if g else None

ks = first("prof_bench/*/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    keep = [r for r in rows if "rq_" in r["Name"]]
    with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    print("kernel stats:", [(r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1)) for r in keep[:6]])

ks1 = first("prof_bench_1s/*/*kernel_stats.csv")
if ks1:
    rows = list(csv.DictReader(open(ks1)))
    with open(os.path.join(out, f"{tag}_bench_single_stream_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows([r for r in rows if "rq_" in r["Name"]])
    print("single-stream kernel stats:", [(r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows if "rq_" in r["Name"]][:4])
for logname in ("bench2", "prof_bench", "prof_bench_1s"):
    lp = os.path.join(ROOT, "gpurun_out", logname + ".log")
    if os.path.exists(lp):
        lines = [l for l in open(lp, errors="replace").read().splitlines() if l.startswith('{"metric"')]
        if lines:
            open(os.path.join(out, f"{tag}_{logname}.json"), "w").write(lines[-1] + "\n")
            d = json.loads(lines[-1])
            print(logname, "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "scan avg us", round(d["roofline"]["avg_launch_us"], 1), "frac", round(d["roofline"]["frac"], 3))

pmc = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = first(f"{name}/*/*counter_collection.csv")
    if not f:
        continue
    recs = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    fused = [r for r in recs if "rq_scan_tail" in r["Kernel_Name"]]     # the bench default: scan + previous batch's tail
    vals = [float(r["Counter_Value"]) for r in (fused or [r for r in recs if "rq_scan" in r["Kernel_Name"]])]
    pmc_kernel = "rq_scan_tail_kernel" if fused else "rq_scan_kernel"
    pmc[counter] = {"launches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals), "unit": "KiB (rocprofv3 derived counter)"}
if "FETCH_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"]["mean"] * 1024
    write = pmc.get("WRITE_SIZE", {"mean": 0.0})["mean"] * 1024
    summary = {
        "kernel": pmc_kernel, "workload": "1000000x768 fp16 corpus, 64 queries per launch",
        "counters": pmc,
        "correction": "MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced "
                      "streaming read (16 B/lane, global_load and LDS-DMA alike) -> read bytes = 2 * FETCH_SIZE * 1024; "
                      "WRITE_SIZE is taken as is (the pooled stores are 4..16 B per lane, an uncalibrated width)",
        "hbm_read_bytes_per_launch": 2 * fetch,
        "hbm_write_bytes_per_launch": write,
        "hbm_bytes_per_launch": 2 * fetch + write,
        "algorithmic_bytes_per_launch": 1_000_000 * 768 * 2,
    }
    summary["traffic_over_algorithmic"] = summary["hbm_bytes_per_launch"] / summary["algorithmic_bytes_per_launch"]
    json.dump(summary, open(os.path.join(out, f"{tag}_pmc_scan.json"), "w"), indent=1)
    print("pmc:", {k: summary[k] for k in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "traffic_over_algorithmic")})
