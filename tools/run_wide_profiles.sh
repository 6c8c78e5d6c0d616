# Runs on the GPU box: rocprofv3 kernel stats + matrix-core / HBM counters of the 128- and 256-query passes (tools/gpu_wide_one.py).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SPEC="wide_batch=3,wide128=0;wide_batch=1,wide256=2"
rm -rf $R/gpurun_out/wide_stats $R/gpurun_out/wide_mfma $R/gpurun_out/wide_fetch
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wide_stats -- python3 $R/tools/gpu_wide_one.py 256 "$SPEC" > $R/gpurun_out/wide_stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/wide_mfma -- python3 $R/tools/gpu_wide_one.py 256 "$SPEC" > $R/gpurun_out/wide_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/wide_fetch -- python3 $R/tools/gpu_wide_one.py 256 "$SPEC" > $R/gpurun_out/wide_fetch.log 2>&1
cd $R
python - <<'PY'
import csv, glob, json, collections
out = {}
ks = sorted(glob.glob("gpurun_out/wide_stats/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(ks)):
    if "rq_scanw" in r["Name"]:
        out.setdefault(r["Name"][:60], {})["avg_launch_us_kernel_stats"] = float(r["AverageNs"]) / 1e3
        out[r["Name"][:60]]["calls"] = int(r["Calls"])
for d in ("wide_mfma", "wide_fetch"):
    f = sorted(glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "rq_scanw" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
for k, v in out.items():
    q = 256 if ", 8, 2, " in k else 128
    us = v.get("avg_launch_us_kernel_stats")
    if us and "SQ_INSTS_VALU_MFMA_MOPS_F16" in v:
        flop = v["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512
        v["queries_per_pass"] = q
        v["mfma_TFLOPs"] = flop / (us * 1e-6) / 1e12
        v["mfma_frac_of_dense_fp16_peak"] = v["mfma_TFLOPs"] / 2500.0
        v["flop_over_algorithmic"] = flop / (2.0 * q * 1_000_000 * 768)
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)
    if us and "FETCH_SIZE" in v:
        v["hbm_read_bytes_per_launch"] = 2 * v["FETCH_SIZE"] * 1024      # gfx950: FETCH_SIZE reports half of a wide streaming read
        v["read_over_algorithmic"] = v["hbm_read_bytes_per_launch"] / 1.536e9
        v["hbm_frac_of_peak"] = 1.536e9 / (us * 1e-6) / 8e12
json.dump(out, open("gpurun_out/r02_wide_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out/wide_stats gpurun_out/wide_mfma gpurun_out/wide_fetch -type f -delete
