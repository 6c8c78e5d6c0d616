# Runs on the GPU box: rocprofv3 kernel stats + matrix-core / LDS / HBM counters of the int8 256-query pass (round 3, csrc/rq_scan_wide.hip I8)
# next to round 2's 128-query int8 pass (rq_scan.hip I8 = 3), via tools/gpu_wide_one.py.  Separate --pmc passes, no trace domains mixed in.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SPEC="scan8=2,wide256_8=31;scan8=2,wide256_8=22;scan8=2,wide256_8=0"
rm -rf $R/gpurun_out/w8_stats $R/gpurun_out/w8_mfma $R/gpurun_out/w8_fetch $R/gpurun_out/w8_lds
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/w8_stats -- python3 $R/tools/gpu_wide_one.py 256 "$SPEC" > $R/gpurun_out/w8_stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/w8_mfma -- python3 $R/tools/gpu_wide_one.py 256 "$SPEC" > $R/gpurun_out/w8_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/w8_fetch -- python3 $R/tools/gpu_wide_one.py 256 "$SPEC" > $R/gpurun_out/w8_fetch.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/w8_lds -- python3 $R/tools/gpu_wide_one.py 256 "$SPEC" > $R/gpurun_out/w8_lds.log 2>&1 || echo "lds counter pass failed (counter names), skipped"
cd $R
python - <<'PY'
import csv, glob, json, collections
out = {}
ks = sorted(glob.glob("gpurun_out/w8_stats/**/*kernel_stats.csv", recursive=True))[-1]
sel = lambda name: ("rq_scanw" in name) or ("rq_scan_kernel" in name and ", 4>" in name)   # rq_scanw32_kernel (32x32x32), rq_scanw_kernel (16x16x64), rq_scan_kernel<.., 4> (128 queries)
for r in csv.DictReader(open(ks)):
    if sel(r["Name"]):
        out.setdefault(r["Name"][:64], {})["avg_launch_us_kernel_stats"] = float(r["AverageNs"]) / 1e3
        out[r["Name"][:64]]["calls"] = int(r["Calls"])
for d in ("w8_mfma", "w8_fetch", "w8_lds"):
    fs = sorted(glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True))
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        if sel(r["Kernel_Name"]):
            acc[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
for k, v in out.items():
    q = 256 if "rq_scanw" in k else 128
    us = v.get("avg_launch_us_kernel_stats")
    v["queries_per_pass"] = q
    if us and "SQ_INSTS_VALU_MFMA_MOPS_I8" in v:
        ops = v["SQ_INSTS_VALU_MFMA_MOPS_I8"] * 512
        v["mfma_TOPs"] = ops / (us * 1e-6) / 1e12
        v["mfma_frac_of_dense_int8_peak"] = v["mfma_TOPs"] / 5000.0
        v["ops_over_algorithmic"] = ops / (2.0 * q * 1_000_000 * 768)
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)
    if us and "FETCH_SIZE" in v:
        v["hbm_read_bytes_per_launch"] = 2 * v["FETCH_SIZE"] * 1024      # gfx950: FETCH_SIZE reports half of a wide streaming read
        v["read_over_algorithmic"] = v["hbm_read_bytes_per_launch"] / 0.768e9
        v["hbm_frac_of_peak"] = 0.768e9 / (us * 1e-6) / 8e12
        v["queries_per_s_per_pass"] = q / (us * 1e-6)
json.dump(out, open("gpurun_out/r03_wide8_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out/w8_stats gpurun_out/w8_mfma gpurun_out/w8_fetch gpurun_out/w8_lds -type f -delete 2>/dev/null || true
