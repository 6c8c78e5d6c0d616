# Runs on the GPU box: matrix-core / VALU / LDS counters of the encoder's attention kernel at 512 tokens (tools/gpu_r03_enc512.py), separate --pmc passes.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/e512_mfma $R/gpurun_out/e512_lds
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/e512_mfma -- python3 $R/tools/gpu_r03_enc512.py > $R/gpurun_out/e512_mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/e512_lds -- python3 $R/tools/gpu_r03_enc512.py > $R/gpurun_out/e512_lds.log 2>&1 || echo "second counter pass failed (counter names), skipped"
cd $R
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("e512_mfma", "e512_lds"):
    files = sorted(glob.glob(f"gpurun_out/{tag}/*/*counter_collection.csv"))
    if not files: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(files[-1])):
        if "rq_nb_attention" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]["v"] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c, d in acc.items(): out[c] = d["v"] / max(n[c], 1)
    out[tag + "_launches"] = max(n.values()) if n else 0
json.dump(out, open("gpurun_out/r03_attention512_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
