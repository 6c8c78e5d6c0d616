cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST" "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL" "TCC_TAG_STALL TCC_WRITE_REQ TCC_NORMAL_WRITEBACK"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmcx_$n
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcx_$n -- python3 $R/tools/gpu_pmc_stores.py > $R/gpurun_out/pmcx_$n.log 2>&1 || exit 1
  find $R/gpurun_out/pmcx_$n -type f ! -name "*counter_collection.csv" -delete
done
