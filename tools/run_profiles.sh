# Runs on the GPU box (gpurun -- bash tools/run_profiles.sh): the default bench under rocprofv3 kernel stats, the plain
# default bench, and the two PMC passes; tools/make_profiles.py condenses the outputs into profiles/.
# (The boxes slow down by ~5 % after half a minute of sustained load: the kernel-stats run goes first.)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_mfma $R/gpurun_out/prof_bench_1s
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R
python bench.py > gpurun_out/bench2.log 2>gpurun_out/bench2.err
tail -1 gpurun_out/bench2.log | cut -c1-400
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
# matrix-core counters (north star: "rocprof HBM GB/s and MFMA utilisation against gfx950 peak"), their own pass
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/bench20.log 2>gpurun_out/bench20.err
find gpurun_out/prof_bench gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
ls -R gpurun_out/prof_bench | head
