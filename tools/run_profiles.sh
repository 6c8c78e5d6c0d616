# Runs on the GPU box (gpurun -- bash tools/run_profiles.sh): the headline loop (bench.py, fp16 rows) and the int8 loop (bench.py --scan int8),
# each WITHOUT the extra legs, under rocprofv3 kernel stats and in three --pmc passes of their own (no trace domains mixed into a counter
# pass); then the plain default bench twice (400 and 20 steps, all legs).  tools/make_profiles.py condenses the outputs into profiles/.
# (The boxes slow down by ~5 % after half a minute of sustained load: the kernel-stats runs go first.)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_bench $O/prof_bench8 $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_fetch8 $O/pmc_write8 $O/pmc_mfma8
LEAN="--no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py $LEAN > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench8 -- python3 $R/bench.py --scan int8 $LEAN > $O/prof_bench8.log 2>&1
cd $R
python bench.py > gpurun_out/bench2.log 2>gpurun_out/bench2.err
tail -1 gpurun_out/bench2.log | cut -c1-300
cd /tmp
for v in "" 8; do
  S=""; [ "$v" = "8" ] && S="--scan int8"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch$v -- python3 $R/bench.py $S --steps 20 --warmup 5 $LEAN > $O/pmc_fetch$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write$v -- python3 $R/bench.py $S --steps 20 --warmup 5 $LEAN > $O/pmc_write$v.log 2>&1
  # matrix-core counters (north star: "rocprof HBM GB/s and MFMA utilisation against gfx950 peak"), their own pass
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma$v -- python3 $R/bench.py $S --steps 20 --warmup 5 $LEAN > $O/pmc_mfma$v.log 2>&1
done
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/bench20.log 2>gpurun_out/bench20.err
find $O/prof_bench $O/prof_bench8 $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_fetch8 $O/pmc_write8 $O/pmc_mfma8 -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
ls -R gpurun_out/prof_bench | head -4
