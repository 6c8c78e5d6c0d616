# Per-rank shapes of the 1/2/4/8-GPU runs, rehearsed on ONE GPU with the all-gather + merge forced into the loop (RQ_BENCH_FORCE_COMM=1:
# a world of one rank, RCCL all-gather of the keys every 16 batches, device merge).  The SAME scan operand at every shape (bench.py --scan),
# first the fp16 rows (the headline basis), then the int8 image.  usage: bash tools/rehearse_scaling.sh [steps]
STEPS=${1:-800}
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]), "q/s", round(d["ms_per_step"]*1e3,1), "us/step | scan", round(r["isolated"]["avg_launch_us"],1), "us alone, frac", round(r["isolated"]["frac"],3), "| dtype", d["dtype"], "| enqueue:", d["config"].get("enqueue"), "| exact:", d.get("ids_exact", d.get("ids_match_exact_fp64_scan")), "| host enqueue us/step", round(d["host_phases"]["enqueue_all_steps_us"]/d["steps"],1))'
for scan in fp16 int8; do
python bench.py --scan $scan --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "$P" "scan=$scan N=1 shape (1M rows, no exchange)" || exit 1
for rows in 1000000 500000 250000 125000; do
RQ_BENCH_FORCE_COMM=1 python bench.py --scan $scan --rows $rows --no-cpu-baseline --steps $STEPS --warmup 80 2>/dev/null | tail -1 | python -c "$P" "scan=$scan per-rank shape $rows rows + exchange" || exit 1
done
done
RQ_BENCH_TRAIN=0 RQ_BENCH_FORCE_COMM=1 python bench.py --scan fp16 --rows 125000 --no-cpu-baseline --steps $STEPS --warmup 80 2>/dev/null | tail -1 | python -c "$P" "scan=fp16 per-rank shape 125000 rows + exchange, python loop (no train call)"
RQ_BENCH_TRAIN=0 RQ_BENCH_FORCE_COMM=1 python bench.py --scan int8 --rows 125000 --no-cpu-baseline --steps $STEPS --warmup 80 2>/dev/null | tail -1 | python -c "$P" "scan=int8 per-rank shape 125000 rows + exchange, python loop (no train call)"
