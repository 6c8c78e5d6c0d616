P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]), "q/s", round(d["ms_per_step"]*1e3,1), "us/step | scan", round(r["avg_launch_us"],1), "frac", round(r["frac"],3), d.get("ids_exact", d.get("ids_match_exact_fp64_scan")))'
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P" "N=1 shape (1M rows)" || exit 1
for rows in 500000 250000 125000; do
RQ_BENCH_FORCE_COMM=1 python bench.py --rows $rows --no-cpu-baseline --steps 800 --warmup 80 2>/dev/null | tail -1 | python -c "$P" "per-rank shape $rows rows + comm" || exit 1
done
