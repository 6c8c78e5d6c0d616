# A/B of caller streams x deferred-tail modes on one box (interleaved: the boxes drift with temperature).
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]), "q/s", round(d["ms_per_step"]*1e3,1), "us/step | scan", round(r["avg_launch_us"],1), "frac", round(r["frac"],3), "iso", round(r["isolated"]["avg_launch_us"],1), d.get("ids_exact", d.get("ids_match_exact_fp64_scan")))'
for cfg in "1 2 4" "2 0 4" "1 2 1" "1 2 4" "2 2 4" "1 2 8"; do set -- $cfg
python bench.py --no-cpu-baseline --streams $1 --pipeline $2 --event-stride $3 2>/dev/null | tail -1 | python -c "$P" "1M s=$1 p=$2 ev=$3" || exit 1
done
for cfg in "2 0" "2 2" "2 0" "2 2"; do set -- $cfg
RQ_BENCH_FORCE_COMM=1 python bench.py --rows 125000 --no-cpu-baseline --steps 1600 --warmup 160 --streams $1 --pipeline $2 2>/dev/null | tail -1 | python -c "$P" "125k+comm s=$1 p=$2" || exit 1
done
