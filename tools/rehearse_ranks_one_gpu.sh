# The N > 1 path of bench.py with REAL ranks (torch.distributed.run, one process per rank) on ONE GPU: RCCL refuses two ranks on one device,
# so the keys travel through host memory (RQ_BENCH_BACKEND=gloo).  What this run is evidence for: the row sharding by rank, the per-rank
# searches (rq_search_train_device), the all-gather + device merge, the repair path and the cross-check of the merged top-k against every
# rank's exact fp64 scan, with world sizes 2 and 4 -- NOT timings (the ranks share one GPU and exchange through the host).
for N in 2 4; do
RQ_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29540 + N)) bench.py --gpus $N --steps 48 --warmup 16 2>/dev/null | tail -1 | python -c '
import json, sys
d = json.loads(sys.stdin.read())
print("ranks", d["n_gpus"], "ranks_seen", d["config"]["ranks_seen"], "rows_per_gpu", d["config"]["rows_per_gpu"], "dtype", d["dtype"], "enqueue:", d["config"]["enqueue"], "| exchange:", d["config"]["exchange"],
      "| ids_match_exact_fp64_scan", d["ids_match_exact_fp64_scan"], "recall", d["recall_at_10_vs_exact_fp64_scan"], "max score diff", d["max_abs_score_diff_vs_exact_fp64_scan"], "| repaired", d["repaired_queries"])' || exit 1
done
