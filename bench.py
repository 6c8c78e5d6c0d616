#!/usr/bin/env python3
"""bench.py -- queries/sec of the dense-retrieval hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): 1M x 768 fp16 synthetic corpus, batches of 64 queries, top-10,
exact results.  A "step" = one 64-query batch searched against the whole corpus.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: the corpus is row-sharded (rank r holds rows [r*N/G, (r+1)*N/G)), every rank searches every
batch on its shard, the per-shard (score, row) keys are exchanged with one RCCL all-gather per
GATHER_EVERY batches and merged on the GPU.  Total work is fixed as N grows => "scaling": "strong".

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (scan kernel: algorithmic bytes / HIP-event
duration measured here, over the timed region) and, at N = 1, "cpu_baseline" (oracle port, fp32
OpenBLAS brute force on this box's host cores, bounded sample).

--scan auto (default) leaves the choice of the scan's corpus operand to the library: shards of 200 000 rows and more are scanned
through their int8 image (768 B per row; every returned row is still re-scored in fp64 from the fp16 rows and
certified, and the results of the timed region are checked against the exact fp64 scan and the oracle afterwards); "roofline"
then counts the bytes of that image, "dtype" says "i8", and "fp16_scan" carries the same loop over the fp16 rows (--scan fp16
times that as the main figure: SURVEY 8(d)'s 1536 B per row).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS = 1_000_000
DIM = 768
BATCH = 64
TOPK = 10
N_QUERY_BATCHES = 16          # distinct query batches cycled through the steps
GATHER_EVERY = int(os.environ.get("RQ_BENCH_GATHER_EVERY", "16"))   # batches per all-gather (N > 1)
CHUNK_ROWS = 125_000          # corpus generated in chunks seeded by global chunk id: same corpus for any N
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--rows", type=int, default=N_ROWS)
    ap.add_argument("--k", type=int, default=TOPK)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=0, help="caller streams the batches alternate over (0 = 1 at one GPU, 2 at several)")
    ap.add_argument("--event-stride", type=int, default=4, help="HIP events around every n-th scan launch of the timed region")
    ap.add_argument("--scan", choices=("auto", "fp16", "int8"), default="auto",
                    help="corpus operand of the scan: fp16 rows, their int8 image (half the bytes; candidates are still re-scored from "
                         "the fp16 rows in fp64), or the library's rule (int8 image on shards of 200k rows and more)")
    ap.add_argument("--no-hint", action="store_true", help="do not announce the next batch (rq_search_hint_next_device): every call prepares its own queries in a separate launch")
    ap.add_argument("--pipeline", type=int, default=2, help="deferred tails: 1 = on the library's internal stream, 2 = fused into the next scan launch")
    ap.add_argument("--workload", default="headline", choices=["headline", "config2", "config3", "config4"],
                    help="headline = BASELINE.json configs[1] (the metric); config2/3/4 = the other GPU configs at their shapes on ONE GPU "
                         "(tests/config_workloads.py, synthetic stand-ins), one JSON line each")
    args = ap.parse_args()
    if args.workload != "headline":
        if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
            raise SystemExit("--workload config2/3/4 are single-GPU legs")
        sys.path.insert(0, os.path.join(ROOT, "tests"))          # the harness lives with the tests (it uses the oracle as its checker)
        import config_workloads as cw
        r = {"config2": cw.run_config2, "config3": cw.run_config3, "config4": cw.run_config4}[args.workload]()
        metric = {"config2": ("queries/sec, 10M x 768 fp16 in 8 row shards on one GPU, batch-64, top-10", r.get("queries_per_s")),
                  "config3": ("text queries/sec end to end (NomicBert forward + search over 1M x 768), 256 per call", r.get("text_queries_per_s")),
                  "config4": ("questions/sec, GPU dense top-100 + CPU BM25 top-100 + fusion, 500 per call", r.get("questions_per_s_hybrid"))}[args.workload]
        print(json.dumps({"metric": metric[0], "value": metric[1], "unit": "queries/s", "n_gpus": 1, "higher_is_better": True,
                          "vs_baseline": None, "dtype": "f16", "data": r["data"], "config": {"workload": r["workload"]}, "details": r}), flush=True)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    import rag_uq_amd  # noqa: F401
    from rag_uq_amd import _native as nat

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run, one rank per GPU "
                         "(python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RQ_BENCH_FORCE_COMM=1 exercises the all-gather + merge path on a single rank (rehearsal on a 1-GPU box)
    use_comm = world > 1 or os.environ.get("RQ_BENCH_FORCE_COMM") == "1"
    if use_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    n_total, k, B = args.rows, args.k, BATCH
    n_chunks = (n_total + CHUNK_ROWS - 1) // CHUNK_ROWS
    c_lo, c_hi = rank * n_chunks // world, (rank + 1) * n_chunks // world
    row_lo = min(c_lo * CHUNK_ROWS, n_total)
    row_hi = min(c_hi * CHUNK_ROWS, n_total)
    n_local = row_hi - row_lo

    # ---- corpus shard: synthetic Gaussian rows, unit norm, fp16, generated in HBM ---------------
    idx = nat.NativeIndex(DIM, local_rank)
    idx.reserve(max(n_local, 1))
    idx.set_row_offset(row_lo)
    idx.set_option("pipeline", args.pipeline)
    idx.set_option("scan8", {"auto": 1, "fp16": 0, "int8": 2}[args.scan])
    for opt in filter(None, os.environ.get("RQ_BENCH_OPTS", "").split(",")):      # development: e.g. RQ_BENCH_OPTS=wg_per_cu=3
        name, val = opt.split("=")
        idx.set_option(name, float(val))
    # One GPU: one caller stream, so scan launches never overlap each other and the per-launch HIP events (and
    # rocprofv3) read the kernel's own duration; the tail of batch i hides inside the scan launch of batch i+1
    # (pipeline 2).  Several GPUs (125k-row shards, ~35 us steps): two caller streams also hide the query prep
    # and the launch gaps (measured 34.8 vs 50.1 us per step); per-launch events are off there.
    if args.streams <= 0:
        args.streams = 2 if use_comm else 1
    for c in range(c_lo, c_hi):
        g = torch.Generator(device=dev)
        g.manual_seed(1235 + c)
        n = min(CHUNK_ROWS, n_total - c * CHUNK_ROWS)
        x = torch.randn((n, DIM), device=dev, generator=g, dtype=torch.float32)
        x = torch.nn.functional.normalize(x, dim=1).half().contiguous()
        idx.add_f16_device(x, n)
        del x
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = [torch.randn((B, DIM), device=dev, generator=gq, dtype=torch.float32) for _ in range(N_QUERY_BATCHES)]
    torch.cuda.synchronize()

    # ---- per-batch output slots (results stay in HBM) -----------------------------------------------
    status_all = torch.zeros((N_QUERY_BATCHES, B), device=dev, dtype=torch.int32)   # one read-back checks every slot

    def slot(j):
        return dict(scores=torch.empty((B, k), device=dev), rows=torch.empty((B, k), device=dev, dtype=torch.int64),
                    keys=torch.empty((B, k), device=dev, dtype=torch.int64), status=status_all[j])
    slots = [slot(j) for j in range(N_QUERY_BATCHES)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    comm_stream = torch.cuda.Stream(device=dev)
    if use_comm:
        ring = [torch.zeros((GATHER_EVERY, B, k), device=dev, dtype=torch.int64) for _ in range(2)]
        gathered = [torch.zeros((world, GATHER_EVERY, B, k), device=dev, dtype=torch.int64) for _ in range(2)]
        m_scores = torch.empty((GATHER_EVERY * B, k), device=dev)
        m_rows = torch.empty((GATHER_EVERY * B, k), device=dev, dtype=torch.int64)
    pending = {"n": 0, "ring": 0, "ev": [None, None]}
    use_hint = args.pipeline == 2 and not args.no_hint

    def flush():
        """all-gather the local keys of the pending batches (one RCCL call) and merge them on the GPU"""
        for s in streams:
            idx.search_flush_device(s.cuda_stream)      # deferred tails of earlier searches run / become ordered on s
        if not use_comm or pending["n"] == 0:
            return
        r = pending["ring"]
        for s in streams:
            comm_stream.wait_stream(s)
        with torch.cuda.stream(comm_stream):
            dist.all_gather_into_tensor(gathered[r], ring[r])
            merged_in = gathered[r].permute(1, 2, 0, 3).contiguous()   # [G][B][world][k]
            nat.merge_keys_device(merged_in, world * k, GATHER_EVERY * B, k, m_scores, m_rows, None, comm_stream.cuda_stream)
            pending["ev"][r] = comm_stream.record_event()
        pending["n"] = 0
        pending["ring"] = 1 - r
        # the ring refilled next was read by the gather issued GATHER_EVERY batches ago: wait for that one only
        ev = pending["ev"][1 - r]
        if ev is not None:
            for s in streams:
                s.wait_event(ev)

    def step(i: int) -> None:
        j = i % N_QUERY_BATCHES
        s = streams[i % len(streams)]
        o = slots[j]
        keys = ring[pending["ring"]][pending["n"]] if use_comm else o["keys"]
        if use_hint:   # the batch this stream searches next: its queries are prepared by extra workgroups of this launch
            idx.search_hint_next_device(queries[(i + len(streams)) % N_QUERY_BATCHES], B, s.cuda_stream)
        idx.search_device(queries[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], keys, o["status"], s.cuda_stream)
        if use_comm:
            pending["n"] += 1
            if pending["n"] == GATHER_EVERY:
                flush()

    status_host = torch.zeros((N_QUERY_BATCHES, B), dtype=torch.int32).pin_memory()
    status_np = status_host.numpy()          # (view made once, outside the timed region)

    def finish():
        """End of a timed run: the certificate status of every slot rides home in stream order (one 4 KB copy into pinned
        memory behind the last tail) and ONE synchronisation follows -- instead of synchronise, blocking copy, synchronise,
        whose host latencies (~300 us in all, kernel timeline in profiles/r02_trace20_gaps.txt) a 20-step run cannot amortise."""
        last = comm_stream if use_comm else streams[0]
        for s in streams:
            if s is not last:
                last.wait_stream(s)
        with torch.cuda.stream(last):
            status_host.copy_(status_all, non_blocking=True)
        torch.cuda.synchronize()

    def fixup_all() -> int:
        """certificate check of every slot (inside the timed region): repairs uncertified queries exactly"""
        fixed = 0
        nslots = min(N_QUERY_BATCHES, args.steps)
        bad = status_np[:nslots].any(axis=1)                          # (finish() has copied and synchronised)
        for j in np.nonzero(bad)[0].tolist():
            o = slots[j]
            fixed += idx.search_fixup_device(queries[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
        if use_comm:
            t = torch.tensor([fixed], device=dev, dtype=torch.int64)
            dist.all_reduce(t)
            if int(t.item()) > 0:
                # rare: some shard repaired a query after its keys were gathered -> redo those batches synchronously
                one = torch.zeros((world, B, k), device=dev, dtype=torch.int64)
                for j in range(nslots):
                    o = slots[j]
                    idx.search_device(queries[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
                    idx.search_fixup_device(queries[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
                    dist.all_gather_into_tensor(one, o["keys"])
                    nat.merge_keys_device(one.permute(1, 0, 2).contiguous(), world * k, B, k, o["scores"], o["rows"], None, 0)
        return fixed

    def sync_all():
        torch.cuda.synchronize()
        if use_comm:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not warm-up steps: a process that has only generated its corpus so far runs its first ~100 launches
    # 10-15 % slower (measured: scan launch 278 us in a 64-step run after 8 warm-up steps, 244 us in a 400-step run).
    # Keep the GPU busy with the same launches for 0.1 s first, so that short --steps/--warmup runs measure the same
    # steady state as long ones (16 steps after 2 warm-up steps: 206 k queries/s without, 240 k with; 400 steps: 260 k
    # vs 263 k).  RQ_BENCH_PREHEAT_MS=0 switches it off.
    preheat_ms = float(os.environ.get("RQ_BENCH_PREHEAT_MS", "100"))
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < preheat_ms:     # rank-local: no collective in here (the ranks' clocks differ)
        for i in range(16):
            o = slots[i % N_QUERY_BATCHES]
            idx.search_device(queries[i % N_QUERY_BATCHES], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"],
                              streams[i % len(streams)].cuda_stream)
        for s in streams:
            idx.search_flush_device(s.cuda_stream)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    flush()
    finish()        # the end-of-run sequence once before the timed region: the first numpy reduction / nonzero of a process cost
    fixup_all()     # ~90 us of lazy initialisation on the host (measured), which a 20-step run would otherwise carry
    sync_all()
    # HIP events around every scan launch of the timed region (the roofline's live figure).  They cost ~5 us per
    # step, which is 2 % at one GPU and 10 % of a 125k-row shard's step: at N > 1 the timed region runs without
    # them and the roofline comes from the calibration launches after it.
    live_events = not use_comm or os.environ.get("RQ_BENCH_LIVE_EVENTS") == "1"
    idx.set_option("profile", 1 if live_events else 0)
    idx.set_option("profile_stride", max(1, args.event_stride))
    idx.reset_timing()
    scan8_before = int(idx.get_option("scan8_used"))
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_enq = time.perf_counter()
    flush()
    finish()
    t_fin = time.perf_counter()
    fixed = fixup_all()
    if use_comm or fixed:
        sync_all()      # (at one GPU with nothing repaired, finish() ended with torch.cuda.synchronize() and nothing was enqueued since:
    elapsed = time.perf_counter() - t0   # that synchronisation IS the closing bracket; a second one costs ~100 us of host time on an idle device)
    host_phases = {"enqueue_all_steps_us": (t_enq - t0) * 1e6, "flush_copy_sync_us": (t_fin - t_enq) * 1e6,
                   "status_check_and_final_sync_us": (t0 + elapsed - t_fin) * 1e6}
    if use_comm:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = idx.timing()
    int8_scan = int(idx.get_option("scan8_used")) - scan8_before == args.steps      # every timed search scanned the int8 image
    # Results of the TIMED path, copied out before anything else runs: the calibration below re-uses the same slots with
    # the plain (pipeline 0) path, and the exactness checks further down must describe the kernel that was timed.
    nslots_used = min(N_QUERY_BATCHES, args.steps)
    timed_rows = [slots[j]["rows"].cpu().numpy().copy() for j in range(nslots_used)]
    timed_scores = [slots[j]["scores"].cpu().numpy().copy() for j in range(nslots_used)]
    timed_status = status_all[:nslots_used].cpu().numpy().copy()
    # calibration outside the timed region: the same launches on ONE stream (no overlap with a second scan), so the
    # stand-alone duration of the kernel can be read next to the live one
    idx.set_option("pipeline", 0)             # plain scan kernel, tail after it: the scan's stand-alone duration
    idx.set_option("profile", 1)
    idx.set_option("profile_stride", 1)
    idx.reset_timing()
    for i in range(24):
        o = slots[i % N_QUERY_BATCHES]
        idx.search_device(queries[i % N_QUERY_BATCHES], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"],
                          streams[0].cuda_stream)
    idx.search_flush_device(streams[0].cuda_stream)
    torch.cuda.synchronize()
    iso = idx.timing()
    iso_us = iso["scan_ms"] * 1e3 / max(iso["scan_launches"], 1)
    idx.set_option("profile", 0)

    qps = args.steps * B / elapsed
    if not live_events:
        timing = iso      # no per-launch events in the timed region: report the calibration launches
    scan_us = timing["scan_ms"] * 1e3 / max(timing["scan_launches"], 1)
    # bytes the scan kernel must read per launch: one pass over the fp16 shard (N * 768 * 2), or over its
    # int8 image (N * 768) when the int8 scan is in use -- the library reports which (rq_timing.scan_bytes)
    algo_bytes = timing["scan_bytes"] // max(timing["scan_launches"], 1) if timing["scan_launches"] else n_local * DIM * (1 if int8_scan else 2)
    achieved = algo_bytes / (scan_us * 1e-6) / 1e9 if scan_us > 0 else 0.0
    # HBM traffic per launch is a PMC figure (rocprofv3 --pmc passes, tools/run_profiles.sh): it cannot be collected
    # inside this process, so the line carries the committed measurement and says where it comes from.
    traffic, traffic_source = None, None
    for name in (("r02_pmc_scan8.json",) if int8_scan else ("r02_pmc_scan.json", "r01_pmc_scan.json")):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if world == 1 and n_total == N_ROWS and os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
                traffic_source = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this command, earlier run; not measured by this process)"
                break
            except Exception:
                traffic = None

    out = {
        "metric": "queries/sec @ Recall@10=1.0 (exact top-10), 1M x 768 fp16 corpus, batch-64 queries",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "i8" if int8_scan else "f16",
        "data": "synthetic",
        "config": {"workload": f"{n_total}x{DIM} fp16 corpus, batch-{B} queries, top-{k}, cosine, exact (certified) results",
                   "scan": ("int8 image of the fp16 shard (per-row scales, i8 matrix cores, exact int32 sums); candidates re-scored from the fp16 rows in fp64, "
                            "certificate from the measured quantisation error") if int8_scan else "fp16 rows (f16 matrix cores); candidates re-scored in fp64",
                   "rows_per_gpu": n_local, "streams": len(streams), "pipeline": args.pipeline, "next_batch_hint": use_hint, "gather_every": GATHER_EVERY if use_comm else 0,
                   "parallelism": f"row-shard x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "kernel": "rq_scan_tail_kernel" if args.pipeline == 2 else "rq_scan_kernel",
                     "avg_launch_us": scan_us, "launches": timing["scan_launches"], "algorithmic_bytes_per_launch": algo_bytes,
                     "scanned": "int8 image, 768 B per row" if int8_scan else "fp16 rows, 1536 B per row",
                     "measured": f"HIP events around every {max(1, args.event_stride)}-th scan launch of the timed region, on the stream it is launched on" if live_events else
                                 "HIP events around 24 single-stream launches right after the timed region (N > 1: no events inside it)",
                     "isolated": {"avg_launch_us": iso_us, "achieved": algo_bytes / (iso_us * 1e-6) / 1e9 if iso_us > 0 else 0.0,
                                  "frac": (algo_bytes / (iso_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if iso_us > 0 else 0.0,
                                  "note": "rq_scan_kernel alone (pipeline 0), 24 launches on one stream right after the timed region.  With one "
                                          "caller stream (the N = 1 default) the live figure is the same scan with the previous batch's "
                                          "tail workgroups riding along; with two caller streams "
                                          "a live interval would start when the launch is dequeued and include the wait for the "
                                          "other stream's scan to release its workgroup slots (queue wait + execution)"}},
        "repaired_queries": fixed,
        "host_phases": host_phases,
        "exact_scans": timing["exact_scans"],
    }

    # ---- outside the timed region: the same loop over the fp16 rows (the scan of SURVEY 8d's byte count), for comparison ----
    if world == 1 and int8_scan:
        idx.set_option("scan8", 0)
        idx.set_option("pipeline", args.pipeline)
        nsteps = min(args.steps, 200)
        for i in range(32):
            step(i)
        flush(); finish()
        idx.set_option("profile", 1); idx.set_option("profile_stride", max(1, args.event_stride)); idx.reset_timing()
        torch.cuda.synchronize()
        tf = time.perf_counter()
        for i in range(nsteps):
            step(i)
        flush(); finish()
        ef = time.perf_counter() - tf
        tm = idx.timing()
        f_us = tm["scan_ms"] * 1e3 / max(tm["scan_launches"], 1)
        f_bytes = n_local * DIM * 2
        out["fp16_scan"] = {"value": nsteps * B / ef, "unit": "queries/s", "ms_per_step": ef / nsteps * 1e3, "steps": nsteps,
                            "uncertified": int(status_np[:min(N_QUERY_BATCHES, nsteps)].sum()),
                            "roofline": {"bound": "hbm", "achieved": f_bytes / (f_us * 1e-6) / 1e9 if f_us > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": (f_bytes / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if f_us > 0 else 0.0,
                                         "avg_launch_us": f_us, "algorithmic_bytes_per_launch": f_bytes},
                            "note": "same loop with option scan8 = 0 (bench.py --scan fp16 times it as the main figure): every launch reads the fp16 rows"}
        idx.set_option("profile", 0)
        idx.set_option("scan8", {"auto": 1, "fp16": 0, "int8": 2}[args.scan])

    # ---- outside the timed region: recall vs the oracle, CPU baseline (rank 0, N = 1 only) ----------
    if world == 1:
        # (a) every slot of the timed path against the library's own exact route (every row of the shard re-scored in
        #     fp64, no approximate scan, no certificate: option slack_bins >= number of bins) -- no oracle involved
        idx.set_option("pipeline", 0)
        idx.set_option("slack_bins", max(n_local, 64))
        e_sc = torch.empty((B, k), device=dev); e_rw = torch.empty((B, k), device=dev, dtype=torch.int64)
        e_st = torch.empty((B,), device=dev, dtype=torch.int32)
        same_rows, max_diff = True, 0.0
        for j in range(nslots_used):
            idx.search_device(queries[j], B, k, nat.METRIC_COSINE, e_sc, e_rw, None, e_st, 0)
            torch.cuda.synchronize()
            same_rows = same_rows and bool((e_rw.cpu().numpy() == timed_rows[j]).all())
            max_diff = max(max_diff, float(np.abs(e_sc.cpu().numpy() - timed_scores[j]).max()))
        idx.set_option("slack_bins", -1)
        out["timed_path_vs_exact_fp64_scan"] = {"slots": nslots_used, "queries": nslots_used * B, "ids_match": same_rows,
                                                "max_abs_score_diff": max_diff, "uncertified_after_fixup": int(timed_status.sum())}
        # (b) the reference-shaped blocking call (host buffers in, host buffers out: what DenseIndex.search uses), reported
        #     beside `value`, never as `value`
        q_host0 = queries[0].cpu().numpy()
        host_api = {}
        for hb in (B, 1):
            idx.search(q_host0[:hb], k)
            tb = time.perf_counter()
            reps = 50
            for _ in range(reps):
                idx.search(q_host0[:hb], k)
            dt = (time.perf_counter() - tb) / reps
            host_api[f"batch_{hb}"] = {"us_per_call": dt * 1e6, "queries_per_s": hb / dt}
        host_api["note"] = "rq_search: pinned staging, H2D of the queries, search, one D2H of rows+scores+status, one synchronisation per call"
        out["host_api"] = host_api
    if world == 1 and not args.no_cpu_baseline:
        from oracle import dense_oracle as orc
        x16 = idx.get_rows_f16(0, n_local)
        # (c) the oracle: ALL queries of the first two slots of the TIMED (fused) path
        nchk_slots = min(2, nslots_used)
        q_chk = np.concatenate([queries[j].cpu().numpy() for j in range(nchk_slots)], 0)
        gs, gr = orc.dense_topk(q_chk, x16, k)
        got_r = np.concatenate(timed_rows[:nchk_slots], 0)
        got_s = np.concatenate(timed_scores[:nchk_slots], 0)
        out["recall_at_10"] = orc.recall_at_k(got_r, gr)
        out["ids_exact"] = bool((got_r == gr).all())
        out["max_abs_score_err"] = float(np.abs(got_s - gs).max())
        out["oracle_checked"] = f"all {nchk_slots * B} queries of result slots 0..{nchk_slots - 1} as written by the timed launches (copied out before the calibration launches)"
        q_host = queries[0].cpu().numpy()
        cores = len(os.sched_getaffinity(0))
        cpu = orc.Fp32BruteForce(x16)
        del x16
        # two ports of the same brute force: one big multi-threaded GEMM + numpy argpartition, or row slices on a
        # thread pool (GEMM and top-k both parallel).  Time one batch of each, keep the faster as the baseline.
        variants = {"blas": cpu.search_blas, "row-slices x%d threads" % cpu.n_threads: cpu.search}
        trial = {}
        for name, fn in variants.items():
            fn(q_host, k)
            tb = time.perf_counter()
            fn(q_host, k)
            trial[name] = time.perf_counter() - tb
        best = min(trial, key=trial.get)
        fn = variants[best]
        nb, tcpu = 0, 0.0
        while tcpu < 12.0 and nb < 256:
            qh = queries[(nb + 1) % N_QUERY_BATCHES].cpu().numpy()
            tb = time.perf_counter()
            fn(qh, k)
            tcpu += time.perf_counter() - tb
            nb += 1
        # the reference's own call pattern: one query per call (streaming_index.py:338-370), a few calls
        n1, t1 = 0, 0.0
        while t1 < 2.0 and n1 < 16:
            tb = time.perf_counter()
            fn(q_host[n1 % B: n1 % B + 1], k)
            t1 += time.perf_counter() - tb
            n1 += 1
        out["cpu_baseline"] = {"value": nb * B / tcpu, "unit": "queries/s", "cores": cores, "kind": "port",
                               "one_query_per_call": {"value": n1 / t1, "unit": "queries/s", "calls": n1},
                               "sample": f"{nb} batches of {B} queries over the full {n_local}x{DIM} corpus, oracle/dense_oracle.py "
                                         f"Fp32BruteForce variant '{best}' (fp32 OpenBLAS GEMM + argpartition; one-batch trials: "
                                         + ", ".join(f"{n_}: {t_ * 1e3:.0f} ms" for n_, t_ in trial.items()) + ")"}
    if use_comm:
        # N > 1 correctness evidence (outside the timed region): the benchmarked path = approximate scan + certified
        # re-score per shard -> RCCL all-gather of keys -> device merge, against the library's own EXACT path (every row
        # of the shard re-scored in fp64, no approximation, no certificate: option slack_bins >= number of bins) ->
        # all-gather -> host merge.  (The oracle-based version of this check runs in tests/ and, at N = 1, above.)
        from rag_uq_amd import distributed as rqd
        nchk = 4
        o = slots[0]
        idx.search_device(queries[0], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
        idx.search_fixup_device(queries[0], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
        allk = torch.zeros((world, B, k), device=dev, dtype=torch.int64)
        dist.all_gather_into_tensor(allk, o["keys"])
        g_scores = torch.empty((B, k), device=dev)
        g_rows = torch.empty((B, k), device=dev, dtype=torch.int64)
        nat.merge_keys_device(allk.permute(1, 0, 2).contiguous(), world * k, B, k, g_scores, g_rows, None, 0)
        torch.cuda.synchronize()
        idx.set_option("slack_bins", max(n_local, 64))          # -> the exact fp64 scan of the whole shard
        e_sc = torch.empty((nchk, k), device=dev); e_rw = torch.empty((nchk, k), device=dev, dtype=torch.int64)
        e_ky = torch.zeros((nchk, k), device=dev, dtype=torch.int64); e_st = torch.empty((nchk,), device=dev, dtype=torch.int32)
        if n_local:
            idx.search_device(queries[0][:nchk].contiguous(), nchk, k, nat.METRIC_COSINE, e_sc, e_rw, e_ky, e_st, 0)
        oall = torch.zeros((world, nchk, k), device=dev, dtype=torch.int64)
        dist.all_gather_into_tensor(oall, e_ky)
        ms, mr = rqd.merge_keys_host(oall.permute(1, 0, 2).reshape(nchk, world * k).cpu().numpy().view(np.uint64), k)
        got_r = g_rows[:nchk].cpu().numpy()
        out["ids_match_exact_fp64_scan"] = bool((got_r == mr).all())
        out["recall_at_10_vs_exact_fp64_scan"] = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / max(len(b), 1) for a, b in zip(got_r, mr)]))
        out["max_abs_score_diff_vs_exact_fp64_scan"] = float(np.abs(g_scores[:nchk].cpu().numpy() - ms).max())
    if rank == 0:
        print(json.dumps(out), flush=True)
    idx.close()
    if use_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
