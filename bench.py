#!/usr/bin/env python3
"""bench.py -- queries/sec of the dense-retrieval hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): 1M x 768 fp16 synthetic corpus, batches of 64 queries, top-10,
exact results.  A "step" = one 64-query batch searched against the whole corpus.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: the corpus is row-sharded (rank r holds rows [r*N/G, (r+1)*N/G)), every rank searches every
batch on its shard, the per-shard (score, row) keys are exchanged with one RCCL all-gather per
GATHER_EVERY batches and merged on the GPU.  Total work is fixed as N grows => "scaling": "strong".

Rank 0 prints ONE JSON line.  `value`, `dtype` and `roofline` describe the scan over the fp16 ROWS (N * 1536 B per
batch) at every N -- `--scan fp16` is the default and the operand is named in config.workload / config.scan.  Extra objects:
  "roofline"      scan kernel: algorithmic bytes / HIP-event duration measured here, over the timed region
  "cpu_baseline"  N = 1: oracle port, fp32 OpenBLAS brute force on this box's host cores, bounded sample
  "int8_scan"     N = 1: the same loop with the library's int8 image of the shard as the scan operand (768 B per row read, +768 B
                  per row of HBM; every returned row is still re-scored in fp64 from the fp16 rows and certified): its own value,
                  roofline, traffic and oracle check
  "k100"          N = 1: the same loop at k = 100 (SURVEY 8(d) quotes configs[1] at k = 10 and k = 100), both operands
  "structured"    N = 1: both operands on document-structured and clustered 1M-row corpora (the int8 scan's data dependence)
  "host_api"      N = 1: the blocking host-buffer calls (rq_search, DenseIndex.search_vectors) beside `value`
`--scan int8` / `--scan auto` make the int8 image / the library's own rule the main figure instead (dtype then says "i8").
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
SCAN_OPT = {"auto": 1, "fp16": 0, "int8": 2}
SCAN_TEXT = {False: "fp16 rows (f16 matrix cores), 1536 B per row; candidates re-scored in fp64",
             True: "int8 image of the fp16 shard (per-row scales, i8 matrix cores, exact int32 sums), 768 B per row; candidates re-scored "
                   "from the fp16 rows in fp64, certificate from the measured quantisation error"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--rows", type=int, default=N_ROWS)
    ap.add_argument("--k", type=int, default=TOPK)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the int8_scan / structured / host_api legs that follow the timed region at N = 1")
    ap.add_argument("--streams", type=int, default=0, help="caller streams the batches alternate over (0 = 1 at one GPU, 2 at several)")
    ap.add_argument("--event-stride", type=int, default=4, help="HIP events around every n-th scan launch of the timed region")
    ap.add_argument("--scan", choices=("auto", "fp16", "int8"), default="fp16",
                    help="corpus operand of the scan for `value` (the same at every N): fp16 rows (default: BASELINE.json configs[1], SURVEY 8(d)'s "
                         "1536 B per row), their int8 image (half the bytes; candidates are still re-scored from the fp16 rows in fp64), or the "
                         "library's rule")
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
    # RQ_BENCH_BACKEND=gloo (rehearsal only): several ranks on ONE GPU -- RCCL refuses two ranks on one device -- exchanging their keys through
    # host memory.  Everything else is the N > 1 path unchanged (row shards, per-rank searches, merge, the exact-scan cross-check); timings
    # of such a run mean nothing, its results (`ids_match_exact_fp64_scan`, `ranks_seen`) do.
    backend = os.environ.get("RQ_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RQ_BENCH_FORCE_COMM=1 exercises the all-gather + merge path on a single rank (rehearsal on a 1-GPU box)
    use_comm = world > 1 or os.environ.get("RQ_BENCH_FORCE_COMM") == "1"
    ranks_seen = 1
    if use_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ranks_seen = dist.get_world_size()

    def all_gather_keys(dst, src):
        """dst [world, ...] <- every rank's src (RCCL over xGMI; through host memory in the gloo rehearsal)"""
        if backend != "gloo":
            dist.all_gather_into_tensor(dst, src)
            return
        torch.cuda.current_stream().synchronize()
        parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)]
        dist.all_gather(parts, src.cpu())
        dst.copy_(torch.stack(parts).to(dst.device))

    def all_reduce_scalar(value, op=None, dtype=None):
        t = torch.tensor([value], device="cpu" if backend == "gloo" else dev, dtype=dtype or torch.int64)
        dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
        return t.item()

    n_total, k, B = args.rows, args.k, BATCH
    n_chunks = (n_total + CHUNK_ROWS - 1) // CHUNK_ROWS
    c_lo, c_hi = rank * n_chunks // world, (rank + 1) * n_chunks // world
    row_lo = min(c_lo * CHUNK_ROWS, n_total)
    row_hi = min(c_hi * CHUNK_ROWS, n_total)
    n_local = row_hi - row_lo

    # One GPU: one caller stream, so scan launches never overlap each other and the per-launch HIP events (and
    # rocprofv3) read the kernel's own duration; the tail of batch i hides inside the scan launch of batch i+1
    # (pipeline 2).  Several GPUs (125k-row shards, ~30 us steps): two caller streams also hide the query prep
    # and the launch gaps (measured 34.8 vs 50.1 us per step); per-launch events are off there.
    if args.streams <= 0:
        args.streams = 2 if use_comm else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    comm_stream = torch.cuda.Stream(device=dev)
    use_hint = args.pipeline == 2 and not args.no_hint
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = [torch.randn((B, DIM), device=dev, generator=gq, dtype=torch.float32) for _ in range(N_QUERY_BATCHES)]

    class Loop:
        """The timed loop over one index: per-batch output slots that stay in HBM, the enqueue / flush / finish / repair steps."""

        def __init__(self, idx, qs, k=k):
            self.idx, self.qs, self.k = idx, qs, k
            self.status_all = torch.zeros((N_QUERY_BATCHES, B), device=dev, dtype=torch.int32)   # one read-back checks every slot
            self.slots = [dict(scores=torch.empty((B, k), device=dev), rows=torch.empty((B, k), device=dev, dtype=torch.int64),
                               keys=torch.empty((B, k), device=dev, dtype=torch.int64), status=self.status_all[j]) for j in range(N_QUERY_BATCHES)]
            self.status_host = torch.zeros((N_QUERY_BATCHES, B), dtype=torch.int32).pin_memory()
            self.status_np = self.status_host.numpy()          # (view made once, outside the timed region)
            self.pending = {"n": 0, "ring": 0, "ev": [None, None]}
            self.trains = None
            if use_comm:
                self.ring = [torch.zeros((GATHER_EVERY, B, k), device=dev, dtype=torch.int64) for _ in range(2)]
                self.gathered = [torch.zeros((world, GATHER_EVERY, B, k), device=dev, dtype=torch.int64) for _ in range(2)]
                self.m_scores = torch.empty((GATHER_EVERY * B, k), device=dev)
                self.m_rows = torch.empty((GATHER_EVERY * B, k), device=dev, dtype=torch.int64)
                # N > 1: the GATHER_EVERY searches between two all-gathers are enqueued by ONE library call (rq_search_train_device):
                # a 125k-row shard answers a batch in ~25 us, a Python loop with two ctypes calls per batch costs 7-8 us of that
                if GATHER_EVERY == N_QUERY_BATCHES and GATHER_EVERY % len(streams) == 0 and use_hint and os.environ.get("RQ_BENCH_TRAIN", "1") == "1":
                    sl = self.slots
                    self.trains = [idx.make_train(qs, [o["scores"] for o in sl], [o["rows"] for o in sl], [self.ring[r][j] for j in range(GATHER_EVERY)],
                                                  [o["status"] for o in sl], [st.cuda_stream for st in streams], qs[:len(streams)]) for r in range(2)]

        def flush(self):
            """all-gather the local keys of the pending batches (one RCCL call) and merge them on the GPU"""
            k = self.k
            idx, pending = self.idx, self.pending
            for s in streams:
                idx.search_flush_device(s.cuda_stream)      # deferred tails of earlier searches run / become ordered on s
            if not use_comm or pending["n"] == 0:
                return
            r = pending["ring"]
            for s in streams:
                comm_stream.wait_stream(s)
            with torch.cuda.stream(comm_stream):
                all_gather_keys(self.gathered[r], self.ring[r])
                merged_in = self.gathered[r].permute(1, 2, 0, 3).contiguous()   # [G][B][world][k]
                nat.merge_keys_device(merged_in, world * k, GATHER_EVERY * B, k, self.m_scores, self.m_rows, None, comm_stream.cuda_stream)
                pending["ev"][r] = comm_stream.record_event()
            pending["n"] = 0
            pending["ring"] = 1 - r
            # the ring refilled next was read by the gather issued GATHER_EVERY batches ago: wait for that one only
            ev = pending["ev"][1 - r]
            if ev is not None:
                for s in streams:
                    s.wait_event(ev)

        def step(self, i: int) -> None:
            k = self.k
            idx, pending = self.idx, self.pending
            j = i % N_QUERY_BATCHES
            s = streams[i % len(streams)]
            o = self.slots[j]
            keys = self.ring[pending["ring"]][pending["n"]] if use_comm else o["keys"]
            if use_hint:   # the batch this stream searches next: its queries are prepared by extra workgroups of this launch
                idx.search_hint_next_device(self.qs[(i + len(streams)) % N_QUERY_BATCHES], B, s.cuda_stream)
            idx.search_device(self.qs[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], keys, o["status"], s.cuda_stream)
            if use_comm:
                pending["n"] += 1
                if pending["n"] == GATHER_EVERY:
                    self.flush()

        def run_steps(self, n: int) -> None:
            k = self.k
            i = 0
            while i < n:
                if self.trains is not None and self.pending["n"] == 0 and i % N_QUERY_BATCHES == 0 and n - i >= GATHER_EVERY:
                    self.idx.search_train_device(self.trains[self.pending["ring"]], B, k, nat.METRIC_COSINE)
                    self.pending["n"] = GATHER_EVERY
                    self.flush()
                    i += GATHER_EVERY
                else:
                    self.step(i)
                    i += 1

        def finish(self):
            """End of a timed run: the certificate status of every slot rides home in stream order (one 4 KB copy into pinned
            memory behind the last tail) and ONE synchronisation follows -- instead of synchronise, blocking copy, synchronise,
            whose host latencies (~300 us in all, kernel timeline in profiles/r02_trace20_gaps.txt) a 20-step run cannot amortise."""
            last = comm_stream if use_comm else streams[0]
            for s in streams:
                if s is not last:
                    last.wait_stream(s)
            with torch.cuda.stream(last):
                self.status_host.copy_(self.status_all, non_blocking=True)
            torch.cuda.synchronize()

        def fixup_all(self, nsteps: int) -> int:
            """certificate check of every slot (inside the timed region): repairs uncertified queries exactly"""
            k = self.k
            idx, fixed = self.idx, 0
            nslots = min(N_QUERY_BATCHES, nsteps)
            bad = self.status_np[:nslots].any(axis=1)                          # (finish() has copied and synchronised)
            for j in np.nonzero(bad)[0].tolist():
                o = self.slots[j]
                fixed += idx.search_fixup_device(self.qs[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
            if use_comm:
                if int(all_reduce_scalar(fixed)) > 0:
                    # rare: some shard repaired a query after its keys were gathered -> redo those batches synchronously
                    one = torch.zeros((world, B, k), device=dev, dtype=torch.int64)
                    for j in range(nslots):
                        o = self.slots[j]
                        idx.search_device(self.qs[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
                        idx.search_fixup_device(self.qs[j], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
                        all_gather_keys(one, o["keys"])
                        nat.merge_keys_device(one.permute(1, 0, 2).contiguous(), world * k, B, k, o["scores"], o["rows"], None, 0)
            return fixed

        def preheat(self):
            k = self.k
            # Set-up, not warm-up steps: a process that has only generated its corpus so far runs its first ~100 launches
            # 10-15 % slower (measured: scan launch 278 us in a 64-step run after 8 warm-up steps, 244 us in a 400-step run).
            # Keep the GPU busy with the same launches for 0.1 s first, so that short --steps/--warmup runs measure the same
            # steady state as long ones (16 steps after 2 warm-up steps: 206 k queries/s without, 240 k with; 400 steps: 260 k
            # vs 263 k).  RQ_BENCH_PREHEAT_MS=0 switches it off.
            idx = self.idx
            preheat_ms = float(os.environ.get("RQ_BENCH_PREHEAT_MS", "100"))
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < preheat_ms:     # rank-local: no collective in here (the ranks' clocks differ)
                for i in range(16):
                    o = self.slots[i % N_QUERY_BATCHES]
                    idx.search_device(self.qs[i % N_QUERY_BATCHES], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"],
                                      streams[i % len(streams)].cuda_stream)
                for s in streams:
                    idx.search_flush_device(s.cuda_stream)
                torch.cuda.synchronize()

        def timed(self, scan: str, steps: int, warmup: int, live_events: bool) -> dict:
            """W warm-up steps, then EXACTLY `steps` timed steps bracketed by barrier + synchronize; results of the timed path are
            copied out before anything else runs."""
            idx = self.idx
            idx.set_option("slack_bins", -1)
            idx.set_option("pipeline", args.pipeline)
            idx.set_option("scan8", SCAN_OPT[scan])
            idx.set_option("profile", 0)
            self.preheat()
            if use_comm:      # the communicator and the all-gather path are warm before the timed region whatever --warmup is (a run with --warmup 0
                with torch.cuda.stream(comm_stream):      # would otherwise pay RCCL's lazy set-up inside it)
                    all_gather_keys(self.gathered[0], self.ring[0])
                comm_stream.synchronize()
            self.run_steps(warmup)
            self.flush()
            self.finish()        # the end-of-run sequence once before the timed region: the first numpy reduction / nonzero of a process cost
            self.fixup_all(max(warmup, 1))     # ~90 us of lazy initialisation on the host (measured), which a 20-step run would otherwise carry
            sync_all()
            # HIP events around every n-th scan launch of the timed region (the roofline's live figure).  They cost ~5 us per
            # step, which is 2 % at one GPU and 10 % of a 125k-row shard's step: at N > 1 the timed region runs without
            # them and the roofline comes from the calibration launches after it.
            idx.set_option("profile", 1 if live_events else 0)
            idx.set_option("profile_stride", max(1, args.event_stride))
            idx.reset_timing()
            scan8_before = int(idx.get_option("scan8_used"))
            sync_all()
            t0 = time.perf_counter()
            self.run_steps(steps)
            t_enq = time.perf_counter()
            self.flush()
            self.finish()
            t_fin = time.perf_counter()
            fixed = self.fixup_all(steps)
            if use_comm or fixed:
                sync_all()      # (at one GPU with nothing repaired, finish() ended with torch.cuda.synchronize() and nothing was enqueued since:
            elapsed = time.perf_counter() - t0   # that synchronisation IS the closing bracket; a second one costs ~100 us of host time on an idle device)
            r = {"host_phases": {"enqueue_all_steps_us": (t_enq - t0) * 1e6, "flush_copy_sync_us": (t_fin - t_enq) * 1e6,
                                 "status_check_and_final_sync_us": (t0 + elapsed - t_fin) * 1e6}}
            if use_comm:
                elapsed = float(all_reduce_scalar(elapsed, dist.ReduceOp.MAX, torch.float64))
            r["elapsed"], r["fixed"], r["steps"] = elapsed, fixed, steps
            r["timing"] = idx.timing()
            r["int8"] = int(idx.get_option("scan8_used")) - scan8_before == steps      # every timed search scanned the int8 image
            r["scan8_level"] = idx.get_option("scan8_level")
            idx.set_option("profile", 0)
            # Results of the TIMED path, copied out before anything else runs: the calibration below re-uses the same slots with
            # the plain (pipeline 0) path, and the exactness checks further down must describe the kernel that was timed.
            ns = min(N_QUERY_BATCHES, steps)
            r["rows"] = [self.slots[j]["rows"].cpu().numpy().copy() for j in range(ns)]
            r["scores"] = [self.slots[j]["scores"].cpu().numpy().copy() for j in range(ns)]
            r["status"] = self.status_all[:ns].cpu().numpy().copy()
            return r

        def calibrate(self) -> dict:
            """outside the timed region: the plain scan kernel (pipeline 0, tail after it) on ONE stream, an event pair on every launch"""
            k = self.k
            idx = self.idx
            idx.set_option("pipeline", 0)
            idx.set_option("profile", 1)
            idx.set_option("profile_stride", 1)
            idx.reset_timing()
            for i in range(24):
                o = self.slots[i % N_QUERY_BATCHES]
                idx.search_device(self.qs[i % N_QUERY_BATCHES], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], streams[0].cuda_stream)
            idx.search_flush_device(streams[0].cuda_stream)
            torch.cuda.synchronize()
            iso = idx.timing()
            idx.set_option("profile", 0)
            idx.set_option("pipeline", args.pipeline)
            return iso

        def exact_fp64_check(self, r: dict, nslots: int) -> dict:
            """slots of the timed path against the library's own exact route (every row of the shard re-scored in fp64, no approximate
            scan, no certificate: option slack_bins >= number of bins) -- no oracle involved"""
            k = self.k
            idx = self.idx
            idx.set_option("pipeline", 0)
            idx.set_option("slack_bins", max(len(idx), 64))
            e_sc = torch.empty((B, k), device=dev); e_rw = torch.empty((B, k), device=dev, dtype=torch.int64)
            e_st = torch.empty((B,), device=dev, dtype=torch.int32)
            same_rows, max_diff = True, 0.0
            nslots = min(nslots, len(r["rows"]))
            for j in range(nslots):
                idx.search_device(self.qs[j], B, k, nat.METRIC_COSINE, e_sc, e_rw, None, e_st, 0)
                torch.cuda.synchronize()
                same_rows = same_rows and bool((e_rw.cpu().numpy() == r["rows"][j]).all())
                max_diff = max(max_diff, float(np.abs(e_sc.cpu().numpy() - r["scores"][j]).max()))
            idx.set_option("slack_bins", -1)
            idx.set_option("pipeline", args.pipeline)
            return {"slots": nslots, "queries": nslots * B, "ids_match": same_rows, "max_abs_score_diff": max_diff,
                    "uncertified_after_fixup": int(r["status"][:nslots].sum())}

    def sync_all():
        torch.cuda.synchronize()
        if use_comm:
            dist.barrier()
        torch.cuda.synchronize()

    def roofline_of(r: dict, iso: dict, n_rows: int, live: bool) -> dict:
        """bytes the scan kernel must read per launch -- one pass over the fp16 shard (N * 768 * 2), or over its int8 image
        (N * 768) when the int8 scan is in use; the library reports which (rq_timing.scan_bytes) -- over the average HIP-event duration"""
        int8 = r["int8"]
        timing = r["timing"] if live else iso      # no per-launch events in the timed region: report the calibration launches
        scan_us = timing["scan_ms"] * 1e3 / max(timing["scan_launches"], 1)
        iso_us = iso["scan_ms"] * 1e3 / max(iso["scan_launches"], 1)
        algo = timing["scan_bytes"] // max(timing["scan_launches"], 1) if timing["scan_launches"] else n_rows * DIM * (1 if int8 else 2)
        achieved = algo / (scan_us * 1e-6) / 1e9 if scan_us > 0 else 0.0
        # HBM traffic per launch is a PMC figure (rocprofv3 --pmc passes, tools/run_profiles.sh): it cannot be collected
        # inside this process, so the line carries the committed measurement and says where it comes from.
        traffic, traffic_source = None, None
        for name in (("r03_pmc_scan8.json", "r02_pmc_scan8.json") if int8 else ("r03_pmc_scan.json", "r02_pmc_scan.json")):
            pmc_path = os.path.join(ROOT, "profiles", name)
            if world == 1 and n_rows == N_ROWS and os.path.exists(pmc_path):
                try:
                    traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this command, earlier run; not measured by this process)"
                    break
                except Exception:
                    traffic = None
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source, "kernel": "rq_scan_tail_kernel" if args.pipeline == 2 else "rq_scan_kernel",
                "avg_launch_us": scan_us, "launches": timing["scan_launches"], "algorithmic_bytes_per_launch": algo,
                "scanned": "int8 image, 768 B per row" if int8 else "fp16 rows, 1536 B per row",
                "measured": f"HIP events attached to every {max(1, args.event_stride)}-th scan dispatch of the timed region, on the stream it is launched on" if live else
                            "HIP events around 24 single-stream launches right after the timed region (N > 1: no events inside it)",
                "isolated": {"avg_launch_us": iso_us, "achieved": algo / (iso_us * 1e-6) / 1e9 if iso_us > 0 else 0.0,
                             "frac": (algo / (iso_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if iso_us > 0 else 0.0,
                             "note": "rq_scan_kernel alone (pipeline 0), 24 launches on one stream right after the timed region.  With one "
                                     "caller stream (the N = 1 default) the live figure is the same scan with the previous batch's "
                                     "tail workgroups riding along; with two caller streams a live interval would start when the launch "
                                     "is dequeued and include the wait for the other stream's scan to release its workgroup slots"}}

    # ---- corpus shard: synthetic Gaussian rows, unit norm, fp16, generated in HBM ---------------
    idx = nat.NativeIndex(DIM, local_rank)
    idx.reserve(max(n_local, 1))
    idx.set_row_offset(row_lo)
    for opt in filter(None, os.environ.get("RQ_BENCH_OPTS", "").split(",")):      # development: e.g. RQ_BENCH_OPTS=wg_per_cu=3
        name, val = opt.split("=")
        idx.set_option(name, float(val))
    for c in range(c_lo, c_hi):
        g = torch.Generator(device=dev)
        g.manual_seed(1235 + c)
        n = min(CHUNK_ROWS, n_total - c * CHUNK_ROWS)
        x = torch.randn((n, DIM), device=dev, generator=g, dtype=torch.float32)
        x = torch.nn.functional.normalize(x, dim=1).half().contiguous()
        idx.add_f16_device(x, n)
        del x
    torch.cuda.synchronize()

    # ---- the timed region: `value` ---------------------------------------------------------------------------------------------
    loop = Loop(idx, queries)
    live_events = not use_comm or os.environ.get("RQ_BENCH_LIVE_EVENTS") == "1"
    main_r = loop.timed(args.scan, args.steps, args.warmup, live_events)
    iso = loop.calibrate()
    int8_scan = main_r["int8"]
    elapsed = main_r["elapsed"]
    out = {
        "metric": "queries/sec @ Recall@10=1.0 (exact top-10), 1M x 768 fp16 corpus, batch-64 queries",
        "value": args.steps * B / elapsed,
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
        "config": {"workload": f"{n_total}x{DIM} fp16 corpus, batch-{B} queries, top-{k}, cosine, exact (certified) results; scan operand: "
                               + ("int8 image of the fp16 rows (768 B per row)" if int8_scan else "the fp16 rows (1536 B per row)"),
                   "scan": SCAN_TEXT[int8_scan], "scan_option": args.scan,
                   "rows_per_gpu": n_local, "streams": len(streams), "pipeline": args.pipeline, "next_batch_hint": use_hint, "gather_every": GATHER_EVERY if use_comm else 0,
                   "enqueue": "rq_search_train_device (one library call per gather period)" if loop.trains is not None else "rq_search_device per step",
                   "parallelism": f"row-shard x{world}", "ranks_seen": ranks_seen, "exchange": ("RCCL all-gather" if backend != "gloo" else "gloo through host memory (one-GPU rehearsal)") if use_comm else "none"},
        "roofline": roofline_of(main_r, iso, n_local, live_events),
        "repaired_queries": main_r["fixed"],
        "host_phases": main_r["host_phases"],
        "exact_scans": main_r["timing"]["exact_scans"],
    }

    # ---- outside the timed region (N = 1 only): exactness of the timed path, the other operand, structured corpora, host calls ----
    if world == 1:
        out["timed_path_vs_exact_fp64_scan"] = loop.exact_fp64_check(main_r, N_QUERY_BATCHES)
    int8_r = None
    if world == 1 and not args.no_extra_legs:
        # (a) the same loop over the other operand.  At --scan fp16 (default) this is the int8 image: half the bytes per pass, +768 B of
        #     HBM per row, and a data-dependent candidate count (see "structured" below)
        other = "fp16" if int8_scan else "int8"
        o_steps = min(args.steps, 400)
        o_r = loop.timed(other, o_steps, min(args.warmup, 40), True)
        o_iso = loop.calibrate()
        leg = {"value": o_steps * B / o_r["elapsed"], "unit": "queries/s", "ms_per_step": o_r["elapsed"] / o_steps * 1e3, "steps": o_steps,
               "dtype": "i8" if o_r["int8"] else "f16", "scan": SCAN_TEXT[o_r["int8"]],
               "roofline": roofline_of(o_r, o_iso, n_local, True), "repaired_queries": o_r["fixed"], "exact_scans": o_r["timing"]["exact_scans"],
               "scan8_level_after": o_r["scan8_level"],
               "timed_path_vs_exact_fp64_scan": loop.exact_fp64_check(o_r, N_QUERY_BATCHES)}
        if o_r["int8"]:
            leg["extra_hbm_bytes_per_row"] = 768
            leg["note"] = ("same loop with option scan8 = 2: every launch reads the int8 image the library keeps beside the fp16 rows (bench.py --scan int8 "
                           "times it as the main figure); the roofline counts the 768 B per row this kernel has to read, NOT SURVEY 8(d)'s 1536")
            int8_r = o_r
            out["int8_scan"] = leg
        else:
            leg["note"] = "same loop with option scan8 = 0 (bench.py --scan fp16, the default, times it as the main figure): every launch reads the fp16 rows"
            out["fp16_scan"] = leg
            int8_r = main_r

        # (a') SURVEY 8(d) quotes configs[1] at k = 10 AND k = 100: the same loop at k = 100 over both operands (their own slots; the
        #      k > 32 class of the int8 image starts on two images per query)
        if k != 100:
            loop100 = Loop(idx, queries, k=100)
            k100 = {"k": 100, "steps": 100, "warmup": 20}
            for scan in ("fp16", "int8"):
                r100 = loop100.timed(scan, 100, 20, True)
                t100 = r100["timing"]
                k100[scan] = {"value": 100 * B / r100["elapsed"], "unit": "queries/s", "ms_per_step": r100["elapsed"] / 100 * 1e3,
                              "scan_launch_us": t100["scan_ms"] * 1e3 / max(t100["scan_launches"], 1), "scanned_int8_image_in_every_timed_step": r100["int8"],
                              "repaired_queries": r100["fixed"], "exact_scans": t100["exact_scans"], "scan8_level_after": r100["scan8_level"],
                              "timed_path_vs_exact_fp64_scan": loop100.exact_fp64_check(r100, 2)}
            out["k100"] = k100
            del loop100

        # (b) the reference-shaped blocking calls (host buffers in, host buffers out), reported beside `value`, never as `value`
        idx.set_option("scan8", 1)          # the library's own rule, as a DenseIndex would run
        idx.set_option("pipeline", 0)
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
        host_api["note"] = ("rq_search: pinned staging, H2D of the queries, search, one D2H of rows+scores+status, one synchronisation per call; "
                            "operand by the library's rule (int8 image from 200k rows)")
        try:
            host_api["python_seam"] = python_seam(idx, q_host0, n_local)
        except Exception as e:   # the seam leg must never cost the headline line
            host_api["python_seam"] = {"error": repr(e)}
        out["host_api"] = host_api
        idx.set_option("pipeline", args.pipeline)

    if world == 1 and not args.no_cpu_baseline:
        from oracle import dense_oracle as orc
        x16 = idx.get_rows_f16(0, n_local)
        # (c) the oracle: ALL queries of the first two slots of the TIMED (fused) path -- and of the int8 leg
        nchk_slots = min(2, len(main_r["rows"]))
        q_chk = np.concatenate([queries[j].cpu().numpy() for j in range(nchk_slots)], 0)
        gs, gr = orc.dense_topk(q_chk, x16, k)

        def against_oracle(r):
            got_r = np.concatenate(r["rows"][:nchk_slots], 0)
            got_s = np.concatenate(r["scores"][:nchk_slots], 0)
            return {"recall_at_10": orc.recall_at_k(got_r, gr), "ids_exact": bool((got_r == gr).all()), "max_abs_score_err": float(np.abs(got_s - gs).max())}
        out.update(against_oracle(main_r))
        out["oracle_checked"] = f"all {nchk_slots * B} queries of result slots 0..{nchk_slots - 1} as written by the timed launches (copied out before the calibration launches)"
        for name in ("int8_scan", "fp16_scan"):
            if name in out:
                out[name].update(against_oracle(o_r))
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
        del cpu

    if use_comm:
        # N > 1 correctness evidence (outside the timed region): the benchmarked path = approximate scan + certified
        # re-score per shard -> RCCL all-gather of keys -> device merge, against the library's own EXACT path (every row
        # of the shard re-scored in fp64, no approximation, no certificate: option slack_bins >= number of bins) ->
        # all-gather -> host merge.  (The oracle-based version of this check runs in tests/ and, at N = 1, above.)
        from rag_uq_amd import distributed as rqd
        nchk = 4
        o = loop.slots[0]
        idx.search_device(queries[0], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
        idx.search_fixup_device(queries[0], B, k, nat.METRIC_COSINE, o["scores"], o["rows"], o["keys"], o["status"], 0)
        allk = torch.zeros((world, B, k), device=dev, dtype=torch.int64)
        all_gather_keys(allk, o["keys"])
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
        all_gather_keys(oall, e_ky)
        ms, mr = rqd.merge_keys_host(oall.permute(1, 0, 2).reshape(nchk, world * k).cpu().numpy().view(np.uint64), k)
        got_r = g_rows[:nchk].cpu().numpy()
        out["ids_match_exact_fp64_scan"] = bool((got_r == mr).all())
        out["recall_at_10_vs_exact_fp64_scan"] = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / max(len(b), 1) for a, b in zip(got_r, mr)]))
        out["max_abs_score_diff_vs_exact_fp64_scan"] = float(np.abs(g_scores[:nchk].cpu().numpy() - ms).max())
    idx.close()
    del loop, idx

    # ---- N = 1: both operands on structured corpora (after the main index is gone: one 1M-row corpus in HBM at a time) ----------
    if world == 1 and not args.no_extra_legs and n_total >= 200_000:
        try:
            out["structured"] = structured_leg(nat, torch, np, dev, n_total, k, Loop, sync_all)
        except Exception as e:
            out["structured"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_comm:
        dist.destroy_process_group()


def python_seam(idx, q_host, n_local):
    """The Python seam on top of the C ABI -- DenseIndex.search_vectors including the (doc_id, score, text) tuple assembly of
    reference streaming_index.py:361-368 -- for the two shapes callers use: a batch of 64 at k = 10, and ONE query at k = 50 (the
    reference's actual call, hybrid_search's pool, streaming_index.py:537)."""
    from rag_uq_amd.streaming_index import DenseIndex
    di = DenseIndex.from_native(idx, [f"d{i}" for i in range(n_local)])
    res = {}
    for name, hb, kk in (("batch_64_k10", 64, 10), ("batch_1_k50", 1, 50)):
        di.search_vectors(q_host[:hb], top_k=kk)
        reps = 30
        tb = time.perf_counter()
        for _ in range(reps):
            r = di.search_vectors(q_host[:hb], top_k=kk)
        dt = (time.perf_counter() - tb) / reps
        assert len(r) == hb and len(r[0]) == kk
        res[name] = {"us_per_call": dt * 1e6, "queries_per_s": hb / dt}
    res["note"] = "DenseIndex.search_vectors (host numpy queries -> list of (doc_id, score, text) tuples per query), texts not stored"
    return res


def structured_leg(nat, torch, np, dev, n, k, Loop, sync_all):
    """What the int8 operand depends on: 1M-row corpora that are NOT isotropic Gaussian (tools/gpu_small_shard.py's): passages in
    'documents' of 16 similar rows (a chunked Wikipedia), and 64 tight 'centroids' (SURVEY 8(d)'s clustered set), searched with
    random and with on-topic queries; 64 timed steps per operand after 48 warm-up steps whose repairs drive the library's ladder
    (one int8 image -> two -> fp16 rows).  Reported per operand: us per step INCLUDING repairs, uncertified queries of the timed
    steps, where the ladder ended up, and whether the timed results equal the library's exact fp64 scan."""
    B = BATCH
    g = torch.Generator(device=dev); g.manual_seed(7)
    cent = torch.randn((64, DIM), device=dev, generator=g)
    docs = torch.randn((n // 16 + 1, DIM), device=dev, generator=g)
    res = {"rows": n, "steps": 64, "warmup": 48, "k": k}
    for mode in ("documents", "centroids"):
        idx = nat.NativeIndex(DIM, dev.index or 0)
        idx.reserve(n)
        for lo in range(0, n, CHUNK_ROWS):          # in chunks: the fp32 temporaries stay small
            m = min(CHUNK_ROWS, n - lo)
            noise = torch.randn((m, DIM), device=dev, generator=g)
            if mode == "centroids":
                x = cent[torch.randint(0, 64, (m,), device=dev, generator=g)] + 0.3 * noise
            else:
                x = docs[(torch.arange(lo, lo + m, device=dev) // 16)] + 0.5 * noise
            idx.add_f16_device(torch.nn.functional.normalize(x, dim=1).half().contiguous(), m)
            del x, noise
        for qmode in ("random", "on-topic"):
            qs = []
            for _ in range(N_QUERY_BATCHES):
                if qmode == "random":
                    q = torch.randn((B, DIM), device=dev, generator=g)
                elif mode == "centroids":
                    q = cent[torch.randint(0, 64, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, DIM), device=dev, generator=g)
                else:
                    q = docs[torch.randint(0, n // 16, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, DIM), device=dev, generator=g)
                qs.append(q)
            loop = Loop(idx, qs)
            entry = {}
            for scan in ("fp16", "int8", "auto"):
                r = loop.timed(scan, 64, 48, False)
                chk = loop.exact_fp64_check(r, 2)
                entry[scan] = {"us_per_step": r["elapsed"] / 64 * 1e6, "queries_per_s": 64 * B / r["elapsed"], "repaired_in_timed_steps": r["fixed"],
                               "scanned_int8_image_in_every_timed_step": r["int8"], "scan8_level_after": r["scan8_level"],
                               "ids_match_exact_fp64_scan": chk["ids_match"], "exact_scans": r["timing"]["exact_scans"], "widened": r["timing"]["widened"]}
            res[f"{mode}/{qmode}"] = entry
            del loop
        idx.close()
        del idx
    res["operands"] = ("fp16 = the fp16 rows (option scan8 = 0); int8 = the int8 image FORCED (scan8 = 2: round 2's fixed start levels, the ladder moves only after "
                       "repairs); auto = the library's rule (scan8 = 1: the ladder's start is measured when the image is built -- 64 stored rows searched through "
                       "every rung -- so a shard the image does not pay on scans its fp16 rows from the first call)")
    res["note"] = ("scan8_level_after: per class of k (units: k <= 32, tens: larger k) 0 = one int8 image per query, 1 = two images, 2 = the class gave "
                   "the int8 image up and scans the fp16 rows; `int8` requests the image (option scan8 = 2), the library may decline per shard / class")
    return res


if __name__ == "__main__":
    main()
