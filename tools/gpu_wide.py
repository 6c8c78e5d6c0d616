"""Throughput of large query batches: passes of 128 / 256 queries (scan variants of csrc/rq_scan.hip) vs passes of 64 (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0"); N = int(os.environ.get("RQ_WIDE_ROWS", "1000000")); k = 10
idx = nat.NativeIndex(768, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, N, 125_000):
    n = min(125_000, N - lo)
    x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, n)
modes = [("64-pass", dict(wide_batch=0)), ("128-pass r1 (8 waves)", dict(wide_batch=2))]
modes += [(f"128-pass wide v{v}", dict(wide_batch=3, wide128=v)) for v in (0, 1, 8, 4, 5, 6, 7)]
modes += [(f"256-pass wide v{v}", dict(wide_batch=1, wide128=0, wide256=v)) for v in (11, 2)]
if os.environ.get("RQ_WIDE_ABLATE"):
    modes += [(f"128-pass ablation v{v}", dict(wide_batch=3, wide128=v)) for v in (90, 91, 92, 93, 94, 95)]
only = os.environ.get("RQ_WIDE_ONLY")
for B in (128, 256, 512):
    q = torch.randn((B, 768), device=dev, generator=g)
    o = (torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32))
    ref = None
    for name, opts in modes:
        if only and name != "64-pass" and only not in name: continue
        if B == 128 and name.startswith("256"): continue
        for n_, v_ in opts.items(): idx.set_option(n_, v_)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.15:          # a GPU that has idled runs its first launches 10-15 % slower
            for i in range(4): idx.search_device(q, B, k, 0, o[0], o[1], o[2], o[3], 0)
            torch.cuda.synchronize()
        idx.set_option("profile", 1); idx.reset_timing()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): idx.search_device(q, B, k, 0, o[0], o[1], o[2], o[3], 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        t = idx.timing(); idx.set_option("profile", 0)
        scan_us = t["scan_ms"] * 1e3 / max(t["scan_launches"], 1)
        rows = o[1].cpu().numpy().copy()
        if ref is None: ref = rows
        print(f"B={B:5d} {name:24s}: scan launch {scan_us:6.1f} us x {t['scan_launches'] // 20}  {dt*1e6:8.1f} us/call  {B/dt:10.0f} q/s  same_rows={bool(np.array_equal(rows, ref))} uncertified={int(o[3].sum())}", flush=True)
