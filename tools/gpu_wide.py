"""Throughput of large query batches (128 queries per corpus pass) vs 64 per pass (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0"); N = 1_000_000; k = 10
idx = nat.NativeIndex(768, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, N, 125_000):
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000)
for B in (64, 128, 256, 1024):
    q = torch.randn((B, 768), device=dev, generator=g)
    o = (torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32))
    ref = None
    for wide in (0, 1):
        idx.set_option("wide_batch", wide)
        for i in range(3): idx.search_device(q, B, k, 0, o[0], o[1], o[2], o[3], 0)
        idx.set_option("profile", 1); idx.reset_timing()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10): idx.search_device(q, B, k, 0, o[0], o[1], o[2], o[3], 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        t = idx.timing(); idx.set_option("profile", 0)
        scan_us = t["scan_ms"] * 1e3 / max(t["scan_launches"], 1)
        rows = o[1].cpu().numpy().copy()
        if ref is None: ref = rows
        print(f"B={B:5d} wide_batch={wide}: scan launch {scan_us:6.1f} us x {t['scan_launches'] // 10}  {dt*1e6:8.1f} us/call  {B/dt:10.0f} q/s  same_rows={bool(np.array_equal(rows, ref))} uncertified={int(o[3].sum())}", flush=True)
