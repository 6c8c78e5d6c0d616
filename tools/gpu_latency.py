"""Latency of the blocking host-buffer API (rq_search through NativeIndex.search): one query per call, the reference's
call pattern (streaming_index.py:338-370), and a batch of 64, on 10k / 100k / 1M rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
for N in (10_000, 100_000, 1_000_000):
    idx = nat.NativeIndex(768, 0); idx.reserve(N)
    for c in range(0, N, 125_000):
        n = min(125_000, N - c)
        idx.add_f16_device(torch.nn.functional.normalize(torch.randn((n, 768), device=dev), dim=1).half().contiguous(), n)
    rng = np.random.default_rng(3)
    for B, k in ((1, 10), (1, 50), (64, 10), (64, 100)):      # (1, 50): the reference's evaluation loop, one question, pool of 50
        q = rng.standard_normal((B, 768)).astype(np.float32)
        line = f"N={N:8d} B={B:2d} k={k:3d}:"
        for scan8 in (1, 0):                                  # library default (int8 image from 200 k rows) / fp16 scan
            idx.set_option("scan8", scan8)
            for _ in range(20): idx.search(q, k)
            t0 = time.perf_counter(); it = 300 if N < 1_000_000 else 100
            for _ in range(it): idx.search(q, k)
            dt = (time.perf_counter() - t0) / it
            line += f"  {'default' if scan8 else 'fp16 scan'} {dt*1e6:7.1f} us per call ({B/dt:8.0f} queries/s)"
        print(line, flush=True)
    idx.close()
