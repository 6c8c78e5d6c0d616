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
    for B in (1, 64):
        q = rng.standard_normal((B, 768)).astype(np.float32)
        for _ in range(20): idx.search(q, 10)
        t0 = time.perf_counter(); it = 300 if N < 1_000_000 else 100
        for _ in range(it): idx.search(q, 10)
        dt = (time.perf_counter() - t0) / it
        print(f"N={N:8d} B={B:2d}: {dt*1e6:8.1f} us per call  ({B/dt:9.0f} queries/s)", flush=True)
    idx.close()
