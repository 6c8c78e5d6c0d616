"""Scan-kernel variants, interleaved rounds in ONE process (development probe): HIP-event time per launch + parity."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
N = int(os.environ.get("RQ_N", 1_000_000)); B = 64; k = 10
idx = nat.NativeIndex(768, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, N, 125_000):
    n = min(125_000, N - lo)
    x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, n)
qs = [torch.randn((B, 768), device=dev, generator=g) for _ in range(8)]
o = (torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32))
variants = [dict(kstage=2, ring=4, prefetch=4, wg_per_cu=2), dict(kstage=2, ring=4, prefetch=1, wg_per_cu=3),
            dict(kstage=1, ring=3, prefetch=4, wg_per_cu=2), dict(kstage=1, ring=3, prefetch=12, wg_per_cu=2),
            dict(kstage=1, ring=2, prefetch=4, wg_per_cu=3), dict(kstage=1, ring=2, prefetch=1, wg_per_cu=3),
            dict(kstage=1, ring=2, prefetch=4, wg_per_cu=2), dict(kstage=1, ring=4, prefetch=4, wg_per_cu=1)]
ref = None
times = {i: [] for i in range(len(variants))}
for rnd in range(4):
    for vi, v in enumerate(variants):
        for name, val in v.items(): idx.set_option(name, val)
        idx.set_option("profile", 1)
        for i in range(3): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
        torch.cuda.synchronize(); idx.reset_timing()
        for i in range(12): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
        t = idx.timing(); times[vi].append(t["scan_ms"] * 1e3 / t["scan_launches"])
        idx.search_device(qs[0], B, k, 0, o[0], o[1], o[2], o[3], 0); torch.cuda.synchronize()
        rows = o[1].cpu().numpy().copy()
        if ref is None: ref = rows
        assert np.array_equal(rows, ref), f"variant {v} changes the result"
        assert int(o[3].sum()) == 0
for vi, v in enumerate(variants):
    ts = times[vi]
    print(f"{str(v):70s} scan us: min {min(ts):6.1f} median {sorted(ts)[len(ts)//2]:6.1f}  ({N*1536/min(ts)/1e3:6.1f} GB/s best)", flush=True)
