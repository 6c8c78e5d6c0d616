"""Phase timing of the fused tail kernel (development probe): profile=2 times the tail launch, tail_stop cuts phases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
B = 64
for N in (1_000_000, 125_000):
    idx = nat.NativeIndex(768, 0); idx.reserve(N)
    g = torch.Generator(device=dev); g.manual_seed(1)
    for lo in range(0, N, 125_000):
        n = min(125_000, N - lo)
        x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()
        idx.add_f16_device(x, n)
    if os.environ.get("RQ_SCAN8"):      # phases of the tail behind the int8 scan (more candidates per query)
        idx.set_option("scan8", 1)
    qs = [torch.randn((B, 768), device=dev, generator=g) for _ in range(8)]
    for k in (10, 100):
        o = (torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32))
        for stop in (1, 2, 3, 4, 0):
            idx.set_option("tail_stop", stop); idx.set_option("profile", 2)
            for i in range(4): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
            torch.cuda.synchronize(); idx.reset_timing()
            t0 = time.perf_counter()
            for i in range(30): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            t = idx.timing()
            print(f"N={N} k={k} tail_stop={stop}: tail {t['scan_ms']*1e3/max(t['scan_launches'],1):7.1f} us   e2e {dt*1e6:7.1f} us", flush=True)
        idx.set_option("tail_stop", 0)
        for mode in (1,):
            idx.set_option("profile", 0)
            for i in range(6): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
            print(f"N={N} k={k} tail mode {mode}: e2e {dt*1e6:7.1f} us (uncertified {int(o[3].sum())})", flush=True)
        idx.set_option("profile", 1)
        for i in range(4): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
        torch.cuda.synchronize(); idx.reset_timing()
        for i in range(30): idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
        t = idx.timing(); print(f"N={N} k={k}: scan {t['scan_ms']*1e3/max(t['scan_launches'],1):7.1f} us", flush=True)
    idx.close()
