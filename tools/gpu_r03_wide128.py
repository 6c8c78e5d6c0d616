"""Round 3 A/B: the 128-query int8 pass -- rq_scan.hip I8 = 3 (4 waves x 2 groups, two workgroups per CU) against the rq_scan_wide.hip form
(8 waves x 1 group, one workgroup per CU, option wide128_8 = 26); B = 128 and B = 100, k = 10, 1M rows, per call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
n, k = 1_000_000, 10
idx = nat.NativeIndex(768, 0); idx.reserve(n)
for c in range(8):
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000); del x
idx.set_option("scan8", 2)
for B in (128, 100, 384):
    q = torch.randn((B, 768), device=dev)
    sc = torch.empty((B, k), device=dev); rw = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.zeros((B,), device=dev, dtype=torch.int32)
    rows = {}
    for v in (0, 26, 0, 26):
        idx.set_option("wide128_8", v); idx.set_option("profile", 0)
        for _ in range(5): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
        torch.cuda.synchronize(); idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 1)
        t0 = time.perf_counter()
        for _ in range(30): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        t = idx.timing(); rows[v] = rw.cpu().numpy().copy()
        print(f"B={B:4d} wide128_8={v:2d}: {dt * 1e6:7.1f} us per call  {B / dt:9.0f} q/s  scan launches {t['scan_launches'] // 30} avg {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us  uncertified {int(st.sum())}", flush=True)
    print(f"B={B}: same rows: {np.array_equal(rows[0], rows[26])}", flush=True)
# remainder passes of fp16 calls (128 + 64, 256 + 64): the 64-query pass now runs on its own two-per-CU grid
idx.set_option("scan8", 0); idx.set_option("wide128_8", 0)
for B in (192, 320, 448):
    q = torch.randn((B, 768), device=dev)
    sc = torch.empty((B, k), device=dev); rw = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.zeros((B,), device=dev, dtype=torch.int32)
    idx.set_option("profile", 0)
    for _ in range(5): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
    torch.cuda.synchronize(); idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 1)
    t0 = time.perf_counter()
    for _ in range(30): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    t = idx.timing()
    print(f"fp16 B={B:4d}: {dt * 1e6:7.1f} us per call  {B / dt:9.0f} q/s  scan launches {t['scan_launches'] // 30} avg {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us  uncertified {int(st.sum())}", flush=True)
idx.close()
