"""Calls of more than 64 queries at k = 10 on 1M x 768: int8 128-query passes against the fp16 wide passes (per call, device API)."""
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
gq = torch.Generator(device=dev); gq.manual_seed(4321)
for B in (128, 256, 512):
    q = torch.randn((B, 768), device=dev, generator=gq)
    sc = torch.empty((B, k), device=dev); rw = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.zeros((B,), device=dev, dtype=torch.int32)
    rows = {}
    for wide8 in (1, 0, 1, 0):
        idx.set_option("wide8", wide8); idx.set_option("profile", 0)
        for _ in range(5): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
        torch.cuda.synchronize(); idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 1)
        t0 = time.perf_counter()
        for _ in range(30): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        t = idx.timing(); rows[wide8] = rw.cpu().numpy().copy()
        print(f"B={B:4d} {'int8 128-query passes' if wide8 else 'fp16 wide passes     '}: {dt * 1e6:7.1f} us per call  {B / dt:9.0f} q/s  "
              f"scan launch {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us x {t['scan_launches'] // 30}  uncertified {int(st.sum())}", flush=True)
    print(f"B={B}: same rows: {np.array_equal(rows[0], rows[1])}")
idx.close()
