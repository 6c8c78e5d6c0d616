"""Round 3: calls of more than 64 queries at k = 10 on 1M x 768 -- the int8 256-query pass (csrc/rq_scan_wide.hip I8, option wide256_8 = 22 / 25) against round 2's 128-query int8 passes (wide256_8 = 0) and the fp16 wide passes (wide8 = 0); per call, device API."""
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
gq = torch.Generator(device=dev); gq.manual_seed(4321)
for B in (256, 512, 500, 192):
    q = torch.randn((B, 768), device=dev, generator=gq)
    sc = torch.empty((B, k), device=dev); rw = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.zeros((B,), device=dev, dtype=torch.int32)
    rows = {}
    for name, w8, v in (("int8 256-pass v22", 1, 22), ("int8 256-pass v25", 1, 25), ("int8 256 32x32 v30", 1, 30), ("int8 256 32x32 v31", 1, 31), ("int8 256 32x32 v33", 1, 33), ("int8 256 32x32 v32", 1, 32), ("int8 128-passes   ", 1, 0), ("fp16 wide passes  ", 0, 22),
                        ("int8 256 32x32 v31", 1, 31), ("int8 256 32x32 v33", 1, 33), ("int8 128-passes   ", 1, 0)):
        idx.set_option("wide8", w8); idx.set_option("wide256_8", v); idx.set_option("profile", 0)
        for _ in range(5): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
        torch.cuda.synchronize(); idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 1)
        t0 = time.perf_counter()
        for _ in range(30): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        t = idx.timing(); rows[name] = rw.cpu().numpy().copy()
        print(f"B={B:4d} {name}: {dt * 1e6:7.1f} us per call  {B / dt:9.0f} q/s  scan launch {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us x {t['scan_launches'] // 30}  "
              f"uncertified {int(st.sum())}", flush=True)
    ref = rows["fp16 wide passes  "]
    print(f"B={B}: same rows as the fp16 passes: {all(np.array_equal(r, ref) for r in rows.values())}", flush=True)
idx.close()
