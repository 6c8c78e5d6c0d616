"""Round 3: calls of more than 64 queries at k = 100 / 50 on 1M x 768: the fp16 wide passes (what the k > 32 class runs today: it starts on two
images per query, and the 256-query int8 pass wants one) against the one-image int8 passes forced with scan8_split = 0."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
n = 1_000_000
idx = nat.NativeIndex(768, 0); idx.reserve(n)
for c in range(8):
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000); del x
gq = torch.Generator(device=dev); gq.manual_seed(4321)
for k in (100, 50):
    for B in (256, 500):
        q = torch.randn((B, 768), device=dev, generator=gq)
        sc = torch.empty((B, k), device=dev); rw = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.zeros((B,), device=dev, dtype=torch.int32)
        rows = {}
        for name, opts in (("fp16 wide passes", {"scan8": 0}), ("int8 one image ", {"scan8": 2, "scan8_split": 0}), ("library default ", {"scan8": 1, "scan8_split": -1})):
            for o, v in opts.items(): idx.set_option(o, v)
            idx.set_option("profile", 0)
            for _ in range(5): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
            torch.cuda.synchronize(); idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 1)
            used = int(idx.get_option("scan8_used"))
            t0 = time.perf_counter()
            for _ in range(20): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            t = idx.timing(); rows[name] = rw.cpu().numpy().copy()
            print(f"k={k:3d} B={B:4d} {name}: {dt * 1e6:7.1f} us per call  {B / dt:9.0f} q/s  scan launch {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us x {t['scan_launches'] // 20}  "
                  f"int8 calls {int(idx.get_option('scan8_used')) - used}/20  uncertified {int(st.sum())}", flush=True)
        print(f"k={k} B={B}: same rows: {all(np.array_equal(r, rows['fp16 wide passes']) for r in rows.values())}", flush=True)
idx.close()
