"""Behaviour on clustered corpora (near-duplicate neighbourhoods, where many rows of a bin reach the threshold):
certificate pass rate, widen / exact-scan counts and time per batch.  No oracle: ids are cross-checked against the
library's own exact fp64 scan (fast_tail=0 + slack -> every path is exact by construction, see tests for parity).
    gpurun -- python tools/gpu_clustered.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_uq_amd import _native as nat

dev = torch.device("cuda:0")
def build(n, mode, seed=7):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    idx = nat.NativeIndex(768, 0); idx.reserve(n)
    cent = torch.randn((64, 768), device=dev, generator=g)
    docs = torch.randn((n // 16 + 1, 768), device=dev, generator=g)      # "documents" of 16 consecutive passages
    for c in range(0, n, 125_000):
        m = min(125_000, n - c)
        noise = torch.randn((m, 768), device=dev, generator=g)
        if mode == "centroids":       # 64 centroids + 0.3 noise, assigned at random
            x = cent[torch.randint(0, 64, (m,), device=dev, generator=g)] + 0.3 * noise
        elif mode == "documents":     # consecutive rows share a document vector: neighbours in a bin are similar
            x = docs[(torch.arange(c, c + m, device=dev) // 16)] + 0.5 * noise
        else:
            x = noise
        idx.add_f16_device(torch.nn.functional.normalize(x, dim=1).half().contiguous(), m)
    return idx, cent, docs

for n in (1_000_000, 125_000):
    for mode in ("gaussian", "centroids", "documents"):
        idx, cent, docs = build(n, mode)
        for opt in filter(None, os.environ.get("RQ_OPTS", "").split(",")):     # e.g. RQ_OPTS=scan8=2,scan8_split=1
            name, val = opt.split("="); idx.set_option(name, float(val))
        g = torch.Generator(device=dev); g.manual_seed(99)
        for qmode in ("random", "on-topic"):
            if qmode == "random":
                q = torch.randn((64, 768), device=dev, generator=g)
            elif mode == "centroids":
                q = cent[torch.randint(0, 64, (64,), device=dev, generator=g)] + 0.3 * torch.randn((64, 768), device=dev, generator=g)
            elif mode == "documents":
                q = docs[torch.randint(0, n // 16, (64,), device=dev, generator=g)] + 0.3 * torch.randn((64, 768), device=dev, generator=g)
            else:
                continue
            for k in (10, 100):
                sc = torch.empty((64, k), device=dev); rw = torch.empty((64, k), device=dev, dtype=torch.int64); st = torch.empty((64,), device=dev, dtype=torch.int32)
                idx.reset_timing()
                idx.search_device(q, 64, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
                unc = int(st.sum())
                t0 = time.perf_counter()
                for _ in range(10):
                    idx.search_device(q, 64, k, 0, sc, rw, None, st, 0)
                    idx.search_fixup_device(q, 64, k, 0, sc, rw, None, st, 0)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 10
                t = idx.timing()
                print(f"N={n:8d} corpus={mode:9s} queries={qmode:8s} k={k:3d}: uncertified first pass {unc:2d}/64, widened {t['widened']:3d} exact scans {t['exact_scans']:3d} (over 11 calls), {dt*1e6:8.1f} us per batch incl. fix-up", flush=True)
        idx.close()
