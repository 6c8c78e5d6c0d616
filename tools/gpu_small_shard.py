"""125 k-row shards (the 8-GPU per-rank shape), fused loop on two caller streams, k = 10: int8 image against fp16 rows on Gaussian,
document-structured and centroid corpora (on-topic and random queries).  Decides the size rule of the int8 scan."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
n, k, B = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000, 10, 64
g = torch.Generator(device=dev); g.manual_seed(7)
cent = torch.randn((64, 768), device=dev, generator=g)
docs = torch.randn((n // 16 + 1, 768), device=dev, generator=g)
for mode in ("gaussian", "documents", "centroids"):
    noise = torch.randn((n, 768), device=dev, generator=g)
    if mode == "centroids": x = cent[torch.randint(0, 64, (n,), device=dev, generator=g)] + 0.3 * noise
    elif mode == "documents": x = docs[(torch.arange(n, device=dev) // 16)] + 0.5 * noise
    else: x = noise
    idx = nat.NativeIndex(768, 0); idx.add_f16_device(torch.nn.functional.normalize(x, dim=1).half().contiguous(), n); del x, noise
    idx.set_option("pipeline", 2)
    for qmode in ("random", "on-topic"):
        if qmode == "on-topic" and mode == "gaussian": continue
        qs = []
        for i in range(16):
            if qmode == "random": q = torch.randn((B, 768), device=dev, generator=g)
            elif mode == "centroids": q = cent[torch.randint(0, 64, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, 768), device=dev, generator=g)
            else: q = docs[torch.randint(0, n // 16, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, 768), device=dev, generator=g)
            qs.append(q)
        outs = [(torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.zeros((B,), device=dev, dtype=torch.int32)) for _ in range(16)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        line = f"N={n} {mode:9s} {qmode:8s}:"
        for scan8 in (0, 2, 0, 2):
            idx.set_option("scan8", scan8)
            def loop(steps):
                for i in range(steps):
                    s = streams[i % 2]; j = i % 16
                    idx.search_hint_next_device(qs[(i + 2) % 16], B, s.cuda_stream)
                    idx.search_device(qs[j], B, k, 0, outs[j][0], outs[j][1], None, outs[j][2], s.cuda_stream)
                for s in streams: idx.search_flush_device(s.cuda_stream)
                torch.cuda.synchronize()
            loop(64)
            t0 = time.perf_counter(); loop(800); dt = (time.perf_counter() - t0) / 800
            unc = int(sum(int(o[2].sum()) for o in outs))
            line += f"  {'int8' if scan8 else 'fp16'} {dt * 1e6:6.1f} us (unc {unc})"
        print(line, flush=True)
    idx.close()
