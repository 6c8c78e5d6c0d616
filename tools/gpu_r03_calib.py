"""Round 3: what the image-build calibration of the int8 ladder ("scan8" = 1) measures and chooses, per corpus kind and shard size, and the
fused two-stream loop's time per batch with the chosen operand against the forced ones.  usage: [rows ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
k, B = 10, 64
for n in [int(a) for a in sys.argv[1:]] or [125_000, 1_000_000]:
    g = torch.Generator(device=dev); g.manual_seed(7)
    cent = torch.randn((64, 768), device=dev, generator=g)
    docs = torch.randn((n // 16 + 1, 768), device=dev, generator=g)
    for mode in ("gaussian", "documents", "centroids"):
        idx = nat.NativeIndex(768, 0); idx.reserve(n)
        for lo in range(0, n, 125_000):
            m = min(125_000, n - lo)
            noise = torch.randn((m, 768), device=dev, generator=g)
            if mode == "centroids": x = cent[torch.randint(0, 64, (m,), device=dev, generator=g)] + 0.3 * noise
            elif mode == "documents": x = docs[(torch.arange(lo, lo + m, device=dev) // 16)] + 0.5 * noise
            else: x = noise
            idx.add_f16_device(torch.nn.functional.normalize(x, dim=1).half().contiguous(), m); del x, noise
        idx.set_option("pipeline", 2)
        qs = []
        for i in range(16):
            if mode == "gaussian": q = torch.randn((B, 768), device=dev, generator=g)
            elif mode == "centroids": q = cent[torch.randint(0, 64, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, 768), device=dev, generator=g)
            else: q = docs[torch.randint(0, n // 16, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, 768), device=dev, generator=g)
            qs.append(q)
        outs = [(torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.zeros((B,), device=dev, dtype=torch.int32)) for _ in range(16)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        def loop(steps):
            for i in range(steps):
                s = streams[i % 2]; j = i % 16
                idx.search_hint_next_device(qs[(i + 2) % 16], B, s.cuda_stream)
                idx.search_device(qs[j], B, k, 0, outs[j][0], outs[j][1], None, outs[j][2], s.cuda_stream)
            for s in streams: idx.search_flush_device(s.cuda_stream)
            torch.cuda.synchronize()
        line = f"N={n} {mode:9s} on-topic:"
        for name, scan8 in (("fp16", 0), ("int8 forced", 2), ("auto", 1)):
            idx.set_option("scan8", scan8)
            t0 = time.perf_counter(); loop(16); first = (time.perf_counter() - t0) * 1e3
            unc_first = int(sum(int(o[2].sum()) for o in outs))
            loop(64)
            t0 = time.perf_counter(); loop(800); dt = (time.perf_counter() - t0) / 800
            unc = int(sum(int(o[2].sum()) for o in outs))
            line += f"  {name} {dt * 1e6:6.1f} us (unc first 16: {unc_first}, steady {unc}, level {int(idx.get_option('scan8_level')) % 10})"
            if scan8 == 1:
                cal = " | calib k<=32 ms(unc): " + " ".join(f"{idx.get_option(f'scan8_calib_ms_0{l}'):.3f}({int(idx.get_option(f'scan8_calib_unc_0{l}'))})" for l in range(3))
                cal += " k>32: " + " ".join(f"{idx.get_option(f'scan8_calib_ms_1{l}'):.3f}({int(idx.get_option(f'scan8_calib_unc_1{l}'))})" for l in range(3))
                cal += f" -> level {idx.get_option('scan8_level'):.0f}; first 16 steps incl. image build + calibration {first:.1f} ms"
                line += cal
        print(line, flush=True)
        idx.close()
