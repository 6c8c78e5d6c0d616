"""int8 scan: stand-alone scan launch, stand-alone tail launch (pipeline 0), by k and threshold multiplier.
usage: python tools/gpu_scan8_parts.py [rows]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from rag_uq_amd import _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
idx = nat.NativeIndex(768, 0)
idx.reserve(n)
for c in range((n + 124_999) // 125_000):
    m = min(125_000, n - c * 125_000)
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((m, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, m); del x
gq = torch.Generator(device=dev); gq.manual_seed(4321)
qs = [torch.randn((64, 768), device=dev, generator=gq) for _ in range(16)]
st = torch.cuda.Stream(device=dev)
idx.set_option("pipeline", 0)
for k in (10, 100):
    outs = [(torch.empty((64, k), device=dev), torch.empty((64, k), device=dev, dtype=torch.int64), torch.zeros((64,), device=dev, dtype=torch.int32)) for _ in range(16)]
    for mode, mult in ((0, 0), (1, 2.25), (1, 1.5), (1, 1.25), (1, 1.1)):
        idx.set_option("scan8", mode)
        if mode:
            idx.set_option("thr_mult8", mult)
        line = f"k={k:4d} scan8={mode} mult={mult:4.2f}:"
        for prof, name in ((1, "scan"), (2, "tail")):
            idx.set_option("profile", 0)
            def loop(steps):
                with torch.cuda.stream(st):
                    for i in range(steps):
                        j = i % 16
                        idx.search_device(qs[j], 64, k, 0, outs[j][0], outs[j][1], None, outs[j][2], st.cuda_stream)
                torch.cuda.synchronize()
            loop(32)
            idx.reset_timing(); idx.set_option("profile", prof); idx.set_option("profile_stride", 1)
            loop(100)
            t = idx.timing()
            line += f"  {name} {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us"
        unc = int(sum(int(o[2].sum()) for o in outs))
        print(line + f"  uncertified {unc}", flush=True)
idx.close()
