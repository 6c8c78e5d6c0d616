"""int8 scan vs fp16 scan on the headline shape (1M x 768, 64 queries, k = 10 / 100): time per batch in a fused loop,
scan launch duration, uncertified queries, agreement of the two paths.  usage: python tools/gpu_scan8.py [rows]"""
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
idx.set_option("pipeline", 2)
res = {}
for k in (10, 100):
    outs = [(torch.empty((64, k), device=dev), torch.empty((64, k), device=dev, dtype=torch.int64), torch.zeros((64,), device=dev, dtype=torch.int32)) for _ in range(16)]
    for mode in (0, 2, 3, 0, 2, 3):            # 0 = fp16 scan, 2 = int8 scan, 3 = int8 scan with split queries (two int8 images per query)
        idx.set_option("scan8", 2 if mode else 0)
        idx.set_option("scan8_split", 1 if mode == 3 else 0)
        idx.set_option("profile", 0)
        def loop(steps):
            with torch.cuda.stream(st):
                for i in range(steps):
                    j = i % 16
                    idx.search_hint_next_device(qs[(i + 1) % 16], 64, st.cuda_stream)
                    idx.search_device(qs[j], 64, k, 0, outs[j][0], outs[j][1], None, outs[j][2], st.cuda_stream)
                idx.search_flush_device(st.cuda_stream)
            torch.cuda.synchronize()
        loop(64)
        idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 4)
        t0 = time.perf_counter(); loop(400); dt = (time.perf_counter() - t0) / 400
        t = idx.timing()
        unc = int(sum(int(o[2].sum()) for o in outs))
        rows = torch.stack([o[1] for o in outs]).cpu().numpy()
        res[(k, mode)] = rows
        print(f"k={k:4d} scan8={mode}: {dt * 1e6:7.1f} us/batch  {64 / dt:9.0f} q/s  scan launch {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us "
              f"({t['scan_bytes'] / max(t['scan_launches'], 1) / 1e6:.0f} MB)  uncertified {unc}  row_err {idx.get_option('scan8_row_err'):.5f}", flush=True)
    print(f"k={k}: int8 rows == fp16 rows: {np.array_equal(res[(k, 0)], res[(k, 2)])}, split-query int8 rows == fp16 rows: {np.array_equal(res[(k, 0)], res[(k, 3)])}")
idx.close()
