"""A/B of an option of the fused scan + tail launch, interleaved in ONE process on one box: us per 64-query batch over 1M
rows, one stream, pipeline 2.   usage: python tools/gpu_ab_epi.py [option=epi]
  epi        selection form: 0 = compare / select, 1 = row position inside the score (v_med3 inserts)
  tail_first 1 = the tail workgroups take the first block ids of the launch"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rag_uq_amd
from rag_uq_amd import _native as nat
OPT = sys.argv[1] if len(sys.argv) > 1 else "epi"
dev = torch.device("cuda:0"); N = 1_000_000; B = 64; k = 10
idx = nat.NativeIndex(768, 0); idx.reserve(N)
for c in range(8):
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000)
g = torch.Generator(device=dev); g.manual_seed(4321)
qs = [torch.randn((B, 768), device=dev, generator=g) for _ in range(16)]
outs = [(torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32)) for _ in range(16)]
st = torch.cuda.Stream(device=dev)
idx.set_option("pipeline", 2)
ref = None
res = {0: [], 1: []}
for rnd in range(7):
    for epi in (0, 1):
        idx.set_option(OPT, epi)
        idx.set_option("profile", 1); idx.set_option("profile_stride", 4); idx.reset_timing()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(240):
            o = outs[i % 16]
            idx.search_device(qs[i % 16], B, k, 0, o[0], o[1], None, o[2], st.cuda_stream)
        idx.search_flush_device(st.cuda_stream)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 240
        t = idx.timing(); idx.set_option("profile", 0)
        rows = torch.stack([o[1] for o in outs]).cpu().numpy()
        bad = int(sum(int(o[2].sum()) for o in outs))
        if ref is None: ref = rows
        if rnd: res[epi].append((dt * 1e6, t["scan_ms"] * 1e3 / max(t["scan_launches"], 1)))
        print(f"round {rnd} {OPT}={epi}: {dt*1e6:7.1f} us/batch  scan launch {t['scan_ms']*1e3/max(t['scan_launches'],1):6.1f} us  same_rows={bool(np.array_equal(rows, ref))} uncertified={bad}", flush=True)
for epi in (0, 1):
    a = np.array(res[epi])
    print(f"{OPT}={epi}: mean {a[:,0].mean():.1f} us/batch (min {a[:,0].min():.1f}), scan launch mean {a[:,1].mean():.1f} us")
