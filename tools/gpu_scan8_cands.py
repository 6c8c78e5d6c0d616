"""Candidate rows per query behind the int8 scan (tail_stop 5 returns the count in the status slot). usage: [rows]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from rag_uq_amd import _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
idx = nat.NativeIndex(768, 0); idx.reserve(n)
for c in range((n + 124_999) // 125_000):
    m = min(125_000, n - c * 125_000)
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((m, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, m); del x
gq = torch.Generator(device=dev); gq.manual_seed(4321)
q = torch.randn((64, 768), device=dev, generator=gq)
for k in (10, 100):
    sc = torch.empty((64, k), device=dev); rw = torch.empty((64, k), device=dev, dtype=torch.int64); st = torch.zeros((64,), device=dev, dtype=torch.int32)
    for mode, mult, bb in ((0, 0, 1), (2, 1.25, 0), (2, 1.25, 1), (2, 1.1, 1)):
        idx.set_option("scan8", mode); idx.set_option("bin_bound", bb); idx.set_option("tail_local", 0)
        if mode: idx.set_option("thr_mult8", mult)
        idx.set_option("tail_stop", 5)
        idx.search_device(q, 64, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
        c = st.cpu().numpy()
        print(f"k={k} scan8={mode} mult={mult} bin_bound={bb}: candidate rows per query min {c.min()} median {int(np.median(c))} max {c.max()}", flush=True)
        idx.set_option("tail_stop", 0)
        idx.search_device(q, 64, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()   # (resets the counters a truncated tail leaves)
idx.close()
