"""Tail phases on a document-structured corpus (16 consecutive similar passages per document), where many rows of a bin
reach the threshold: profile = 2 times the stand-alone tail, tail_stop cuts it after phase A..D.  (development probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0"); n = 1_000_000; B = 64
g = torch.Generator(device=dev); g.manual_seed(7)
idx = nat.NativeIndex(768, 0); idx.reserve(n)
docs = torch.randn((n // 16 + 1, 768), device=dev, generator=g)
for c in range(0, n, 125_000):
    noise = torch.randn((125_000, 768), device=dev, generator=g)
    x = docs[(torch.arange(c, c + 125_000, device=dev) // 16)] + 0.5 * noise
    idx.add_f16_device(torch.nn.functional.normalize(x, dim=1).half().contiguous(), 125_000)
g = torch.Generator(device=dev); g.manual_seed(99)
q = torch.randn((B, 768), device=dev, generator=g)
for k in (10, 100):
    o = (torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32))
    for stop in (1, 2, 3, 4, 0):
        idx.set_option("tail_stop", stop); idx.set_option("profile", 2)
        for i in range(3): idx.search_device(q, B, k, 0, o[0], o[1], None, o[2], 0)
        torch.cuda.synchronize(); idx.reset_timing()
        for i in range(10): idx.search_device(q, B, k, 0, o[0], o[1], None, o[2], 0)
        torch.cuda.synchronize()
        t = idx.timing()
        print(f"documents k={k} tail_stop={stop}: tail {t['scan_ms']*1e3/max(t['scan_launches'],1):7.1f} us  uncertified {int(o[2].sum()) if stop == 0 else '-'}", flush=True)
    idx.set_option("tail_stop", 0); idx.set_option("profile", 0)
