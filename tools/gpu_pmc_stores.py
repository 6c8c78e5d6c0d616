"""Workload for `rocprofv3 --pmc ...`: 6 scan launches with the bin-record stores, then 6 without (1M x 768)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_uq_amd import _native as nat
N = 1_000_000
dev = torch.device("cuda:0")
idx = nat.NativeIndex(768, 0)
idx.reserve(N)
for c in range(0, N, 125_000):
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000)
q = torch.randn((64, 768), device=dev)
sc = torch.empty((64, 10), device=dev); rw = torch.empty((64, 10), device=dev, dtype=torch.int64); st = torch.empty((64,), device=dev, dtype=torch.int32)
for mode in (0, 1):
    idx.set_option("scan_nostore", mode)
    for _ in range(6):
        idx.search_device(q, 64, 10, 0, sc, rw, None, st, 0)
    torch.cuda.synchronize()
