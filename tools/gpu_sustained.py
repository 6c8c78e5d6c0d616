"""Scan launch time over 8 s of back-to-back batches (fused single-stream arrangement): does the box hold its rate?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_uq_amd import _native as nat
N = 1_000_000; dev = torch.device("cuda:0")
idx = nat.NativeIndex(768, 0); idx.reserve(N)
for c in range(0, N, 125_000):
    idx.add_f16_device(torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev), dim=1).half().contiguous(), 125_000)
q = torch.randn((64, 768), device=dev)
sc = torch.empty((64, 10), device=dev); rw = torch.empty((64, 10), device=dev, dtype=torch.int64); st = torch.empty((64,), device=dev, dtype=torch.int32)
s = torch.cuda.Stream(device=dev)
idx.set_option("pipeline", 2)
t_start = time.perf_counter()
while time.perf_counter() - t_start < 8.0:
    t0 = time.perf_counter()
    for _ in range(400):
        idx.search_device(q, 64, 10, 0, sc, rw, None, st, s.cuda_stream)
    idx.search_flush_device(s.cuda_stream); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400
    print(f"t = {time.perf_counter() - t_start:5.2f} s: {dt * 1e6:6.1f} us per batch  ({64 / dt:8.0f} queries/s)", flush=True)
