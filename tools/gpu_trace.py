"""Small fixed workload for rocprofv3 --kernel-trace --stats: 1M x 768 corpus, B=64, k=10, 40 searches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
N = int(os.environ.get("RQ_N", 1_000_000)); B = int(os.environ.get("RQ_B", 64)); k = int(os.environ.get("RQ_K", 10))
iters = int(os.environ.get("RQ_ITERS", 40))
g = torch.Generator(device=dev); g.manual_seed(1235)
idx = nat.NativeIndex(768, 0); idx.reserve(N)
for lo in range(0, N, 250_000):
    n = min(250_000, N - lo)
    x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, n)
del x
qs = [torch.randn((B, 768), device=dev, generator=g) for _ in range(8)]
sc = torch.empty((B, k), device=dev); rows = torch.empty((B, k), device=dev, dtype=torch.int64)
keys = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.empty((B,), device=dev, dtype=torch.int32)
for opt in os.environ.get("RQ_OPTS", "").split(","):
    if "=" in opt:
        a, b = opt.split("="); idx.set_option(a, float(b))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    idx.search_device(qs[i % 8], B, k, 0, sc, rows, keys, st, 0)
torch.cuda.synchronize()
print("us/batch", (time.perf_counter() - t0) / iters * 1e6, "uncertified", int(st.sum().item()))
