"""Run a few searches with chosen wide-pass variants (for rocprofv3 --pmc / --kernel-trace runs).
usage: python3 tools/gpu_wide_one.py B "wide_batch=3,wide128=0;wide_batch=2" [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import rag_uq_amd
from rag_uq_amd import _native as nat
B = int(sys.argv[1]); specs = sys.argv[2].split(";"); N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dev = torch.device("cuda:0"); k = 10
idx = nat.NativeIndex(768, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, N, 125_000):
    n = min(125_000, N - lo)
    x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, n)
q = torch.randn((B, 768), device=dev, generator=g)
o = (torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32))
for spec in specs:
    for kv in spec.split(","):
        n_, v_ = kv.split("="); idx.set_option(n_, float(v_))
    for i in range(24): idx.search_device(q, B, k, 0, o[0], o[1], None, o[2], 0)
    torch.cuda.synchronize()
print("done", flush=True)
