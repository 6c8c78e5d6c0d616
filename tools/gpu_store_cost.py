"""Cost of the scan's bin-record stores for several scan variants (HIP events, interleaved, one process).
    gpurun -- python tools/gpu_store_cost.py"""
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
s_ = torch.cuda.Stream(device=dev)
def scan_us(iters=20):
    idx.set_option("profile", 1); idx.reset_timing()
    for _ in range(iters):
        idx.search_device(q, 64, 10, 0, sc, rw, None, st, s_.cuda_stream)
    torch.cuda.synchronize()
    t = idx.timing(); idx.set_option("profile", 0)
    return t["scan_ms"] * 1e3 / t["scan_launches"]
variants = [(1, 2, 1, 2), (1, 3, 4, 2), (2, 4, 1, 2), (2, 6, 4, 2), (2, 3, 1, 3), (1, 2, 1, 3)]
for rnd in range(2):
    for ks, ring, pf, wg in variants:
        for name, v in (("kstage", ks), ("ring", ring), ("prefetch", pf), ("wg_per_cu", wg)):
            idx.set_option(name, v)
        scan_us(4)
        a = scan_us()
        idx.set_option("scan_nostore", 1); b = scan_us(); idx.set_option("scan_nostore", 0)
        print(f"round {rnd} kstage={ks} ring={ring} prefetch={pf} wg/cu={wg}: with stores {a:6.1f} us, without {b:6.1f} us, cost {a - b:5.1f} us", flush=True)
