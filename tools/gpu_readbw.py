"""Plain streaming-read rate of this GPU next to the scan kernel's (same shard, same process, interleaved rounds).
    gpurun -- python tools/gpu_readbw.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_uq_amd import _native as nat

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
idx = nat.NativeIndex(768, 0)
idx.reserve(N)
for c in range(0, N, 125_000):
    n = min(125_000, N - c)
    x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev), dim=1).half().contiguous()
    idx.add_f16_device(x, n)
q = torch.randn((64, 768), device=dev)
sc = torch.empty((64, 10), device=dev); rw = torch.empty((64, 10), device=dev, dtype=torch.int64); st = torch.empty((64,), device=dev, dtype=torch.int32)
st_ = torch.cuda.Stream(device=dev)
def scan_us(iters=20):
    idx.set_option("profile", 1); idx.reset_timing()
    for _ in range(iters):
        idx.search_device(q, 64, 10, 0, sc, rw, None, st, st_.cuda_stream)
    torch.cuda.synchronize()
    t = idx.timing(); idx.set_option("profile", 0)
    return t["scan_ms"] * 1e3 / t["scan_launches"]
scan_us(5)
for rnd in range(3):
    for wg in (8,):
        for nt in (0, 1):
            print(f"round {rnd} read probe wg/cu={wg:2d} nt={nt}: {idx.read_bandwidth(20, nt, wg):7.1f} GB/s", flush=True)
    us = scan_us()
    print(f"round {rnd} scan kernel: {us:6.1f} us = {N * 1536 / us / 1e3:7.1f} GB/s", flush=True)
    for mode, what in ((1, "WITHOUT its bin-record stores"),):
        idx.set_option("scan_nostore", mode)
        us = scan_us()
        idx.set_option("scan_nostore", 0)
        print(f"round {rnd} scan kernel {what}: {us:6.1f} us = {N * 1536 / us / 1e3:7.1f} GB/s", flush=True)
