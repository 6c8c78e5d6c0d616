"""e2e us/batch for pipeline x profile x streams at 1M rows (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
N = int(os.environ.get("RQ_N", 1_000_000)); B = 64; k = 10
idx = nat.NativeIndex(768, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, N, 125_000):
    n = min(125_000, N - lo)
    x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, n)
qs = [torch.randn((B, 768), device=dev, generator=g) for _ in range(8)]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
outs = [(torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32)) for _ in range(3)]
def run(ns, iters=int(os.environ.get("RQ_ITERS", 60))):
    def go(i):
        j = i % ns; o = outs[j]
        idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], streams[j].cuda_stream)
    for i in range(8): go(i)
    for s in streams: idx.search_flush_device(s.cuda_stream)
    torch.cuda.synchronize(); idx.reset_timing()
    t0 = time.perf_counter()
    for i in range(iters): go(i)
    th = time.perf_counter() - t0
    for s in streams: idx.search_flush_device(s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    t = idx.timing()
    return dt * 1e6, th / iters * 1e6, (t["scan_ms"] * 1e3 / t["scan_launches"]) if t["scan_launches"] else 0.0
for opts in os.environ.get("RQ_OPTSETS", "ring=4,prefetch=4,wg_per_cu=2").split(";"):
    for o in opts.split(","):
        a, b = o.split("="); idx.set_option(a, float(b))
    for pipe in [int(v) for v in os.environ.get("RQ_PIPES", "0,2").split(",")]:
        for prof in (0,):
            for ns in [int(v) for v in os.environ.get("RQ_STREAMS", "1,2").split(",")]:
                idx.set_option("pipeline", pipe); idx.set_option("profile", prof)
                e2e, host, scan = run(ns)
                print(f"{opts:32s} pipeline={pipe} profile={prof} streams={ns}: e2e {e2e:7.1f} us/batch  host-enqueue {host:6.1f} us/call  scan(ev) {scan:6.1f} us", flush=True)
