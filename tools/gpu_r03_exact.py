"""Round 3: the exact scan of a whole shard (the repair ladder's last rung) -- fp64 matrix cores (csrc/rq_exact.hip) against rounds 1-2's
per-(bin, query) kernel.  1M x 768, 64 queries, k = 10, forced with slack_bins; ms per batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
n, k, B = 1_000_000, 10, 64
idx = nat.NativeIndex(768, 0); idx.reserve(n)
for c in range(8):
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000); del x
q = torch.randn((B, 768), device=dev)
sc = torch.empty((B, k), device=dev); rw = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.zeros((B,), device=dev, dtype=torch.int32)
idx.set_option("scan8", 0)
idx.search_device(q, B, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
ref = rw.cpu().numpy().copy()
idx.set_option("slack_bins", n)
for name, v in (("fp64 matrix cores (rq_exact.hip)", 1), ("per (bin, query) workgroups (rq_rescore_kernel)", 0), ("fp64 matrix cores (rq_exact.hip)", 1)):
    idx.set_option("exact_mfma", v)
    for _ in range(2): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): idx.search_device(q, B, k, 0, sc, rw, None, st, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"exact scan, {name}: {dt * 1e3:7.2f} ms per 64-query batch at 1M rows; same rows as the certified path: {np.array_equal(rw.cpu().numpy(), ref)}", flush=True)
idx.close()
