"""Development probe: centroid corpus, on-topic queries, k = 10: candidate rows per query (tail_stop 5) and certificate
outcome for one- and two-image queries."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
n = 1_000_000
g = torch.Generator(device=dev); g.manual_seed(7)
idx = nat.NativeIndex(768, 0); idx.reserve(n)
cent = torch.randn((64, 768), device=dev, generator=g)
for c in range(0, n, 125_000):
    noise = torch.randn((125_000, 768), device=dev, generator=g)
    x = cent[torch.randint(0, 64, (125_000,), device=dev, generator=g)] + 0.3 * noise
    idx.add_f16_device(torch.nn.functional.normalize(x, dim=1).half().contiguous(), 125_000)
q = cent[torch.randint(0, 64, (64,), device=dev, generator=g)] + 0.3 * torch.randn((64, 768), device=dev, generator=g)
k = 10
sc = torch.empty((64, k), device=dev); rw = torch.empty((64, k), device=dev, dtype=torch.int64); st = torch.zeros((64,), device=dev, dtype=torch.int32)
ref = None
for scan8, split, mult in ((0, 0, 1.25), (2, 0, 1.25), (2, 1, 1.25), (2, 1, 1.6), (2, 1, 2.25)):
    idx.set_option("scan8", scan8); idx.set_option("scan8_split", split); idx.set_option("thr_mult8", mult)
    idx.set_option("tail_stop", 5)
    idx.search_device(q, 64, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
    c = st.cpu().numpy().copy()
    idx.set_option("tail_stop", 0)
    idx.search_device(q, 64, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()      # (a truncated tail leaves counters behind)
    idx.search_device(q, 64, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
    unc = int(st.sum())
    rows = rw.cpu().numpy().copy(); scores = sc.cpu().numpy().copy()
    if ref is None: ref = (rows, scores)
    print(f"scan8={scan8} split={split} mult={mult}: published keys per query min {c.min()} median {int(np.median(c))} max {c.max()}  uncertified {unc}/64  "
          f"rows == fp16 path: {np.array_equal(rows, ref[0])}  top score {scores[:, 0].mean():.4f} 10th {scores[:, 9].mean():.4f}  eps_rows {idx.get_option('scan8_row_err'):.5f}", flush=True)
idx.close()
