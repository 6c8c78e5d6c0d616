"""Encoder throughput on passage-like batches (ragged lengths up to 512 tokens): fused path vs the stock transformers module."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_uq_amd
from rag_uq_amd.embedders import NomicBertEmbedder
torch.manual_seed(0)
fused = NomicBertEmbedder(random_init=True, num_layers=12, device="cuda:0", dtype="float16", batch_size=256)
stock = NomicBertEmbedder(random_init=True, num_layers=12, device="cuda:0", dtype="float16", batch_size=256, fused=False)
stock.model.load_state_dict(fused.model.state_dict())
rng = np.random.default_rng(0)
for name, lens in (("512-token passages", np.full(1024, 512)), ("ragged 60..512", rng.integers(60, 513, size=1024)), ("ragged 20..200", rng.integers(20, 201, size=1024))):
    texts = ["".join(chr(97 + (i * 7 + j) % 26) for j in range(int(n))) for i, n in enumerate(lens)]
    res = {}
    for tag, e in (("fused", fused), ("stock", stock)):
        e.embed(texts[:256]); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = e.embed(texts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[tag] = (dt, out)
    a, b = res["fused"][1], res["stock"][1]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    print(f"{name:20s}: fused {1024 / res['fused'][0]:8.0f} texts/s ({res['fused'][0] * 1e3:7.1f} ms)   stock {1024 / res['stock'][0]:8.0f} texts/s ({res['stock'][0] * 1e3:7.1f} ms)   min cosine {cos.min():.6f}", flush=True)
