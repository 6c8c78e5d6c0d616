"""BASELINE.json configs[3] shape with random-init weights: 256 raw text queries -> NomicBert forward (PyTorch-ROCm)
-> HIP search over a 1M x 768 corpus.  Times only; retrieval quality needs real weights (none offline)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rag_uq_amd
from rag_uq_amd import _native as nat
from rag_uq_amd.embedders import NomicBertEmbedder
dev = torch.device("cuda:0")
emb = NomicBertEmbedder(random_init=True, device="cuda:0", dtype="float16", batch_size=256)
N = 1_000_000
idx = nat.NativeIndex(768, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, N, 125_000):
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000)
texts = [f"what is the answer to question number {i} about topic {i % 17} and entity {i * 31 % 101}?" for i in range(256)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = emb.embed(texts)
    t1 = time.perf_counter()
    s, r = idx.search(v, 10)
    t2 = time.perf_counter()
    print(f"rep {rep}: encode 256 texts {1e3*(t1-t0):7.2f} ms, search top-10 over {N} rows {1e3*(t2-t1):7.2f} ms, total {1e3*(t2-t0):7.2f} ms -> {256/(t2-t0):8.0f} text queries/s", flush=True)
