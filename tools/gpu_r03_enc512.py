"""Round 3: one batch of 256 x 512-token texts through the fused encoder forward (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_uq_amd  # noqa: F401
from rag_uq_amd.embedders import NomicBertEmbedder
torch.manual_seed(0)
e = NomicBertEmbedder(random_init=True, num_layers=12, device="cuda:0", dtype="float16", batch_size=256)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
texts = ["".join(chr(97 + (i * 7 + j) % 26) for j in range(L)) for i in range(256)]
e.embed_device(texts); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): e.embed_device(texts)
torch.cuda.synchronize()
print(f"L={L}: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms per 256 texts", flush=True)
