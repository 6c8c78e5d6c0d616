"""Round 3: DenseIndex.search_device_vectors at configs[3]'s shape (256 queries, k = 10, 1M ids) -- the search itself against the tuple assembly,
with the collector paused during assembly (the shipped form) and not (gc.disable patched out)."""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_uq_amd  # noqa: F401
from rag_uq_amd import streaming_index as si

n = 1_000_000
idx = si.DenseIndex(persist_directory="", embedder=None, load_persisted=False, auto_persist=False)
rng = np.random.default_rng(1235)
for lo in range(0, n, 125_000):
    idx.add_vectors([f"d{i}" for i in range(lo, lo + 125_000)], rng.standard_normal((125_000, 768), dtype=np.float32), texts=[""] * 125_000)
dq = torch.randn((256, 768), device="cuda:0")
idx.search_device_vectors(dq, 10); torch.cuda.synchronize()
real_disable = gc.disable
for rep in range(3):
    for paused in (True, False):
        si.gc.disable = real_disable if paused else (lambda: None)
        t_rows = t_asm = t_all = 0.0
        for _ in range(100):
            dq = torch.randn((256, 768), device="cuda:0"); torch.cuda.synchronize()
            t0 = time.perf_counter(); sr = idx._search_device_rows(dq, 10); t1 = time.perf_counter(); idx._assemble(*sr); t2 = time.perf_counter()
            t_rows += t1 - t0; t_asm += t2 - t1
            dq = torch.randn((256, 768), device="cuda:0"); torch.cuda.synchronize()
            t0 = time.perf_counter(); idx.search_device_vectors(dq, 10); t_all += time.perf_counter() - t0
        print(f"collector paused={paused}: search+D2H {t_rows * 1e4:.1f} us, assembly {t_asm * 1e4:.1f} us, search_device_vectors {t_all * 1e4:.1f} us", flush=True)
si.gc.disable = real_disable
