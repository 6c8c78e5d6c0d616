"""Cost of the cross-shard merge kernel in the shape the 8-GPU bench uses: 16 batches x 64 queries per call,
world x 10 keys per query."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_uq_amd import _native as nat
nat.load_library()
dev = torch.device("cuda:0")
for world in (1, 2, 4, 8):
    for k in (10, 100):
        B = 16 * 64
        keys = torch.randint(1, 2**62, (B, world * k), device=dev, dtype=torch.int64)
        sc = torch.empty((B, k), device=dev); rw = torch.empty((B, k), device=dev, dtype=torch.int64)
        for _ in range(5): nat.merge_keys_device(keys, world * k, B, k, sc, rw, None, 0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): nat.merge_keys_device(keys, world * k, B, k, sc, rw, None, 0)
        e1.record(); torch.cuda.synchronize()
        print(f"world={world} k={k:3d}: merge of {B} queries x {world * k:4d} keys: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per call", flush=True)
