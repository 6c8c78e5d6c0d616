import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
from oracle import dense_oracle as orc
n = 60_000
x16 = orc.synthetic_corpus(n, 768, seed=81)
idx = nat.NativeIndex(768, 0); idx.add_f16(x16); idx.set_option("scan8", 2)
B, k = 256, 10
nb = (n + 63) // 64
q = orc.synthetic_queries(B, 768, seed=5)
dq = torch.from_numpy(q).cuda(); sc = torch.empty((B, k), device="cuda"); rw = torch.empty((B, k), device="cuda", dtype=torch.int64); st = torch.zeros((B,), device="cuda", dtype=torch.int32)
def pooled(variant):
    idx.set_option("wide256_8", variant)
    idx.search_device(dq, B, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
    return np.stack([idx.debug_pooled(j, nb) for j in range(B)])
ref = pooled(0)            # two 128-query passes (rq_scan.hip I8 = 3)
print("grid quads per wg:", nb / 256)
for rep in range(12):
    for v in (22, 25, 30, 31, 32, 33):
        got = pooled(v)
        bad = np.argwhere(got != ref)
        if len(bad):
            print(f"rep {rep} variant {v}: {len(bad)} differing (query, bin) pairs; queries {sorted(set(bad[:,0].tolist()))[:20]}")
            for qi, b in bad[:6]:
                wg = None
                for w in range(256):
                    lo, hi = w * nb // 256, (w + 1) * nb // 256
                    if lo <= b < hi: wg = (w, b - lo, hi - lo)
                print(f"   q {qi} (wave {qi // 32} group {(qi // 16) % 2} r16 {qi % 16}) bin {b} wg/pos/nloc {wg}: got {got[qi, b]:.6f} want {ref[qi, b]:.6f}")
        else:
            print(f"rep {rep} variant {v}: identical")
idx.close()
