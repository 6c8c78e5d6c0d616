import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
from oracle import dense_oracle as orc
x16 = orc.synthetic_corpus(60_000, 768, seed=81)
x16[200:260] = x16[7]
idx = nat.NativeIndex(768, 0)
idx.add_f16(x16[:50_011])
idx.set_option("scan8", 2)
q0 = orc.synthetic_queries(65, 768, seed=75); idx.search(q0, 10)
idx.add_f16(x16[50_011:])
B, k = 200, 32
q = orc.synthetic_queries(B, 768, seed=B + k); q[1] = 0; q[2] = x16[7].astype(np.float32)
gs, gr = orc.dense_topk(q, x16, k)
for name, opts in (("default", {}), ("bin_bound=0", {"bin_bound": 0}), ("wide256_8=0", {"wide256_8": 0}), ("wide256_8=22", {"wide256_8": 22}), ("tail_local=0", {"tail_local": 0})):
    idx.set_option("bin_bound", 1); idx.set_option("wide256_8", 20); idx.set_option("tail_local", 1)
    for o, v in opts.items(): idx.set_option(o, v)
    dq = torch.from_numpy(q).cuda(); sc = torch.empty((B, k), device="cuda"); rw = torch.empty((B, k), device="cuda", dtype=torch.int64); st = torch.zeros((B,), device="cuda", dtype=torch.int32)
    idx.search_device(dq, B, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
    print(" uncertified before fixup:", int(st.sum()))
    idx.search_fixup_device(dq, B, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
    s, r = sc.cpu().numpy(), rw.cpu().numpy()
    idx.search_device(dq, B, k, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
    bad = np.argwhere(r != gr)
    print(name, "mismatches:", len(bad), bad[:6].tolist(), flush=True)
    if len(bad) and name == "default":
        qi = int(bad[0][0])
        missing = sorted(set(gr[qi].tolist()) - set(r[qi].tolist()))
        print(" query", qi, "missing rows", missing, "exact scores", [float(gs[qi][list(gr[qi]).index(m)]) for m in missing], "k-th exact", float(gs[qi][-1]))
        be = idx.debug_bin_err((60_000 + 63) // 64)
        pooled = idx.debug_pooled(qi, (60_000 + 63) // 64)
        ex = orc.exact_scores(q[qi:qi + 1], x16)[0]
        for m in missing:
            b = m // 64
            qn = q[qi].astype(np.float64); qa = np.abs(qn).max(); sq = float(np.float32(qa) / np.float32(127.0))
            qq = np.clip(np.rint(q[qi] * (np.float32(127.0) / np.float32(qa))), -127, 127)
            eq = float(np.sqrt(((qn - sq * qq) ** 2).sum()) / np.sqrt((qn * qn).sum()))
            print("  e_q", eq, "bound with bin err", eq + (1 + eq) * (float(be[b]) + 2e-5), "with shard err", eq + (1 + eq) * (float(be.max()) + 2e-5))
            for bb in range(max(b - 1, 0), b + 2): print("   bin", bb, "approx", float(pooled[bb]), "exact", float(ex[bb * 64:(bb + 1) * 64].max()))
            print("  row", m, "bin", b, "binerr", float(be[b]), "max binerr", float(be.max()), "approx bin max", float(pooled[b]), "exact bin max", float(ex[b * 64:(b + 1) * 64].max()), "exact row", float(ex[m]))
idx.close()
