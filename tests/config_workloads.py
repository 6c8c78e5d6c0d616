"""BASELINE.json configs[2], [3] and [4] at their shapes on ONE MI355X -- measurement + parity harness.

Used by tests/test_gpu_configs.py (assertions) and by `bench.py --workload config2|config3|config4` (one JSON line each).
Data and weights named by those configs (Wikipedia passage embeddings, nomic-embed-text weights, NQ-dev-500) do not exist
offline: every result carries "data": "synthetic stand-in" and says what stands in for what.  The oracle (oracle/*.py) is
used only as the checker, after the timed regions.
"""
from __future__ import annotations

import os
import sys
import time
from typing import Dict, List

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this file lives in tests/)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _gen_rows(torch, dev, n: int, seed: int):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: 10M x 768 fp16 corpus row-sharded 8 ways; per-shard search -> keys -> merge (here: 8 shards on one GPU, the
# exchange step is a device-to-device stack instead of the RCCL all-gather of rag_uq_amd.distributed)
# ---------------------------------------------------------------------------------------------------------------------
def run_config2(n_total: int = 10_000_000, shards: int = 8, B: int = 64, k: int = 10, steps: int = 12, oracle_queries: int = 3,
                compare_single: bool = True) -> Dict:
    import torch
    import rag_uq_amd  # noqa: F401
    from rag_uq_amd import _native as nat
    from oracle import dense_oracle as orc

    dev = torch.device("cuda:0")
    per = n_total // shards
    chunk = 125_000
    idxs = []
    big = nat.NativeIndex(768, 0) if compare_single else None
    if big is not None:
        big.reserve(n_total)
    for s in range(shards):
        idx = nat.NativeIndex(768, 0)
        idx.reserve(per)
        idx.set_row_offset(s * per)
        for c in range(0, per, chunk):
            m = min(chunk, per - c)
            x = _gen_rows(torch, dev, m, 1236 + s * 1000 + c // chunk)       # seed 1236 + shard, chunked
            idx.add_f16_device(x, m)
            if big is not None:
                big.add_f16_device(x, m)
            del x
        idxs.append(idx)
    # queries: copies of stored rows at the shard edges (must come back at rank 1 with score 1) + Gaussian ones
    planted = [0, per - 1, per, 3 * per + 17, n_total // 2 - 1, n_total // 2, n_total - per, n_total - 1]
    q = orc.synthetic_queries(B, 768, seed=4321)
    for j, gid in enumerate(planted):
        q[j] = idxs[gid // per].get_rows_f16(gid % per, 1)[0].astype(np.float32)
    dq = torch.from_numpy(q).to(dev)

    def search_sharded(kk: int):
        keys = torch.zeros((shards, B, kk), device=dev, dtype=torch.int64)
        sc = torch.empty((B, kk), device=dev); rw = torch.empty((B, kk), device=dev, dtype=torch.int64)
        st = torch.empty((shards, B), device=dev, dtype=torch.int32)
        for s, idx in enumerate(idxs):
            idx.search_device(dq, B, kk, 0, sc, rw, keys[s], st[s], 0)
        repaired = 0
        if int(st.sum()) != 0:
            for s, idx in enumerate(idxs):
                repaired += idx.search_fixup_device(dq, B, kk, 0, sc, rw, keys[s], st[s], 0)
        ms = torch.empty((B, kk), device=dev); mr = torch.empty((B, kk), device=dev, dtype=torch.int64)
        nat.merge_keys_device(keys.permute(1, 0, 2).contiguous(), shards * kk, B, kk, ms, mr, None, 0)
        torch.cuda.synchronize()
        return ms.cpu().numpy(), mr.cpu().numpy(), repaired

    s10, r10, rep10 = search_sharded(k)
    s50, r50, rep50 = search_sharded(50)
    used8 = sum(int(i.get_option("scan8_used")) for i in idxs)
    t0 = time.perf_counter()
    for _ in range(steps):
        search_sharded(k)
    dt = (time.perf_counter() - t0) / steps
    int8_scan = sum(int(i.get_option("scan8_used")) for i in idxs) - used8 == steps * shards     # every timed search scanned the int8 image
    row_bytes = 768 if int8_scan else 1536
    out = {"workload": f"{n_total}x768 fp16 in {shards} row shards of {per} on one GPU, batch-{B}, top-{k}: per-shard search -> keys -> device merge",
           "data": "synthetic stand-in (Gaussian unit rows, seeds 1236+shard)", "ms_per_batch": dt * 1e3, "queries_per_s": B / dt,
           "scanned": "int8 image of every shard, 768 B per row" if int8_scan else "fp16 rows, 1536 B per row",
           "achieved_GBs": n_total * row_bytes / dt / 1e9, "repaired_queries": rep10 + rep50,
           "planted_at_rank_1": bool(r10[: len(planted), 0].tolist() == planted and np.allclose(s10[: len(planted), 0], 1.0, atol=1e-6)),
           "top10_is_prefix_of_top50": bool(np.array_equal(r50[:, :k], r10) and np.array_equal(s50[:, :k], s10))}
    if big is not None:
        bs, br = big.search(q, k)
        out["merged_equals_single_index"] = bool(np.array_equal(br, r10) and np.array_equal(bs, s10))
        big.close()
    if oracle_queries:
        sel = [0, 3, len(planted)][:oracle_queries] if oracle_queries <= 3 else list(range(oracle_queries))
        parts = []
        for s, idx in enumerate(idxs):
            x16 = idx.get_rows_f16(0, per)
            parts.append(orc.dense_topk(q[sel], x16, k, row_offset=s * per))
            del x16
        gs, gr = orc.merge_topk(parts, k)
        out["oracle_queries"] = len(sel)
        out["ids_exact_vs_oracle"] = bool(np.array_equal(r10[sel], gr))
        out["max_abs_score_err_vs_oracle"] = float(np.abs(s10[sel] - gs).max())
    for idx in idxs:
        idx.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: 256 raw text queries -> NomicBert forward on PyTorch-ROCm -> search over 1M passage vectors
# ---------------------------------------------------------------------------------------------------------------------
def run_config3(n: int = 1_000_000, n_queries: int = 256, k: int = 10, num_layers: int = 12, reps: int = 5) -> Dict:
    import torch
    import rag_uq_amd  # noqa: F401
    from rag_uq_amd import streaming_index as si
    from rag_uq_amd.embedders import NomicBertEmbedder
    from oracle import dense_oracle as orc

    torch.manual_seed(0)
    emb = NomicBertEmbedder(random_init=True, num_layers=num_layers, device="cuda:0", dtype="float16", batch_size=256)
    idx = si.DenseIndex(persist_directory="", embedder=emb, load_persisted=False, auto_persist=False)
    rng = np.random.default_rng(1235)
    chunk = 125_000
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        idx.add_vectors([f"d{i}" for i in range(lo, lo + m)], rng.standard_normal((m, 768), dtype=np.float32), texts=[""] * m)
    queries = [f"question {i}: what is known about topic {i * 7919 % 1000} and the river number {i % 13}?" for i in range(n_queries)]
    qv = emb.embed(queries)                       # warm-up of the encoder (and the vectors the oracle will use)
    idx.search_vectors(qv, k)                     # warm-up of the search
    torch.cuda.synchronize()
    idx.search_batch(queries, k)                  # warm-up of the device route (its output block)
    torch.cuda.synchronize()
    t_enc = t_search = t_search_host = t_e2e = 0.0
    for _ in range(reps):
        t0 = time.perf_counter(); qv2 = emb.embed(queries); t_enc += time.perf_counter() - t0
        t0 = time.perf_counter(); idx.search_vectors(qv2, k); t_search_host += time.perf_counter() - t0
        dq = emb.embed_device(queries); torch.cuda.synchronize()
        # the search as search_batch runs it: the query matrix stays in HBM (rq_search_device), one D2H of rows + scores, tuple assembly
        t0 = time.perf_counter(); idx.search_device_vectors(dq, k); t_search += time.perf_counter() - t0
        t0 = time.perf_counter(); res = idx.search_batch(queries, k); t_e2e += time.perf_counter() - t0
    x16 = idx._index.get_rows_f16(0, n)
    gs, gr = orc.dense_topk(qv, x16, k)
    got_r = np.array([[int(d[1:]) for d, _, _ in r] for r in res])
    got_s = np.array([[s for _, s, _ in r] for r in res], dtype=np.float32)
    out = {"workload": f"{n_queries} raw text queries -> NomicBert ({num_layers} layers, 768 hidden, fp16; hipBLASLt GEMMs + fused gfx950 kernels, csrc/rq_encoder.hip) -> exact top-{k} over {n}x768 fp16",
           "data": "synthetic stand-in (random-init NomicBert + byte-level tokenizer: no nomic-embed-text weights offline; Gaussian passage vectors)",
           "encode_ms": t_enc / reps * 1e3, "search_ms": t_search / reps * 1e3, "search_host_buffers_ms": t_search_host / reps * 1e3, "end_to_end_ms": t_e2e / reps * 1e3,
           "text_queries_per_s": n_queries / (t_e2e / reps),
           "ids_exact_vs_oracle": bool(np.array_equal(got_r, gr)), "max_abs_score_err_vs_oracle": float(np.abs(got_s - gs).max()),
           "oracle_queries": n_queries}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# configs[4]: GPU dense top-100 + CPU BM25 top-100 -> fusion -> router -> Recall@10, on a synthetic QA set
# ---------------------------------------------------------------------------------------------------------------------
class OracleDenseIndex:
    """Duck type of DenseIndex (`search`, `search_batch`, `__len__`) answered by oracle/dense_oracle.py on the SAME stored
    fp16 vectors: the CPU side of the config-4 parity check."""

    def __init__(self, x16: np.ndarray, ids: List[str], texts: List[str], embedder):
        self.x16, self.ids, self.texts, self.embedder = x16, ids, texts, embedder

    def __len__(self):
        return len(self.ids)

    def search_batch(self, queries, top_k=10):
        from oracle import dense_oracle as orc
        s, r = orc.dense_topk(np.asarray(self.embedder.embed(list(queries)), np.float32), self.x16, min(top_k, len(self.ids)))
        return [[(self.ids[j], float(v), self.texts[j]) for v, j in zip(sv, rv) if j >= 0] for sv, rv in zip(s, r)]

    def search(self, query, top_k=10):
        return self.search_batch([query], top_k)[0]


def synthetic_qa(n_passages: int, n_questions: int, seed: int = 7):
    """Passages of 40 words from a 5000-word vocabulary (Zipf-like); question i = 8 words of its answer passage + 4 others."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, 5001) ** 0.8
    p /= p.sum()
    words = rng.choice(5000, size=(n_passages, 40), p=p)
    passages = [" ".join(f"w{w}" for w in row) for row in words]
    answers = rng.choice(n_passages, size=n_questions, replace=False)
    questions = []
    for a in answers:
        own = rng.choice(words[a], size=8, replace=False)
        other = rng.choice(5000, size=4, p=p)
        questions.append(" ".join(f"w{w}" for w in np.concatenate([own, other])))
    return passages, questions, [f"p{a}" for a in answers]


def rerank_with_router(router, arrays, top: int = 10) -> List[str]:
    """experiments/run_evaluation.py:165-199 for one question: router weights -> hybrid score -> reorder -> ids (padding dropped)."""
    bm25, dense, ids, _ = arrays
    return [ids[i] for i in router.eval_loop_order(bm25, dense) if ids[i]][:top]


def run_config4(n_passages: int = 50_000, n_questions: int = 500, pool: int = 100, num_passages: int = 100, golden_dir: str = None) -> Dict:
    import json
    import tempfile
    import rag_uq_amd  # noqa: F401
    from rag_uq_amd import streaming_index as si
    from rag_uq_amd.embedders import RandomProjectionEmbedder
    from oracle import router_oracle as ro

    golden_dir = golden_dir or os.path.join(ROOT, "tests", "golden")
    router = ro.RouterOracle(json.load(open(os.path.join(golden_dir, "g2_router.json")))["state_dict"])
    passages, questions, gold = synthetic_qa(n_passages, n_questions)
    emb = RandomProjectionEmbedder(768)
    tmp = tempfile.mkdtemp()
    dense = si.DenseIndex(persist_directory="", embedder=emb, load_persisted=False, auto_persist=False)
    r = si.HybridRetriever(bm25_persist_path=os.path.join(tmp, "bm25.pkl"), chroma_persist_path=os.path.join(tmp, "c"), dense_index=dense)
    r.bm25_index.persist_path = None                                   # (index build persistence is measured elsewhere)
    docs = [si.Document(id=f"p{i}", text=t) for i, t in enumerate(passages)]
    t0 = time.perf_counter()
    for lo in range(0, n_passages, 5000):
        r.add_documents(docs[lo: lo + 5000], batch_size=5000)
    t_build = time.perf_counter() - t0
    # per corpus state, not per batch: the CSR form of the posting lists (+ the librq_bm25 handle) and the key space of the batched fusion
    # are built by the first batched call after an add; timed on their own here, then every later batch runs against them
    t0 = time.perf_counter()
    r.get_scores_for_router_batch(questions[:8], num_passages=num_passages, retrieval_pool_size=pool)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    gpu_arrays = r.get_scores_for_router_batch(questions, num_passages=num_passages, retrieval_pool_size=pool)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    r.bm25_index.search_batch_rows(questions, pool)
    t_bm25 = time.perf_counter() - t0
    t0 = time.perf_counter()
    dense_only = dense.search_batch(questions, pool)
    t_dense = time.perf_counter() - t0
    gpu_lists = [rerank_with_router(router, a) for a in gpu_arrays]
    # the same harness with the dense side answered by the oracle on the same stored vectors
    x16 = dense._index.get_rows_f16(0, n_passages)
    r.dense_index = OracleDenseIndex(x16, [f"p{i}" for i in range(n_passages)], passages, emb)
    cpu_arrays = r.get_scores_for_router_batch(questions, num_passages=num_passages, retrieval_pool_size=pool)
    cpu_lists = [rerank_with_router(router, a) for a in cpu_arrays]
    rec = lambda lists: float(np.mean([ro.recall_at_k(l, [g], 10) for l, g in zip(lists, gold)]))
    fused_only = lambda arrays: [[d for d in a[2] if d][:10] for a in arrays]
    return {"workload": f"{n_questions} questions over {n_passages} passages: GPU dense top-{pool} + CPU BM25 top-{pool} -> fusion (top {num_passages}) "
                        f"-> router (G2-pinned restatement, batch-wise normalisation) -> Recall@10",
            "data": "synthetic stand-in (planted-answer QA over a 5000-word vocabulary, RandomProjectionEmbedder; NQ-dev-500 and nomic weights are absent offline)",
            "recall_at_10_router_gpu_dense": rec(gpu_lists), "recall_at_10_router_oracle_dense": rec(cpu_lists),
            "recall_at_10_fusion_only": rec(fused_only(gpu_arrays)),
            "recall_at_10_dense_only": float(np.mean([ro.recall_at_k([d for d, _, _ in l], [g], 10) for l, g in zip(dense_only, gold)])),
            "id_lists_identical": bool(gpu_lists == cpu_lists),
            "router_inputs_identical": bool(all(a[2] == b[2] and np.allclose(a[1], b[1], atol=1e-6) and a[0] == b[0] for a, b in zip(gpu_arrays, cpu_arrays))),
            "index_build_s": t_build, "first_batched_call_ms": t_first * 1e3, "bm25_top100_batch_ms": t_bm25 * 1e3, "hybrid_batch_ms": t_gpu * 1e3, "dense_top100_batch_ms": t_dense * 1e3,
            "questions_per_s_hybrid": n_questions / t_gpu}


if __name__ == "__main__":
    import json
    which = sys.argv[1] if len(sys.argv) > 1 else "config2"
    print(json.dumps({"config2": run_config2, "config3": run_config3, "config4": run_config4}[which]()), flush=True)
