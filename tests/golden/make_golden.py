"""Generates tests/golden/*.json by RUNNING THE REFERENCE in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

What is captured (inputs + the reference's outputs, nothing of its source):
  g1_hybrid_fusion.json   rag_uq.streaming_index.HybridRetriever.hybrid_search / get_scores_for_router
                          (streaming_index.py:464-557) driven by stub sparse/dense backends
  g3_retrieval_metrics.json  rag_uq.eval_protocol.RAGEvaluator._recall_at_k / _reciprocal_rank / _ndcg_at_k
  g5_hash_embedding.json  rag_uq.streaming_index.DenseIndex._get_embedding fallback (:269-273), HAS_OLLAMA False
  g6_records.json         Document.to_dict / from_dict (:62-77)
  g2_router.json          rag_uq.router.RetrievalRouter(RouterConfig()) with torch.manual_seed(0), .eval(): its state dict and
                          forward(update_stats=False) / hybrid_rerank outputs on fixed [4,20] / [1,10] / [1,100] score tensors
                          (router.py:44-202), with batch-wise and with running-statistics normalisation -- pins the
                          consumer of the hot path that BASELINE.json configs[4] feeds
  g4_oracle_dense.json    the oracle's own answers on seeded inputs (regression pin of oracle/dense_oracle.py;
                          NOT a reference output: the reference's dense arithmetic lives in ChromaDB, absent here)
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import rag_uq.streaming_index as ref  # noqa: E402
from rag_uq.eval_protocol import RAGEvaluator  # noqa: E402


class StubSparse:
    def __init__(self, pairs):
        self.pairs = pairs

    def search(self, query, top_k):
        return [tuple(p) for p in self.pairs[:top_k]]


class StubDense:
    def __init__(self, pairs):
        self.pairs = pairs

    def search(self, query, top_k):
        return [(d, s, "text of " + d) for d, s in self.pairs[:top_k]]


def docs(ids):
    return {i: {"id": i, "text": f"passage {i} body", "title": f"T{i}", "metadata": {"n": len(i)}} for i in ids}


SCENARIOS = [
    dict(name="overlap", doc_ids=["a", "b", "c", "d", "e", "f"],
         bm25=[["a", 7.5], ["b", 5.25], ["c", 2.0], ["d", 0.5]], dense=[["c", 0.91], ["a", 0.80], ["e", 0.55], ["f", 0.31]],
         top_k=4, pool=50, num_passages=6),
    dict(name="ghost_id_not_in_documents", doc_ids=["a", "b", "c"],
         bm25=[["a", 3.0], ["zz", 9.0], ["b", 1.0]], dense=[["yy", 0.99], ["c", 0.75], ["a", 0.5]],
         top_k=10, pool=50, num_passages=5),
    dict(name="all_negative_dense", doc_ids=["a", "b", "c", "d"],
         bm25=[["a", 4.0], ["b", 1.0]], dense=[["c", -0.10], ["d", -0.40], ["a", -0.25]],
         top_k=4, pool=50, num_passages=4),
    dict(name="empty_sparse", doc_ids=["a", "b", "c"], bm25=[], dense=[["b", 0.7], ["a", 0.6], ["c", 0.2]],
         top_k=2, pool=50, num_passages=3),
    dict(name="empty_dense", doc_ids=["a", "b", "c"], bm25=[["c", 2.5], ["a", 1.5]], dense=[],
         top_k=5, pool=50, num_passages=4),
    dict(name="both_empty", doc_ids=["a"], bm25=[], dense=[], top_k=3, pool=50, num_passages=3),
    dict(name="pool_smaller_than_lists", doc_ids=["a", "b", "c", "d", "e"],
         bm25=[["a", 5.0], ["b", 4.0], ["c", 3.0], ["d", 2.0]], dense=[["b", 0.9], ["d", 0.8], ["c", 0.7], ["e", 0.6]],
         top_k=5, pool=2, num_passages=5),
    dict(name="zero_max_scores", doc_ids=["a", "b"], bm25=[["a", 0.0]], dense=[["b", 0.0], ["a", 0.3]],
         top_k=2, pool=50, num_passages=2),
]


def run_fusion():
    out = []
    tmp = tempfile.mkdtemp()
    for sc in SCENARIOS:
        r = ref.HybridRetriever(bm25_persist_path=os.path.join(tmp, "x.pkl"), chroma_persist_path=os.path.join(tmp, "c"))
        r.bm25_index = StubSparse(sc["bm25"])
        r.dense_index = StubDense(sc["dense"])
        r.documents = {k: ref.Document.from_dict(v) for k, v in docs(sc["doc_ids"]).items()}
        res = r.hybrid_search("q", top_k=sc["top_k"], retrieval_pool_size=sc["pool"])
        hs = [x.hybrid_score for x in res]
        assert len(set(hs)) == len(hs), f"scenario {sc['name']} has tied hybrid scores: order would depend on set iteration"
        arrays = r.get_scores_for_router("q", num_passages=sc["num_passages"])
        out.append(dict(sc, documents=docs(sc["doc_ids"]),
                        expected_results=[dict(doc_id=x.doc_id, text=x.text, bm25_score=x.bm25_score, dense_score=x.dense_score,
                                               hybrid_score=x.hybrid_score, title=x.title, metadata=x.metadata) for x in res],
                        expected_router=dict(bm25_scores=arrays[0], dense_scores=arrays[1], doc_ids=arrays[2], texts=arrays[3])))
    return out


def run_metrics():
    ev = RAGEvaluator(output_dir=tempfile.mkdtemp())
    cases = [
        dict(retrieved=["d1", "d2", "d3", "d4", "d5"], relevant=["d2", "d9"], k=3),
        dict(retrieved=["d1", "d2", "d3", "d4", "d5"], relevant=["d5"], k=5),
        dict(retrieved=["d1", "d2"], relevant=[], k=10),
        dict(retrieved=[], relevant=["d1"], k=10),
        dict(retrieved=["d7", "d8", "d9", "d1", "d2", "d3", "d4", "d5", "d6", "d0"], relevant=["d0", "d1", "d2"], k=10),
        dict(retrieved=["d3", "d3", "d1"], relevant=["d1", "d3"], k=2),
    ]
    for c in cases:
        c["recall_at_k"] = ev._recall_at_k(c["retrieved"], c["relevant"], c["k"])
        c["precision_at_k"] = ev._precision_at_k(c["retrieved"], c["relevant"], c["k"])
        c["reciprocal_rank"] = ev._reciprocal_rank(c["retrieved"], c["relevant"])
        c["ndcg_at_10"] = ev._ndcg_at_k(c["retrieved"], {d: 1.0 for d in c["relevant"]}, 10)
    return cases


def run_hash_embedding():
    assert ref.HAS_OLLAMA is False, "fixture is for the no-ollama branch"
    texts = ["", "hello", "The sky is blue.", "What color is the sky?", "naïve café ☕", "x" * 1000]
    return [dict(text=t, embedding=ref.DenseIndex._get_embedding(None, t)) for t in texts]


def run_records():
    d1 = ref.Document(id="1", text="t")
    d2 = ref.Document(id="2", text="u", title="Title", metadata={"a": 1})
    rt = ref.Document.from_dict({"id": "3", "text": "v"})
    return dict(to_dict=[d1.to_dict(), d2.to_dict()], from_dict_minimal=dict(id=rt.id, text=rt.text, title=rt.title, metadata=rt.metadata))


def run_router():
    import torch
    from rag_uq.router import RetrievalRouter, RouterConfig
    torch.manual_seed(0)
    router = RetrievalRouter(RouterConfig()).eval()
    state = {k: v.detach().cpu().numpy().tolist() for k, v in router.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    cases = []
    for shape, scale, k in (((4, 20), 1.0, 10), ((1, 10), 1.0, 10), ((1, 100), 1.0, 10), ((2, 100), 8.0, 100), ((1, 10), 1.0, 3)):
        bm25 = (torch.rand(shape, generator=g) * 12.0 * scale).float()
        dense = (torch.rand(shape, generator=g) * 0.6 + 0.2).float()
        if shape == (1, 10):
            bm25[0, 7:] = 0.0; dense[0, 8:] = 0.0           # the zero padding get_scores_for_router appends (:539-557)
        for stats in (False, True):
            router.stats_initialized = stats
            if stats:
                router.bm25_mean.fill_(3.5); router.bm25_std.fill_(2.25); router.dense_mean.fill_(0.45); router.dense_std.fill_(0.125)
            with torch.no_grad():
                w = router(bm25, dense, update_stats=False)
                top_s, top_i = router.hybrid_rerank(bm25, dense, top_k=k)
                # the evaluation loop's own rerank (experiments/run_evaluation.py:170-184)
                hyb = w * dense + (1 - w) * bm25
                order = [torch.argsort(hyb[b], descending=True).tolist() for b in range(shape[0])]
            cases.append(dict(shape=list(shape), top_k=k, stats_initialized=stats,
                              stats=dict(bm25_mean=float(router.bm25_mean), bm25_std=float(router.bm25_std),
                                         dense_mean=float(router.dense_mean), dense_std=float(router.dense_std)),
                              bm25=bm25.tolist(), dense=dense.tolist(), weights=w.tolist(),
                              rerank_scores=top_s.tolist(), rerank_indices=top_i.tolist(), eval_loop_order=order))
    return dict(config=dict(hidden_dim=64, dropout=0.1, num_layers=2, use_batch_norm=False), state_dict=state, cases=cases)


def run_oracle_pin():
    from oracle import dense_oracle as orc
    out = []
    for n, dim, B, k, seed in [(300, 768, 4, 10, 1234), (50, 32, 3, 5, 7), (1000, 384, 2, 20, 99)]:
        x16 = orc.synthetic_corpus(n, dim, seed=seed)
        q = orc.synthetic_queries(B, dim, seed=seed + 1)
        s, r = orc.dense_topk(q, x16, k)
        out.append(dict(n=n, dim=dim, B=B, k=k, seed=seed, rows=r.tolist(), scores=[[float(v) for v in row] for row in s]))
    return out


def main():
    dumps = {
        "g1_hybrid_fusion.json": run_fusion(),
        "g2_router.json": run_router(),
        "g3_retrieval_metrics.json": run_metrics(),
        "g5_hash_embedding.json": run_hash_embedding(),
        "g6_records.json": run_records(),
        "g4_oracle_dense.json": run_oracle_pin(),
    }
    for name, obj in dumps.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
        print("wrote", name)


if __name__ == "__main__":
    main()
