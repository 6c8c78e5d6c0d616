"""Drop-in check against the LIVE reference (build container only; skipped where /root/reference is absent,
e.g. on the GPU box).  The consumers of the hot path -- `RetrievalRouter.hybrid_rerank`
(rag_uq/router.py:179-202, used by experiments/run_evaluation.py:170-184) and `RAGEvaluator`
(rag_uq/eval_protocol.py) -- are imported from the reference UNCHANGED and fed with what this package's
HybridRetriever returns; and the reference's own HybridRetriever accepts this package's index objects."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rag_uq")), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REF)
    import torch  # noqa: F401
    import rag_uq.router as router
    import rag_uq.eval_protocol as ev
    import rag_uq.streaming_index as rsi
    yield router, ev, rsi
    sys.path.remove(REF)


class FakeDense:
    """Stands in for DenseIndex on a GPU-less machine: same duck type (search/add_documents/__len__)."""
    def __init__(self):
        self.docs = {}

    def add_documents(self, documents, batch_size=100):
        new = [d for d in documents if d.id not in self.docs]
        self.docs.update({d.id: d for d in new})
        return len(new)

    def search(self, query, top_k=10):
        toks = set(query.lower().split())
        scored = sorted(((len(toks & set(d.text.lower().split())) / (1 + len(toks)), i) for i, d in self.docs.items()), reverse=True)
        return [(i, s, self.docs[i].text) for s, i in scored[:top_k]]

    def __len__(self):
        return len(self.docs) or 1


CORPUS = [(f"d{i}", t) for i, t in enumerate([
    "the sky is blue on a clear day", "grass is green in the spring", "the sun is a bright star", "blue whales swim in the sea",
    "stars shine in the night sky", "rain makes the green grass grow", "a star is born in a nebula", "the sea is deep and blue",
    "night follows day", "clear water in a mountain lake", "mountains rise above the clouds", "clouds bring rain to the valley"])]


def test_unchanged_router_and_evaluator_consume_our_retriever(ref, tmp_path):
    router_mod, ev_mod, _ = ref
    import torch
    from rag_uq_amd import streaming_index as si
    r = si.HybridRetriever(bm25_persist_path=str(tmp_path / "b.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=FakeDense())
    r.add_documents([si.Document(id=i, text=t) for i, t in CORPUS])
    torch.manual_seed(0)
    model = router_mod.RetrievalRouter(router_mod.RouterConfig()).eval()
    retrieved, gold = [], []
    for question, answer_doc in [("what color is the sky", "d0"), ("where do blue whales swim", "d3"), ("what makes grass grow", "d5")]:
        bm25, dense, ids, texts = r.get_scores_for_router(question, num_passages=10)          # run_evaluation.py:165-167
        assert len(bm25) == len(dense) == len(ids) == len(texts) == 10
        bm25_t, dense_t = torch.tensor([bm25]), torch.tensor([dense])                          # run_evaluation.py:170-173
        with torch.no_grad():
            out = model.hybrid_rerank(bm25_t, dense_t, top_k=5)
        idxs = out[0] if isinstance(out, tuple) else out
        order = [int(i) for i in idxs.reshape(-1)[:5]]
        retrieved.append([ids[i] for i in order if ids[i]])
        gold.append([answer_doc])
    evaluator = ev_mod.RAGEvaluator(output_dir=str(tmp_path / "res"))
    rec = np.mean([evaluator._recall_at_k(a, b, 5) for a, b in zip(retrieved, gold)])
    assert 0.0 <= rec <= 1.0
    # identical to the oracle-side restatement of the metric
    from oracle import dense_oracle as orc
    ids_all = {d: i for i, (d, _) in enumerate(CORPUS)}
    found = np.full((3, 5), -1, dtype=np.int64)
    for b, lst in enumerate(retrieved):
        for j, d in enumerate(lst[:5]):
            found[b, j] = ids_all[d]
    assert orc.recall_at_k(found, np.array([[ids_all[g[0]]] for g in gold])) == pytest.approx(rec)


def test_reference_hybrid_retriever_accepts_our_indexes(ref, tmp_path):
    """INTEGRATION.md option B: the reference's HybridRetriever with this package's BM25Index + a dense index object"""
    _, _, rsi = ref
    from rag_uq_amd import streaming_index as si
    theirs = rsi.HybridRetriever(bm25_persist_path=str(tmp_path / "x.pkl"), chroma_persist_path=str(tmp_path / "c"))
    theirs.bm25_index = si.BM25Index()
    theirs.dense_index = FakeDense()
    docs = [rsi.Document(id=i, text=t) for i, t in CORPUS]
    theirs.bm25_index.add_documents(docs)            # (their add_documents skips empty indexes: `if self.bm25_index:` defect)
    theirs.dense_index.add_documents(docs)
    theirs.documents = {d.id: d for d in docs}
    ours = si.HybridRetriever(bm25_persist_path=str(tmp_path / "y.pkl"), chroma_persist_path=str(tmp_path / "c2"), dense_index=FakeDense())
    ours.add_documents([si.Document(id=i, text=t) for i, t in CORPUS])
    for q in ["blue sea", "green grass rain", "night sky stars"]:
        a = theirs.hybrid_search(q, top_k=5)
        b = ours.hybrid_search(q, top_k=5)
        # entries tied at hybrid 0.0 are cut at top_k in set-iteration order (reference :489,:521): compare the rest
        key = lambda r: (-(r.hybrid_score or 0), r.doc_id)
        pos = lambda rs: [(x.doc_id, x.bm25_score, x.dense_score, x.hybrid_score) for x in sorted(rs, key=key) if x.hybrid_score > 0]
        assert pos(a) == pos(b) and len(pos(a)) >= 2
