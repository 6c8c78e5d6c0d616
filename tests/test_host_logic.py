"""Host-side mirror of reference rag_uq/streaming_index.py, checked on the CPU against fixtures
captured from the reference itself (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import bm25_oracle
from rag_uq_amd import streaming_index as si
from rag_uq_amd.embedders import CallableEmbedder, HashEmbedder


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


def _retriever(tmp_path, sc):
    r = si.HybridRetriever(bm25_persist_path=str(tmp_path / "x.pkl"), chroma_persist_path=str(tmp_path / "c"),
                           dense_index=StubDense(sc["dense"]))
    r.bm25_index = StubSparse(sc["bm25"])
    r.documents = {k: si.Document.from_dict(v) for k, v in sc["documents"].items()}
    return r


def test_hybrid_fusion_matches_reference(golden_dir, tmp_path):
    """reference :464-557 through stub backends: union, ghost ids, `max(...) or 1`, ordering, padding"""
    for sc in json.load(open(os.path.join(golden_dir, "g1_hybrid_fusion.json"))):
        r = _retriever(tmp_path, sc)
        res = r.hybrid_search("q", top_k=sc["top_k"], retrieval_pool_size=sc["pool"])
        got = [dict(doc_id=x.doc_id, text=x.text, bm25_score=x.bm25_score, dense_score=x.dense_score,
                    hybrid_score=x.hybrid_score, title=x.title, metadata=x.metadata) for x in res]
        assert got == sc["expected_results"], sc["name"]
        a = r.get_scores_for_router("q", num_passages=sc["num_passages"])
        e = sc["expected_router"]
        assert (a[0], a[1], a[2], a[3]) == (e["bm25_scores"], e["dense_scores"], e["doc_ids"], e["texts"]), sc["name"]
        # the batched entry point is the same computation
        b = r.get_scores_for_router_batch(["q", "q"], num_passages=sc["num_passages"])
        assert b[0] == a and b[1] == a


def test_records_match_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g6_records.json")))
    d1, d2 = si.Document(id="1", text="t"), si.Document(id="2", text="u", title="Title", metadata={"a": 1})
    assert [d1.to_dict(), d2.to_dict()] == g["to_dict"]
    rt = si.Document.from_dict({"id": "3", "text": "v"})
    assert dict(id=rt.id, text=rt.text, title=rt.title, metadata=rt.metadata) == g["from_dict_minimal"]
    rr = si.RetrievalResult(doc_id="x", text="y", bm25_score=1.0, dense_score=2.0)
    assert rr.hybrid_score is None and rr.title is None and rr.metadata is None


def test_hash_embedder_matches_reference_fallback(golden_dir):
    """reference :269-273 (HAS_OLLAMA False): 32 floats = sha256 bytes / 255"""
    emb = HashEmbedder()
    for c in json.load(open(os.path.join(golden_dir, "g5_hash_embedding.json"))):
        v = emb.embed([c["text"]])[0]
        assert v.shape == (32,)
        np.testing.assert_array_equal(v, np.asarray(c["embedding"], dtype=np.float32))


CORPUS = [
    ("d0", "The sky is blue and the sea is blue"), ("d1", "Grass is green in spring"), ("d2", "The sun is a star"),
    ("d3", "Blue whales live in the sea"), ("d4", "Stars shine at night in the sky"), ("d5", "Spring rain makes green grass grow"),
    ("d6", "A star is born"), ("d7", "the the the the"), ("d8", ""), ("d9", "Night sky full of stars and a blue moon"),
]


def test_bm25_restatement_matches_brute_force_oracle(tmp_path):
    idx = si.BM25Index(persist_path=str(tmp_path / "bm25.pkl"))
    docs = [si.Document(id=i, text=t) for i, t in CORPUS]
    assert idx.add_documents(docs[:4]) == 4
    assert idx.add_documents(docs[2:]) == 6          # two already known
    assert len(idx) == 10 and idx.get_document("d3").text == CORPUS[3][1]
    ids, texts = [i for i, _ in CORPUS], [t for _, t in CORPUS]
    for q in ["blue sky", "green grass in spring", "star", "the", "whales", "nothing matches here zebra", "blue blue moon", ""]:
        want = bm25_oracle.bm25_search(ids, texts, q, 5)
        got = idx.search(q, 5)
        assert [d for d, _ in got] == [d for d, _ in want], q
        np.testing.assert_allclose([s for _, s in got], [s for _, s in want], rtol=1e-12, atol=0)
    # persistence: same pickle keys as the reference (:192-198), reload gives the same answers
    import pickle
    data = pickle.load(open(tmp_path / "bm25.pkl", "rb"))
    assert sorted(data) == ["b", "doc_ids", "documents", "k1", "tokenized_corpus"]
    again = si.BM25Index(persist_path=str(tmp_path / "bm25.pkl"))
    assert again.search("blue sky", 5) == idx.search("blue sky", 5)
    assert si.BM25Index().search("anything", 3) == []


def test_streaming_index_checkpoint_and_resume(tmp_path):
    """reference :593-679: batches, skipped bad lines, checkpoint schema, resume by line offset"""
    path = tmp_path / "corpus.jsonl"
    lines = [json.dumps({"id": f"p{i}", "text": f"passage number {i}", "title": f"t{i}"}) for i in range(7)]
    lines.insert(3, "{not json")
    lines.insert(5, json.dumps({"id": "noid-text-missing"}))
    path.write_text("\n".join(lines) + "\n")
    r = si.HybridRetriever(bm25_persist_path=str(tmp_path / "b.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=None)
    r.dense_index = None
    s = si.StreamingIndex(r, checkpoint_path=str(tmp_path / "ck.json"), batch_size=3)
    it = s.stream_from_jsonl(str(path))
    assert next(it) == 3
    ck = json.load(open(tmp_path / "ck.json"))
    assert ck == {"last_offset": 3, "total_indexed": 3, "files_completed": []}      # committed right after the third good line
    # a fresh process resumes at line 3 (the malformed one) and finishes the file
    r2 = si.HybridRetriever(bm25_persist_path=str(tmp_path / "b.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=None)
    r2.dense_index = None
    s2 = si.StreamingIndex(r2, checkpoint_path=str(tmp_path / "ck.json"), batch_size=3)
    assert list(s2.stream_from_jsonl(str(path))) == [3, 1]
    ck = json.load(open(tmp_path / "ck.json"))
    assert ck == {"last_offset": 9, "total_indexed": 7, "files_completed": [str(path)]}
    assert s2.get_progress()["retriever_size"] == 7    # 3 restored from the BM25 pickle + 4 new (the reference would say 4)
    assert len(r2.bm25_index) == 7                     # BM25 pickle carried the first three over
    with pytest.raises(FileNotFoundError):
        list(s2.stream_from_jsonl(str(tmp_path / "missing.jsonl")))


def test_hybrid_retriever_without_gpu_degrades_like_reference(tmp_path):
    """No GPU here: dense side is disabled with a warning (reference :418-420), nothing raises"""
    from rag_uq_amd import _native
    if _native.device_count() > 0:
        pytest.skip("GPU present")
    r = si.HybridRetriever(bm25_persist_path=str(tmp_path / "b.pkl"), chroma_persist_path=str(tmp_path / "c"))
    assert r.dense_index is None
    stats = r.add_documents([si.Document(id=i, text=t) for i, t in CORPUS])
    assert stats == {"bm25_added": 10, "total_documents": 10}
    assert r.dense_search("blue") == []
    res = r.hybrid_search("blue whales", top_k=3)
    assert res and res[0].doc_id == "d3" and res[0].dense_score == 0.0
    with pytest.raises(ImportError):
        si.DenseIndex(persist_directory=str(tmp_path / "c"))


def test_callable_embedder_shape_check():
    e = CallableEmbedder(lambda ts: np.ones((len(ts), 5)), 5)
    assert e.embed(["a", "b"]).shape == (2, 5)
    with pytest.raises(ValueError):
        CallableEmbedder(lambda ts: np.ones((1, 5)), 5).embed(["a", "b"])


def test_document_store_is_restored_from_bm25_pickle(tmp_path):
    """a second process over the same persist paths can answer hybrid queries (the reference returns [] there)"""
    docs = [si.Document(id=i, text=t, title=f"T{i}") for i, t in CORPUS]
    r1 = si.HybridRetriever(bm25_persist_path=str(tmp_path / "b.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=None)
    r1.dense_index = None
    r1.add_documents(docs)
    r2 = si.HybridRetriever(bm25_persist_path=str(tmp_path / "b.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=None)
    r2.dense_index = None
    assert len(r2) == len(docs) and r2.documents["d3"].title == "Td3"
    assert [x.doc_id for x in r2.hybrid_search("blue whales", top_k=2)] == [x.doc_id for x in r1.hybrid_search("blue whales", top_k=2)]
