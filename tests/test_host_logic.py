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
    idx.save()                                        # (adds go to the append-only log; the snapshot is written on save / close)
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


def test_bm25_persistence_is_incremental_and_reference_compatible(tmp_path):
    """add_documents appends to <path>.log.jsonl instead of re-pickling the whole index (reference :141-146 + :185-201 is
    O(N^2) over an indexing run); snapshots keep the reference's pickle layout; reload = snapshot + log replay; a torn last
    log line is dropped; snapshot bytes written over a run stay linear in the corpus."""
    import pickle
    p = tmp_path / "bm25.pkl"
    docs = [si.Document(id=f"p{i}", text=f"passage {i} about topic {i % 17} word{i * 7 % 101}", title=f"T{i}") for i in range(3_000)]
    b = si.BM25Index(str(p))
    b.SNAPSHOT_MIN_DOCS = 400                                  # (instance attribute: small thresholds for the test)
    snapshots, written = 0, 0
    for lo in range(0, 3_000, 100):
        before = p.stat().st_mtime_ns if p.exists() else None
        b.add_documents(docs[lo: lo + 100])
        if p.exists() and p.stat().st_mtime_ns != before:
            snapshots += 1
            written += p.stat().st_size
    assert 2 <= snapshots <= 5                                 # at 400, 800, 1600 (+ close): each as large as everything before
    assert written < 4 * len(pickle.dumps({"documents": {d.id: d.to_dict() for d in docs}, "tokenized_corpus": [d.text.lower().split() for d in docs]}))
    fresh = si.BM25Index(str(p))                               # snapshot + replay of the log
    assert len(fresh) == 3_000 and fresh.doc_ids == b.doc_ids
    q = "passage about topic 5 word35"
    assert fresh.search(q, 10) == b.search(q, 10)
    with open(str(p) + ".log.jsonl", "a") as f:
        f.write('{"id": "torn", "text": "half a li')           # a process killed in the middle of a write
    assert len(si.BM25Index(str(p))) == 3_000
    b.close()                                                  # final snapshot, log gone
    assert not (tmp_path / "bm25.pkl.log.jsonl").exists()
    data = pickle.load(open(p, "rb"))                          # the reference's own loader (:203-222) reads exactly this layout
    assert sorted(data) == ["b", "doc_ids", "documents", "k1", "tokenized_corpus"] and len(data["doc_ids"]) == 3_000
    assert data["documents"]["p7"] == {"id": "p7", "text": docs[7].text, "title": "T7", "metadata": {}}
    # snapshot_every = 1: the reference's behaviour (a full pickle after every add)
    p1 = tmp_path / "every.pkl"
    e = si.BM25Index(str(p1), snapshot_every=1)
    e.add_documents(docs[:5])
    assert len(pickle.load(open(p1, "rb"))["doc_ids"]) == 5 and not (tmp_path / "every.pkl.log.jsonl").exists()
    # a HybridRetriever over the same files restores its document store, log included
    r = si.HybridRetriever(bm25_persist_path=str(tmp_path / "h.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=StubDense([]))
    r.dense_index = None                                       # (sparse side only: no GPU in this tier)
    r.add_documents(docs[:50])
    r2 = si.HybridRetriever(bm25_persist_path=str(tmp_path / "h.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=StubDense([]))
    assert len(r2) == 50 and r2.documents["p3"].text == docs[3].text


def test_bm25_snapshot_loader_resolves_no_classes(tmp_path):
    """The snapshot is plain containers: the loader refuses anything that needs a class lookup (a pickle that would run
    code on load is rejected instead of executed)."""
    import pickle

    class Boom:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    p = tmp_path / "evil.pkl"
    pickle.dump({"documents": {}, "doc_ids": [], "tokenized_corpus": [Boom()], "k1": 1.5, "b": 0.75}, open(p, "wb"))
    with pytest.raises(pickle.UnpicklingError, match="must not reference"):
        si.BM25Index(str(p))


def test_dense_collection_files_are_repaired_to_the_last_commit(tmp_path):
    """<collection>.f16 / .docs.jsonl / .meta: the meta file is the commit point.  Data appended by an add that never
    committed (process killed between the steps) is cut off on reopen; data SHORTER than the commit raises."""
    base, docs = tmp_path / "rag_documents", tmp_path / "rag_documents.docs.jsonl"
    dim, rows = 4, 3
    lines = [(json.dumps({"id": f"p{i}", "text": f"t{i}", "metadata": {}}) + "\n").encode() for i in range(5)]
    committed_docs = b"".join(lines[:rows])

    def write(f16_rows, doc_lines, meta_rows, with_bytes=True):
        open(str(base) + ".f16", "wb").write(np.arange(f16_rows * dim, dtype=np.uint16).tobytes())
        open(docs, "wb").write(b"".join(doc_lines))
        extra = f"docs_bytes {len(committed_docs)}\n" if with_bytes else ""
        open(str(base) + ".meta", "w").write(f"rq-index 1\ndim {dim}\nrows {meta_rows}\ndtype f16\n{extra}")
    assert si.repair_persisted_collection(base, docs) is None                     # nothing committed yet
    write(5, lines, 3)                                                            # killed after appending rows AND records
    assert si.repair_persisted_collection(base, docs) == {"dim": 4, "rows": 3, "docs_bytes": len(committed_docs)}
    assert os.path.getsize(str(base) + ".f16") == 3 * dim * 2 and open(docs, "rb").read() == committed_docs
    write(5, lines[:3] + [lines[3][:10]], 3)                                      # killed in the middle of a record
    si.repair_persisted_collection(base, docs)
    assert open(docs, "rb").read() == committed_docs
    write(4, lines[:4], 3, with_bytes=False)                                      # a meta written by rq_save alone: first `rows` lines
    assert si.repair_persisted_collection(base, docs)["docs_bytes"] == len(committed_docs)
    assert open(docs, "rb").read() == committed_docs
    write(2, lines[:3], 3)                                                        # rows missing: real corruption
    with pytest.raises(RuntimeError, match="inconsistent"):
        si.repair_persisted_collection(base, docs)
    write(3, lines[:2], 3)                                                        # records missing
    with pytest.raises(RuntimeError, match="inconsistent"):
        si.repair_persisted_collection(base, docs)
    # a process that merely OPENS the collection (DenseIndex._load) reads the commit and leaves the bytes beyond it
    # alone -- they may belong to a writer that sits between appending its data and replacing the meta
    write(5, lines, 3)
    assert si.repair_persisted_collection(base, docs, truncate=False) == {"dim": 4, "rows": 3, "docs_bytes": len(committed_docs)}
    assert os.path.getsize(str(base) + ".f16") == 5 * dim * 2 and open(docs, "rb").read() == b"".join(lines)


def test_package_reexports_the_reference_names():
    """reference rag_uq/__init__.py:13,19-22: `from rag_uq import HybridRetriever, StreamingIndex`"""
    import rag_uq_amd
    from rag_uq_amd import HybridRetriever, StreamingIndex, Document
    assert HybridRetriever is si.HybridRetriever and StreamingIndex is si.StreamingIndex and Document is si.Document
    with pytest.raises(AttributeError):
        rag_uq_amd.RetrievalRouter                                               # the router stays in the reference


def test_bm25_numpy_posting_views_follow_appends_and_reloads(tmp_path):
    """BM25Index.get_scores keeps numpy views of the (append-only) posting lists between queries.  A search, an append that
    extends lists already viewed, and a reload from disk must give exactly what a freshly built index gives."""
    rng = np.random.default_rng(3)
    vocab = [f"w{i}" for i in range(300)]
    texts = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 40)))) for _ in range(600)]
    docs = [si.Document(id=f"d{i}", text=t) for i, t in enumerate(texts)]
    queries = [" ".join(rng.choice(vocab, size=6)) for _ in range(25)]
    inc = si.BM25Index(persist_path=str(tmp_path / "bm25.pkl"))
    inc.add_documents(docs[:350])
    first = [inc.search(q, 20) for q in queries]                     # views of the first 350 documents' lists exist now
    inc.add_documents(docs[350:])
    fresh = si.BM25Index()
    fresh.add_documents(docs)
    assert [inc.search(q, 20) for q in queries] == [fresh.search(q, 20) for q in queries]
    assert first != [fresh.search(q, 20) for q in queries]            # (the append did change the answers)
    for q in queries[:5]:
        assert np.array_equal(inc.get_scores(inc._tokenize(q)), fresh.get_scores(fresh._tokenize(q)))
    inc.close()
    again = si.BM25Index(persist_path=str(tmp_path / "bm25.pkl"))
    assert [again.search(q, 20) for q in queries] == [fresh.search(q, 20) for q in queries]


def test_fused_encoder_accepts_only_the_architecture_it_was_written_for():
    """embedders.FusedNomicBertForward inspects the module it wraps (no GPU needed for that): the stock NomicBert layout is
    accepted and its q/k/v and gate/up weights are concatenated; a projection with a bias, another head size or another
    activation make it decline, and NomicBertEmbedder then keeps the stock forward."""
    torch = pytest.importorskip("torch")
    from transformers.models.nomic_bert import NomicBertConfig, NomicBertModel
    from rag_uq_amd.embedders import FusedNomicBertForward
    cfg = NomicBertConfig(); cfg.num_hidden_layers = 2
    model = NomicBertModel(cfg).eval()
    f = FusedNomicBertForward(model)
    assert f.ok and len(f.layers) == 2 and f.theta == 1000.0 and f.heads == 12
    assert tuple(f.layers[0]["wqkv"].shape) == (3 * 768, 768) and tuple(f.layers[0]["wgu"].shape) == (2 * 3072, 768)
    a = model.layers[0].self_attn
    assert torch.equal(f.layers[0]["wqkv"][768:1536], a.k_proj.weight)
    ids = torch.zeros((2, 5), dtype=torch.long); mask = torch.ones((2, 5), dtype=torch.long)
    assert not f.usable(ids, mask)                                   # CPU tensors / fp32 weights: the stock forward
    a.q_proj.bias = torch.nn.Parameter(torch.zeros(768))
    assert not FusedNomicBertForward(model).ok
    cfg2 = NomicBertConfig(); cfg2.num_hidden_layers = 1; cfg2.hidden_act = "gelu"
    assert not FusedNomicBertForward(NomicBertModel(cfg2)).ok
    cfg3 = NomicBertConfig(); cfg3.num_hidden_layers = 1; cfg3.num_attention_heads = 6; cfg3.head_dim = 128
    assert not FusedNomicBertForward(NomicBertModel(cfg3)).ok


def test_bm25_log_survives_a_torn_tail_followed_by_more_adds(tmp_path):
    """a torn last line used to stay in <path>.log.jsonl; the next add was glued onto it and every document added from
    then on vanished at the following load (sparse and dense side of a HybridRetriever diverge).  Now the load cuts the log back to
    its last good record: torn tail -> reload -> add -> reload must hold every acknowledged document."""
    p = tmp_path / "bm25.pkl"
    mk = lambda lo, hi: [si.Document(id=f"p{i}", text=f"passage {i} topic {i % 5}") for i in range(lo, hi)]
    a = si.BM25Index(str(p))
    a.add_documents(mk(0, 5))
    if a._log_file is not None:
        a._log_file.close(); a._log_file = None            # (the process dies here: no close(), no snapshot)
    log = tmp_path / "bm25.pkl.log.jsonl"
    good = log.stat().st_size
    with open(log, "ab") as f:
        f.write(b'{"id": "torn", "text": "half a li')
    b = si.BM25Index(str(p))
    assert len(b) == 5 and log.stat().st_size == good       # the torn bytes are gone from the file
    b.add_documents(mk(5, 10))
    assert len(b) == 10
    if b._log_file is not None:
        b._log_file.close(); b._log_file = None
    c = si.BM25Index(str(p))
    assert len(c) == 10 and c.doc_ids == [f"p{i}" for i in range(10)]
    assert c.search("passage 7 topic 2", 3)[0][0] == "p7"
    # a record that lost only its newline is kept and gets the newline back, so the next append starts a fresh line
    raw = log.read_bytes()
    assert raw.endswith(b"\n")
    log.write_bytes(raw[:-1])
    d = si.BM25Index(str(p))
    assert len(d) == 10 and log.read_bytes() == raw
    d.add_documents(mk(10, 12))
    d._log_file.close(); d._log_file = None
    assert len(si.BM25Index(str(p))) == 12
    # invalid UTF-8 / a JSON value that is not a record also end the replay cleanly
    with open(log, "ab") as f:
        f.write(b'[1, 2]\n{"id": "after", "text": "x"}\n')
    e = si.BM25Index(str(p))
    assert len(e) == 12 and "after" not in e.documents


def test_bm25_batch_equals_per_query_equals_oracle_with_ties():
    """BM25Index.search_batch (librq_bm25.so: one pass over the posting lists on the host cores) == [search(q) for q] bit for
    bit (same float64 additions in query-token order, same tie rule) == the numpy batch path == oracle/bm25_oracle.py (scores to
    1e-12: the oracle sums per document, not per posting list).  The corpus is built to TIE: duplicate passages, equal lengths."""
    rng = np.random.default_rng(5)
    vocab = [f"w{i}" for i in range(60)]
    texts = [" ".join(rng.choice(vocab, size=8)) for _ in range(300)]
    texts[50:60] = [texts[7]] * 10                              # exact duplicates: exactly tied scores
    texts[100] = ""                                             # an empty passage
    ids = [f"p{i}" for i in range(len(texts))]
    b = si.BM25Index()
    b.add_documents([si.Document(id=i, text=t) for i, t in zip(ids, texts)])
    queries = [" ".join(rng.choice(vocab, size=5)) for _ in range(40)] + [texts[7], "w3 w3 w3 unknownword", "nothing known here", ""]
    b.SINGLE_NATIVE_AFTER = None                                # `search` on its own (numpy) path: the reference the batch forms are held against
    for k in (1, 10, 400):
        per_query = [b.search(q, k) for q in queries]
        assert b.search_batch(queries, k) == per_query
        assert b.search_batch(queries, k, use_native=False) == per_query
        assert b.search_batch(queries, k, n_threads=3) == per_query
        for q, got in zip(queries[:12] + queries[-4:], per_query[:12] + per_query[-4:]):
            want = bm25_oracle.bm25_search(ids, texts, q, k)
            assert [d for d, _ in got] == [d for d, _ in want]
            assert all(abs(a - c) <= 1e-12 for (_, a), (_, c) in zip(got, want))
    tied = b.search(texts[7], 11)
    assert [d for d, _ in tied][:11] == ["p59", "p58", "p57", "p56", "p55", "p54", "p53", "p52", "p51", "p50", "p7"]   # ties: descending row
    # the CSR arrays follow the corpus: an add invalidates them
    b.add_documents([si.Document(id="new", text=texts[7])])
    assert b.search_batch([texts[7]], 3) == [b.search(texts[7], 3)] and b.search(texts[7], 1)[0][0] == "new"
    assert si.BM25Index().search_batch(["x"], 3) == [[]]
    # one query per call (the reference's evaluation loop): after a few searches over an unchanged corpus `search` itself goes through the
    # batch scorer -- same lists -- and an add sends it back to its own path until the corpus has been stable again
    want = [b.search(q, 10) for q in queries]
    b.SINGLE_NATIVE_AFTER = 4
    b._drop_csr()
    for i, q in enumerate(queries[:4]):
        assert b.search(q, 10) == want[i] and "_csr_cache" not in b.__dict__
    got = [b.search(q, 10) for q in queries]
    assert got == want
    from rag_uq_amd import _native as nat_
    assert ("_csr_cache" in b.__dict__) == nat_.bm25_available()
    b.add_documents([si.Document(id="newer", text="w1 w2")])
    n_before = b.__dict__.get("_csr_cache", {}).get("n_docs")
    assert b.search("w1 w2", 1)[0][0] == "newer"
    assert b.__dict__.get("_csr_cache", {}).get("n_docs") == n_before          # (stale arrays were not rebuilt for one search)
    assert b.search("w1 w2", 0) == [] and b.search("", 5) == []


class _StubNative:
    """Host stand-in for _native.NativeIndex in CPU tests of the Python seam: brute-force inner product / cosine in float64 with the
    canonical order.  Test scaffolding (the product has no CPU backend); only what DenseIndex.from_native / search_rows_batch touch."""
    device, devices = 0, [0]

    def __init__(self, x):
        self.x = np.asarray(x, np.float64)
        self.dim = self.x.shape[1]

    def __len__(self):
        return self.x.shape[0]

    def search(self, q, k, metric=0):
        q = np.atleast_2d(np.asarray(q, np.float64))
        sc = q @ self.x.T
        if metric == 0:
            sc = sc / (np.linalg.norm(q, axis=1)[:, None] * np.linalg.norm(self.x, axis=1)[None, :] + 1e-30)
        sc = sc.astype(np.float32)
        rows = np.full((q.shape[0], k), -1, np.int64); out = np.zeros((q.shape[0], k), np.float32)
        for b in range(q.shape[0]):
            o = np.lexsort((np.arange(len(self)), -sc[b].astype(np.float64)))[:k]
            rows[b, :len(o)] = o; out[b, :len(o)] = sc[b, o]
        return out, rows


@pytest.mark.parametrize("metric", ["cosine", "ip"])
def test_batched_fusion_in_row_space_equals_the_per_query_path(tmp_path, metric):
    """HybridRetriever.get_scores_for_router_batch / hybrid_search_batch fuse the whole batch on integer keys (`_fuse_batch_rows`); the
    result must be the per-query path's, value for value and in the same order: documents in both pools, only in BM25, only in the dense
    index (not indexed by BM25), dense ids `self.documents` does not know (dropped, reference :491-493), questions without any BM25 hit,
    pools smaller than top_k (padding), exactly tied hybrid scores (stable order), and -- inner product with a negated query -- all-negative
    dense scores, which invert the order through `max(...) or 1` (reference :506-511)."""
    from rag_uq_amd.embedders import RandomProjectionEmbedder
    rng = np.random.default_rng(3)
    vocab = [f"w{i}" for i in range(40)]
    texts = [" ".join(rng.choice(vocab, size=6)) for _ in range(120)]
    texts[30:34] = [texts[5]] * 4                                          # duplicates: tied BM25 AND dense scores
    docs = [si.Document(id=f"p{i}", text=t, title=f"T{i}") for i, t in enumerate(texts)]
    emb = RandomProjectionEmbedder(16)

    class NegEmb:                                                           # every dense score negative under the inner product
        dim = 16
        def embed(self, ts): return -np.abs(emb.embed(ts))
    vec = emb.embed(texts)
    if metric == "ip":
        vec = np.abs(vec)
    extra_ids = [f"ghost{i}" for i in range(5)] + [f"donly{i}" for i in range(6)]      # dense-only rows: 5 unknown to the store, 6 known
    extra_vec = emb.embed([texts[i] + " w1" for i in range(11)])
    if metric == "ip":
        extra_vec = np.abs(extra_vec)
    dense = si.DenseIndex.from_native(_StubNative(np.concatenate([vec, extra_vec])), [d.id for d in docs] + extra_ids,
                                      embedder=NegEmb() if metric == "ip" else emb, metric=metric)
    r = si.HybridRetriever(bm25_persist_path=str(tmp_path / "b.pkl"), chroma_persist_path=str(tmp_path / "c"), dense_index=dense)
    r.bm25_index.add_documents(docs[:100])                                  # p100..p119 are not in BM25
    for d in docs:
        r.documents[d.id] = d
    for i in range(6):
        r.documents[f"donly{i}"] = si.Document(id=f"donly{i}", text=f"dense only {i}")
    queries = [" ".join(rng.choice(vocab, size=4)) for _ in range(30)] + [texts[5], "nothing known here", texts[110], ""]
    for num, pool in ((10, 50), (20, 7), (100, 100), (3, 1)):
        want = [r.get_scores_for_router(q, num_passages=num, retrieval_pool_size=pool) for q in queries]
        got = r.get_scores_for_router_batch(queries, num_passages=num, retrieval_pool_size=pool)
        assert r._fuse_batch_rows(queries, num, pool) is not None           # (the row-space path really ran)
        assert [tuple(g) for g in got] == [tuple(w) for w in want], (num, pool)
        assert r.hybrid_search_batch(queries, top_k=num, retrieval_pool_size=pool) == [r.hybrid_search(q, top_k=num, retrieval_pool_size=pool) for q in queries]
    flat = [i for q in queries for i in r.get_scores_for_router(q, 100, retrieval_pool_size=100)[2]]
    assert any(i.startswith("donly") for i in flat) and not any(i.startswith("ghost") for i in flat)
    # texts come from the store at call time: a document whose text was replaced answers with the new text on both paths
    r.documents["p5"] = si.Document(id="p5", text="replaced text", title="T5")
    assert r.get_scores_for_router_batch([texts[5]], 5, retrieval_pool_size=50) == [r.get_scores_for_router(texts[5], 5, retrieval_pool_size=50)]
    assert "replaced text" in r.get_scores_for_router_batch([texts[5]], 50, retrieval_pool_size=50)[0][3]
    # the key space follows the stores: a document added later is found by the batch path too
    r.bm25_index.add_documents([docs[100]])
    assert r.get_scores_for_router_batch([texts[100]], 5) == [r.get_scores_for_router(texts[100], 5)]


def test_committed_bench_line_has_the_contract_shape():
    """profiles/r03_bench20.json is a `python bench.py --steps 20 --warmup 5` line from the GPU box (tools/run_profiles.sh): the keys the
    driver and the review read, on BASELINE's basis -- fp16 rows as `value`, 1536 B per row in the roofline, the int8 image and the
    structured corpora as sibling objects -- and the internal consistency of the numbers (kernel <= step, achieved = bytes / time)."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_bench20.json")
    d = json.loads(open(path).read())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "int8_scan", "structured", "host_api", "ids_exact", "recall_at_10", "timed_path_vs_exact_fp64_scan"):
        assert key in d, key
    assert d["dtype"] == "f16" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["vs_baseline"] is None and d["scaling"] == "strong"
    assert "workload" in d["config"] and "1536 B per row" in d["config"]["workload"] and d["config"]["ranks_seen"] == 1 and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["algorithmic_bytes_per_launch"] == 1_000_000 * 1536
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"] and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["avg_launch_us"] <= d["ms_per_step"] * 1e3 * 1.001                     # the kernel fits in the step
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert 0.6 <= r["frac"] <= 1.0 and r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.1
    assert d["ids_exact"] and d["recall_at_10"] == 1.0 and d["timed_path_vs_exact_fp64_scan"]["ids_match"]
    i8 = d["int8_scan"]
    assert i8["dtype"] == "i8" and i8["extra_hbm_bytes_per_row"] == 768 and i8["roofline"]["algorithmic_bytes_per_launch"] == 1_000_000 * 768 and i8["ids_exact"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and d["value"] >= 10 * c["value"]            # north star: >= 10x the CPU path
    for corpus in ("documents/random", "documents/on-topic", "centroids/random", "centroids/on-topic"):
        for operand in ("fp16", "int8", "auto"):
            assert d["structured"][corpus][operand]["ids_match_exact_fp64_scan"], (corpus, operand)
    assert "python_seam" in d["host_api"] and "batch_1_k50" in d["host_api"]["python_seam"]
