"""Round 3: index build with the encoder in the loop (reference rows a2 / a4: DenseIndex.add_documents = embed + append, batches of 100, streaming_index.py:290-336).
NomicBert architecture with random weights + byte-level stand-in tokenizer (no nomic weights offline): ragged passages of 40-400 'tokens'."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_uq_amd  # noqa: F401
from rag_uq_amd import streaming_index as si
from rag_uq_amd.embedders import NomicBertEmbedder

torch.manual_seed(0)
emb = NomicBertEmbedder(random_init=True, num_layers=12, device="cuda:0", dtype="float16", batch_size=256)
rng = np.random.default_rng(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
lens = rng.integers(40, 401, size=n)
docs = [si.Document(id=f"p{i}", text=f"doc{i} " + "".join(chr(97 + (i * 7 + j * 3) % 26) if j % 6 else " " for j in range(int(L))), title=f"t{i}") for i, L in enumerate(lens)]
for bs in (100, 1000):
    tmp = tempfile.mkdtemp()
    r = si.HybridRetriever(bm25_persist_path=os.path.join(tmp, "bm25.pkl"), chroma_persist_path=os.path.join(tmp, "c"),
                           dense_index=si.DenseIndex(persist_directory=os.path.join(tmp, "c"), embedder=emb))
    r.add_documents(docs[:bs], batch_size=bs)                       # warm-up (workspaces, first launches)
    t_emb = 0.0
    real = emb.embed
    def timed_embed(texts):
        global t_emb
        t0 = time.perf_counter(); out = real(texts); t_emb += time.perf_counter() - t0
        return out
    emb.embed = timed_embed
    t0 = time.perf_counter()
    for lo in range(bs, n, bs):
        r.add_documents(docs[lo: lo + bs], batch_size=bs)
    dt = time.perf_counter() - t0
    emb.embed = real
    m = n - bs
    print(f"batches of {bs:4d}: {m} passages in {dt:.2f} s = {m / dt:7.0f} passages/s  (embedding {t_emb:.2f} s = {m / t_emb:7.0f} passages/s; BM25 + append + persistence {dt - t_emb:.2f} s)", flush=True)
    hit = r.hybrid_search(docs[777].text, top_k=3)
    print("   query = passage 777 ->", [x.doc_id for x in hit][:3], "dense rows", len(r.dense_index), flush=True)
    assert hit[0].doc_id == "p777"
