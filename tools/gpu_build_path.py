"""Index build path at scale: StreamingIndex over a synthetic passage JSONL, HashEmbedder (the reference's
own fallback embedding), BM25 + dense + document store + checkpoint, batches of 100 as in the reference.  Prints the
time of every 20 000-passage segment: linear build = flat segment times.
usage: python tools/gpu_build_path.py [n_passages=200000]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rag_uq_amd
from rag_uq_amd import streaming_index as si
from rag_uq_amd.embedders import HashEmbedder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(5)
path = os.path.join(tmp, "passages.jsonl")
with open(path, "w") as f:
    for i in range(n):
        words = " ".join(f"w{w}" for w in rng.integers(0, 20_000, size=60))
        f.write(json.dumps({"id": f"p{i}", "text": words, "title": f"T{i % 977}", "metadata": {"chunk": i % 5}}) + "\n")
r = si.HybridRetriever(bm25_persist_path=os.path.join(tmp, "bm25.pkl"), chroma_persist_path=os.path.join(tmp, "chroma"), embedder=HashEmbedder())
s = si.StreamingIndex(r, checkpoint_path=os.path.join(tmp, "ckpt.json"), batch_size=100)
t0 = time.perf_counter(); seg_t0 = t0; done = 0; seg = 20_000
for added in s.stream_from_jsonl(path):
    done += added
    if done % seg == 0:
        now = time.perf_counter()
        print(f"passages {done - seg:7d}..{done:7d}: {now - seg_t0:6.2f} s  ({seg / (now - seg_t0):8.0f} passages/s)", flush=True)
        seg_t0 = now
r.close()
total = time.perf_counter() - t0
files = {f: os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs if f != "passages.jsonl"}
print(f"total {total:.1f} s for {n} passages ({n / total:.0f} passages/s); files: " + ", ".join(f"{k} {v >> 20} MiB" for k, v in sorted(files.items())), flush=True)
t0 = time.perf_counter()
r2 = si.HybridRetriever(bm25_persist_path=os.path.join(tmp, "bm25.pkl"), chroma_persist_path=os.path.join(tmp, "chroma"), embedder=HashEmbedder())
print(f"fresh-process reload: {time.perf_counter() - t0:.1f} s, {len(r2)} documents, dense {len(r2.dense_index)}, bm25 {len(r2.bm25_index)}", flush=True)
q = "w17 w4242 w19999"
assert [x.doc_id for x in r2.hybrid_search(q, 5)] == [x.doc_id for x in r.hybrid_search(q, 5)]
print("reload answers the same hybrid query identically", flush=True)
