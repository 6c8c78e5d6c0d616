"""BM25 batch scoring at scale on the host cores (no GPU): N passages of 40 words over a 50 000-word Zipf vocabulary, 500 questions, top-100:
BM25Index.search_batch (librq_bm25.so) against the per-query path.  usage: python tools/bm25_scale.py [N] [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rag_uq_amd
from rag_uq_amd import streaming_index as si
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(7)
V = 50_000
p = 1.0 / np.arange(1, V + 1) ** 0.9; p /= p.sum()
t0 = time.perf_counter()
words = rng.choice(V, size=(n, 40), p=p)
vocab = np.array([f"w{i}" for i in range(V)], dtype=object)
docs = [si.Document(id=f"p{i}", text=" ".join(vocab[row].tolist())) for i, row in enumerate(words)]
print(f"corpus of {n} passages generated in {time.perf_counter() - t0:.1f} s", flush=True)
bm = si.BM25Index()
t0 = time.perf_counter()
for lo in range(0, n, 50_000): bm.add_documents(docs[lo: lo + 50_000])
print(f"BM25Index.add_documents: {time.perf_counter() - t0:.1f} s", flush=True)
ans = rng.choice(n, size=500, replace=False)
questions = [" ".join(vocab[np.concatenate([rng.choice(words[a], 8, replace=False), rng.choice(V, 4, p=p)])].tolist()) for a in ans]
t0 = time.perf_counter(); bm._csr(); print(f"CSR arrays + contributions ({bm._csr()['indptr'][-1]} postings): {time.perf_counter() - t0:.1f} s", flush=True)
for thr in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 4, 16, 0]):
    t0 = time.perf_counter(); res = bm.search_batch(questions, 100, n_threads=thr); dt = time.perf_counter() - t0
    print(f"search_batch 500 questions top-100, n_threads={thr or 'auto'}: {dt * 1e3:.1f} ms ({500 / dt:.0f} questions/s)", flush=True)
bm.SINGLE_NATIVE_AFTER = None       # BM25Index.search on its own (numpy) path
t0 = time.perf_counter(); ref = [bm.search(q, 100) for q in questions[:40]]; dt = (time.perf_counter() - t0) / 40
bm.SINGLE_NATIVE_AFTER = 4          # the default: one-query searches over an unchanged corpus go through the batch scorer's arrays (round 3)
t0 = time.perf_counter(); ref1 = [bm.search(q, 100) for q in questions[:200]]; dt1 = (time.perf_counter() - t0) / 200
print(f"one query per call through the batch scorer: {dt1 * 1e3:.2f} ms per question; identical to the batch: {ref1 == res[:200]}", flush=True)
print(f"per-query search: {dt * 1e3:.1f} ms per question; identical to the batch: {ref == res[:40]}; answer in top-100: {np.mean([f'p{a}' in [d for d, _ in r] for a, r in zip(ans, res)]):.3f}", flush=True)
