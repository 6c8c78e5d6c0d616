"""CPU oracle for the dense-retrieval hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates
----------------
reference rag_uq/streaming_index.py:338-370 (DenseIndex.search): embed the query, ask the vector
store for the `top_k` nearest rows under cosine distance (collection created with
{"hnsw:space": "cosine"}, :260-263), return them best first with `score = 1.0 - distance`
(:363-368).  The store is ChromaDB `chromadb>=0.4.0` (requirements.txt:7), an un-vendored
third-party dependency that is absent from /root/reference and not installed here; its published
behaviour for the cosine space is distance = 1 - (q.x)/(|q||x|) over fp32 embeddings, searched
approximately with HNSW.  The reference holds no test, fixture or golden vector for this call
(no test imports rag_uq.streaming_index), therefore:

    PARITY UNPINNED upstream -- this oracle is the exact (brute force) restatement of that
    definition and is the parity target (BASELINE.json configs[0]: "CPU brute-force cosine top-10").

Definition fixed here (DESIGN.md "score definition"):
    corpus rows are the STORED fp16 values (the north star stores fp16 passage vectors)
    cosine: s = fp32( dot64(q, x) / (|q|_64 * |x|_64 + 1e-30) )      zero-norm row or query -> 0.0
            (matches the zero-vector failure path of streaming_index.py:281-284)
    ip:     s = fp32( dot64(q, x) )
    order:  s descending, then row ascending; k_eff = min(k, N); missing entries row = -1, score 0
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import numpy as np

METRIC_COSINE = 0
METRIC_IP = 1


def prepare_rows_f32(x: np.ndarray, normalize: bool = True) -> np.ndarray:
    """fp32 rows -> the fp16 rows the index stores (include/rq.h rq_index_add_f32).

    normalize: each row is divided by its fp64 L2 norm (zero rows stay zero), rounded to fp32, then
    to fp16 (round-to-nearest-even both times)."""
    x = np.asarray(x, dtype=np.float32)
    if not normalize:
        return x.astype(np.float16)
    x64 = x.astype(np.float64)
    nrm = np.sqrt((x64 * x64).sum(axis=1, keepdims=True))
    nrm[~(nrm > 0.0)] = 1.0
    return (x64 / nrm).astype(np.float32).astype(np.float16)


def exact_scores(q: np.ndarray, x16: np.ndarray, metric: int = METRIC_COSINE, chunk: int = 65536) -> np.ndarray:
    """Canonical fp32 scores [B][N] (fp64 arithmetic, one rounding to fp32)."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float32))
    x16 = np.asarray(x16)
    assert x16.dtype == np.float16 and x16.ndim == 2 and q.shape[1] == x16.shape[1]
    B, N = q.shape[0], x16.shape[0]
    q64 = q.astype(np.float64)
    qn = np.sqrt((q64 * q64).sum(axis=1))
    out = np.empty((B, N), dtype=np.float32)
    for lo in range(0, N, chunk):
        x64 = x16[lo:lo + chunk].astype(np.float64)
        dot = q64 @ x64.T
        if metric == METRIC_COSINE:
            xn = np.sqrt((x64 * x64).sum(axis=1))
            dot = dot / (qn[:, None] * xn[None, :] + 1e-30)
        out[:, lo:lo + chunk] = dot.astype(np.float32)
    return out


def topk_from_scores(scores: np.ndarray, k: int, row_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Canonical order over a score matrix: (score desc, row asc); pads with (0.0, -1)."""
    B, N = scores.shape
    ke = min(k, N)
    out_s = np.zeros((B, k), dtype=np.float32)
    out_r = np.full((B, k), -1, dtype=np.int64)
    rows = np.arange(N, dtype=np.int64)
    for b in range(B):
        s = scores[b]
        if ke < N:
            # every row tied with the ke-th score must take part in the tie-break
            kth = np.partition(s, N - ke)[N - ke]
            cand = np.nonzero(s >= kth)[0]
        else:
            cand = rows
        order = np.lexsort((cand, -s[cand].astype(np.float64)))[:ke]
        sel = cand[order]
        out_s[b, :ke] = s[sel]
        out_r[b, :ke] = sel + row_offset
    return out_s, out_r


def dense_topk(q: np.ndarray, x16: np.ndarray, k: int, metric: int = METRIC_COSINE, row_offset: int = 0
               ) -> Tuple[np.ndarray, np.ndarray]:
    """The oracle: exact top-k of each query over the stored fp16 rows."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float32))
    if x16.shape[0] == 0:
        return np.zeros((q.shape[0], k), np.float32), np.full((q.shape[0], k), -1, np.int64)
    return topk_from_scores(exact_scores(q, x16, metric), k, row_offset)


def merge_topk(parts: Sequence[Tuple[np.ndarray, np.ndarray]], k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge per-shard (scores, global rows) lists into the global canonical top-k."""
    s = np.concatenate([p[0] for p in parts], axis=1)
    r = np.concatenate([p[1] for p in parts], axis=1)
    B = s.shape[0]
    out_s = np.zeros((B, k), np.float32)
    out_r = np.full((B, k), -1, np.int64)
    for b in range(B):
        ok = r[b] >= 0
        sb, rb = s[b][ok], r[b][ok]
        order = np.lexsort((rb, -sb.astype(np.float64)))[:k]
        out_s[b, :len(order)] = sb[order]
        out_r[b, :len(order)] = rb[order]
    return out_s, out_r


# ---------------------------------------------------------------------------------------------
# fp32 BLAS variant: the CPU baseline that bench.py times.  Same result
# contract, fp32 arithmetic (scores within ~1e-6 of the canonical ones; ranks can differ on
# near-ties, which is why it is the *timed baseline* and not the parity oracle).
# ---------------------------------------------------------------------------------------------
class Fp32BruteForce:
    """Row-blocked, multi-threaded brute force: every worker thread owns a slice of corpus rows, runs a
    single-threaded OpenBLAS GEMM on it and keeps the slice's top-k; the slices are merged at the end.
    (numpy releases the GIL inside matmul / argpartition, so the threads really run in parallel.)"""

    def __init__(self, x16: np.ndarray, n_threads: int = 0, block_rows: int = 16384):
        import os
        self.x32 = np.ascontiguousarray(x16.astype(np.float32))
        self.inv = (1.0 / np.maximum(np.sqrt((self.x32.astype(np.float64) ** 2).sum(axis=1)), 1e-30)).astype(np.float32)
        self.n_threads = n_threads or len(os.sched_getaffinity(0))
        self.block_rows = block_rows

    def _block(self, q: np.ndarray, qn: np.ndarray, lo: int, hi: int, k: int):
        s = (q @ self.x32[lo:hi].T) * self.inv[None, lo:hi] / (qn[:, None] + 1e-30)
        ke = min(k, hi - lo)
        part = np.argpartition(-s, ke - 1, axis=1)[:, :ke]
        return np.take_along_axis(s, part, axis=1), part.astype(np.int64) + lo

    def search_blas(self, q: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """One multi-threaded OpenBLAS GEMM over the whole corpus, then numpy's (single-threaded) argpartition."""
        q = np.atleast_2d(np.asarray(q, dtype=np.float32))
        qn = np.sqrt((q.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
        s, r = self._block(q, qn, 0, self.x32.shape[0], k)
        o = np.argsort(r, axis=1, kind="stable")
        s, r = np.take_along_axis(s, o, axis=1), np.take_along_axis(r, o, axis=1)
        o = np.argsort(-s, axis=1, kind="stable")
        return np.take_along_axis(s, o, axis=1), np.take_along_axis(r, o, axis=1)

    def search(self, q: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        from concurrent.futures import ThreadPoolExecutor
        q = np.atleast_2d(np.asarray(q, dtype=np.float32))
        qn = np.sqrt((q.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
        n = self.x32.shape[0]
        spans = [(lo, min(lo + self.block_rows, n)) for lo in range(0, n, self.block_rows)]
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1)      # one BLAS thread per worker: the pool provides the parallelism
        except Exception:                            # pragma: no cover
            limit = None
        try:
            if self.n_threads > 1 and len(spans) > 1:
                with ThreadPoolExecutor(max_workers=self.n_threads) as ex:
                    parts = list(ex.map(lambda sp: self._block(q, qn, sp[0], sp[1], k), spans))
            else:
                parts = [self._block(q, qn, lo, hi, k) for lo, hi in spans]
        finally:
            if limit is not None:
                limit.restore_original_limits()
        s = np.concatenate([p[0] for p in parts], axis=1)
        r = np.concatenate([p[1] for p in parts], axis=1)
        ke = min(k, n)
        # rows ascending first, then a stable sort by score: ties end up ordered by row
        o = np.argsort(r, axis=1, kind="stable")
        s, r = np.take_along_axis(s, o, axis=1), np.take_along_axis(r, o, axis=1)
        o = np.argsort(-s, axis=1, kind="stable")[:, :ke]
        return np.take_along_axis(s, o, axis=1), np.take_along_axis(r, o, axis=1)


def recall_at_k(found_rows: np.ndarray, gold_rows: np.ndarray) -> float:
    """Set-overlap recall, reference rag_uq/eval_protocol.py:170-181 with gold = oracle top-k."""
    hits = 0
    total = 0
    for f, g in zip(found_rows, gold_rows):
        gs = set(int(v) for v in g if v >= 0)
        if not gs:
            continue
        hits += len(gs & set(int(v) for v in f if v >= 0))
        total += len(gs)
    return hits / max(total, 1)


def synthetic_corpus(n: int, dim: int = 768, seed: int = 1234, clustered: bool = False) -> np.ndarray:
    """SURVEY 8(d) synthetic inputs: Gaussian rows, unit norm, fp16 (clustered: 64 centroids + 0.3 noise)."""
    rng = np.random.default_rng(seed)
    if clustered:
        cent = rng.standard_normal((64, dim)).astype(np.float32)
        x = cent[rng.integers(0, 64, size=n)] + 0.3 * rng.standard_normal((n, dim)).astype(np.float32)
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    return prepare_rows_f32(x, normalize=True)


def synthetic_queries(n: int, dim: int = 768, seed: int = 4321) -> np.ndarray:
    return np.random.default_rng(seed).standard_normal((n, dim)).astype(np.float32)
