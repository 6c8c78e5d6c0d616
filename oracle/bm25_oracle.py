"""CPU oracle for the sparse side -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, as plain loops, what reference rag_uq/streaming_index.py:141-142,168-177 obtains from
`rank_bm25.BM25Okapi` (rank-bm25>=0.2.2, requirements.txt:8; third-party, not vendored, not installed
here -> PARITY UNPINNED against the real package; the reference holds no BM25 fixture).  Published
algorithm of rank-bm25 0.2.2 (BM25Okapi, k1=1.5, b=0.75, epsilon=0.25):

    nd[t]   = number of documents containing t
    idf[t]  = ln(N - nd[t] + 0.5) - ln(nd[t] + 0.5)
    average_idf = sum(idf) / len(idf);  every negative idf[t] is replaced by epsilon * average_idf
    score(d) = sum over query tokens t (with repetition):
               (idf.get(t) or 0) * f(t,d) * (k1 + 1) / (f(t,d) + k1 * (1 - b + b * len(d) / avgdl))

and the reference's selection on top of it (:172-177): np.argsort(scores)[::-1][:top_k], keep score > 0.
numpy's default argsort is not stable: the order of EXACTLY tied scores is unspecified (and differs between numpy builds /
SIMD paths).  The oracle fixes one admissible instance, the stable one: np.argsort(scores, kind="stable")[::-1], i.e. tied
documents by descending row.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np


def tokenize(text: str) -> List[str]:
    """reference :118-120"""
    return text.lower().split()


def bm25_scores(corpus_tokens: Sequence[Sequence[str]], query_tokens: Sequence[str], k1: float = 1.5, b: float = 0.75,
                epsilon: float = 0.25) -> np.ndarray:
    n = len(corpus_tokens)
    doc_len = [len(d) for d in corpus_tokens]
    avgdl = sum(doc_len) / n
    freqs: List[Dict[str, int]] = []
    nd: Dict[str, int] = {}
    for d in corpus_tokens:
        f: Dict[str, int] = {}
        for t in d:
            f[t] = f.get(t, 0) + 1
        freqs.append(f)
        for t in f:
            nd[t] = nd.get(t, 0) + 1
    idf: Dict[str, float] = {}
    idf_sum = 0.0
    neg = []
    for t, c in nd.items():
        v = math.log(n - c + 0.5) - math.log(c + 0.5)
        idf[t] = v
        idf_sum += v
        if v < 0:
            neg.append(t)
    eps = epsilon * (idf_sum / len(idf))
    for t in neg:
        idf[t] = eps
    scores = np.zeros(n)
    for t in query_tokens:
        w = idf.get(t) or 0
        for i in range(n):
            f = float(freqs[i].get(t, 0))
            scores[i] += w * (f * (k1 + 1) / (f + k1 * (1 - b + b * doc_len[i] / avgdl)))
    return scores


def bm25_search(doc_ids: Sequence[str], texts: Sequence[str], query: str, top_k: int) -> List[Tuple[str, float]]:
    scores = bm25_scores([tokenize(t) for t in texts], tokenize(query))
    top = np.argsort(scores, kind="stable")[::-1][:top_k]
    return [(doc_ids[i], float(scores[i])) for i in top if scores[i] > 0]
