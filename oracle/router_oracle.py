"""CPU restatement of the reference's RetrievalRouter forward pass -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/ (tests/config_workloads.py is what `bench.py --workload config4` runs) may import this module.

The router is the CONSUMER of the hot path (out of scope, "must run unchanged on its outputs"):
BASELINE.json configs[4] feeds GPU dense top-100 + CPU BM25 top-100 through it and compares Recall@10.  The reference's
own class needs /root/reference, which does not exist on the GPU box, so the config-4 harness uses this numpy twin.
It is pinned by tests/golden/g2_router.json, which tests/golden/make_golden.py captured from the reference itself
(seeded state dict, inputs, outputs).

What it restates
----------------
reference rag_uq/router.py:
  :74-93    scorer = Linear(3, hidden) -> ReLU -> Dropout -> Linear(hidden, 1) -> Sigmoid   (RouterConfig defaults :34-41;
            Dropout is the identity in eval mode)
  :104-142  _normalize_scores: running statistics when `stats_initialized`, else batch-wise (x - mean) / (std + 1e-6) with
            torch's unbiased std over the WHOLE tensor; `stats_initialized` is a plain attribute, not part of the state
            dict, so a router restored from a checkpoint (:499-517) normalises batch-wise
  :144-177  forward: features [bm25_norm, dense_norm, dense_norm - bm25_norm]
  :179-202  hybrid_rerank: weights * dense + (1 - weights) * bm25, torch.topk
experiments/run_evaluation.py:170-184: the evaluation loop's own rerank (argsort of the same hybrid score, descending).
All arithmetic in float32, like torch's.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np

_EPS = np.float32(1e-6)


class RouterOracle:
    def __init__(self, state_dict: Dict[str, Sequence]):
        f32 = lambda k: np.asarray(state_dict[k], dtype=np.float32)
        self.w0, self.b0 = f32("scorer.0.weight"), f32("scorer.0.bias")          # [hidden, 3], [hidden]
        self.w1, self.b1 = f32("scorer.3.weight"), f32("scorer.3.bias")          # [1, hidden], [1]
        self.bm25_mean, self.bm25_std = f32("bm25_mean"), f32("bm25_std")
        self.dense_mean, self.dense_std = f32("dense_mean"), f32("dense_std")
        self.stats_initialized = False

    def set_stats(self, bm25_mean: float, bm25_std: float, dense_mean: float, dense_std: float) -> None:
        self.bm25_mean, self.bm25_std = np.float32(bm25_mean), np.float32(bm25_std)
        self.dense_mean, self.dense_std = np.float32(dense_mean), np.float32(dense_std)
        self.stats_initialized = True

    def _normalize(self, bm25: np.ndarray, dense: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        if self.stats_initialized:
            return (bm25 - self.bm25_mean) / (self.bm25_std + _EPS), (dense - self.dense_mean) / (self.dense_std + _EPS)
        std = lambda a: np.float32(a.astype(np.float64).std(ddof=1)) if a.size > 1 else np.float32(np.nan)
        mean = lambda a: np.float32(a.astype(np.float64).mean())
        return (bm25 - mean(bm25)) / (std(bm25) + _EPS), (dense - mean(dense)) / (std(dense) + _EPS)

    def forward(self, bm25_scores, dense_scores) -> np.ndarray:
        """[batch, passages] x 2 -> gating weights [batch, passages] (0 favours BM25, 1 favours dense)."""
        bm25 = np.asarray(bm25_scores, dtype=np.float32)
        dense = np.asarray(dense_scores, dtype=np.float32)
        b, d = self._normalize(bm25, dense)
        feats = np.stack([b, d, d - b], axis=-1).reshape(-1, 3).astype(np.float32)
        h = np.maximum(feats @ self.w0.T + self.b0, np.float32(0.0))
        z = (h @ self.w1.T + self.b1).astype(np.float32)
        return (np.float32(1.0) / (np.float32(1.0) + np.exp(-z))).reshape(bm25.shape).astype(np.float32)

    def hybrid_scores(self, bm25_scores, dense_scores) -> np.ndarray:
        bm25 = np.asarray(bm25_scores, dtype=np.float32)
        dense = np.asarray(dense_scores, dtype=np.float32)
        w = self.forward(bm25, dense)
        return (w * dense + (np.float32(1.0) - w) * bm25).astype(np.float32)

    def hybrid_rerank(self, bm25_scores, dense_scores, top_k: int = 10) -> Tuple[np.ndarray, np.ndarray]:
        """(top-k hybrid scores, their passage indices), best first (router.py:179-202)."""
        hyb = self.hybrid_scores(bm25_scores, dense_scores)
        k = min(int(top_k), hyb.shape[-1])
        order = np.argsort(-hyb, axis=-1, kind="stable")[:, :k]
        return np.take_along_axis(hyb, order, axis=-1), order

    def eval_loop_order(self, bm25_scores: Sequence[float], dense_scores: Sequence[float]) -> list:
        """experiments/run_evaluation.py:170-184 for one question: passage order by hybrid score, descending."""
        hyb = self.hybrid_scores(np.asarray([bm25_scores]), np.asarray([dense_scores]))[0]
        return np.argsort(-hyb, kind="stable").tolist()


def recall_at_k(retrieved: Sequence[str], relevant: Sequence[str], k: int) -> float:
    """reference rag_uq/eval_protocol.py:170-181 (_recall_at_k), pinned by tests/golden/g3_retrieval_metrics.json."""
    if not relevant:
        return 0.0
    return len(set(retrieved[:k]) & set(relevant)) / len(set(relevant))
