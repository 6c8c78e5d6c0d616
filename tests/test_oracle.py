"""Oracle checks (CPU): golden pins, definition properties, host key packing."""
import json
import os

import numpy as np
import pytest

from oracle import dense_oracle as orc
from oracle import bm25_oracle


def test_oracle_matches_golden_pin(golden_dir):
    for case in json.load(open(os.path.join(golden_dir, "g4_oracle_dense.json"))):
        x16 = orc.synthetic_corpus(case["n"], case["dim"], seed=case["seed"])
        q = orc.synthetic_queries(case["B"], case["dim"], seed=case["seed"] + 1)
        s, r = orc.dense_topk(q, x16, case["k"])
        assert r.tolist() == case["rows"]
        assert np.array_equal(s, np.asarray(case["scores"], dtype=np.float32))


def test_oracle_against_plain_python_definition():
    """score = dot64/(|q||x| + 1e-30) on the stored fp16 values, order (score desc, row asc)."""
    rng = np.random.default_rng(0)
    x16 = rng.standard_normal((23, 9)).astype(np.float16)
    q = rng.standard_normal((3, 9)).astype(np.float32)
    s, r = orc.dense_topk(q, x16, 5)
    for b in range(3):
        ref = []
        for i in range(23):
            xs = [float(v) for v in x16[i]]
            qs = [float(v) for v in q[b]]
            dot = sum(a * c for a, c in zip(qs, xs))
            den = (sum(a * a for a in qs) ** 0.5) * (sum(c * c for c in xs) ** 0.5) + 1e-30
            ref.append((-float(np.float32(dot / den)), i))
        ref.sort()
        assert [i for _, i in ref[:5]] == r[b].tolist()
        np.testing.assert_allclose([-v for v, _ in ref[:5]], s[b], rtol=0, atol=1e-7)


def test_oracle_ties_zero_vectors_and_padding():
    x16 = np.zeros((6, 4), np.float16)
    x16[1] = x16[4] = [1, 0, 0, 0]
    x16[2] = [0.5, 0.5, 0, 0]
    q = np.array([[1, 0, 0, 0], [0, 0, 0, 0]], np.float32)
    s, r = orc.dense_topk(q, x16, 8)
    assert r[0].tolist() == [1, 4, 2, 0, 3, 5, -1, -1]          # ties by row, zero rows score 0, -1 padding
    assert s[0, 0] == 1.0 and s[0, 3] == 0.0
    assert r[1].tolist() == [0, 1, 2, 3, 4, 5, -1, -1] and not s[1].any()   # zero query: everything 0
    s0, r0 = orc.dense_topk(q, x16[:0], 3)
    assert (r0 == -1).all() and not s0.any()


def test_merge_of_shards_equals_global():
    x16 = orc.synthetic_corpus(500, 64, seed=3)
    q = orc.synthetic_queries(5, 64, seed=4)
    gs, gr = orc.dense_topk(q, x16, 10)
    parts = [orc.dense_topk(q, x16[lo:hi], 10, row_offset=lo) for lo, hi in [(0, 130), (130, 131), (131, 500)]]
    ms, mr = orc.merge_topk(parts, 10)
    assert np.array_equal(mr, gr) and np.array_equal(ms, gs)


def test_fp32_baseline_agrees_on_generic_data():
    x16 = orc.synthetic_corpus(4000, 128, seed=11)
    q = orc.synthetic_queries(8, 128, seed=12)
    gs, gr = orc.dense_topk(q, x16, 10)
    bs, br = orc.Fp32BruteForce(x16).search(q, 10)
    assert orc.recall_at_k(br, gr) == 1.0
    np.testing.assert_allclose(bs, gs, atol=2e-6)


def test_recall_metric_matches_reference(golden_dir):
    """reference rag_uq/eval_protocol.py:170-181"""
    for c in json.load(open(os.path.join(golden_dir, "g3_retrieval_metrics.json"))):
        if not c["relevant"]:
            continue
        ids = {d: i for i, d in enumerate(sorted(set(c["retrieved"]) | set(c["relevant"])))}
        found = np.array([[ids[d] for d in c["retrieved"][: c["k"]]] + [-1] * 16], dtype=np.int64)[:, :16]
        gold = np.array([[ids[d] for d in c["relevant"]] + [-1] * 16], dtype=np.int64)[:, :16]
        assert orc.recall_at_k(found, gold) == pytest.approx(c["recall_at_k"])


def test_prepare_rows_is_unit_norm_fp16():
    x = np.random.default_rng(5).standard_normal((50, 768)).astype(np.float32) * 7
    x[3] = 0
    h = orc.prepare_rows_f32(x, True)
    assert h.dtype == np.float16 and not h[3].any()
    n = np.linalg.norm(h.astype(np.float64), axis=1)
    assert np.all(np.abs(np.delete(n, 3) - 1) < 2e-3)


def test_bm25_oracle_basics():
    ids = ["a", "b", "c", "d"]
    texts = ["the cat sat", "the dog sat down", "a bird flew", "cat and dog and cat"]
    res = bm25_oracle.bm25_search(ids, texts, "bird", 10)
    assert [d for d, _ in res] == ["c"] and res[0][1] > 0
    # "cat" is in 2 of 4 documents: idf = ln(2.5) - ln(2.5) = 0 -> score 0 -> filtered by `score > 0`
    assert bm25_oracle.bm25_search(ids, texts, "cat", 10) == []
    assert bm25_oracle.bm25_search(ids, texts, "zebra", 10) == []
    # "the" and "sat" are in 2 of 4 too; "down" only in b
    assert [d for d, _ in bm25_oracle.bm25_search(ids, texts, "the dog sat down", 10)] == ["b"]


def test_router_restatement_matches_reference_g2(golden_dir):
    """oracle/router_oracle.py against the reference's RetrievalRouter (rag_uq/router.py:44-202) as captured in
    g2_router.json: gating weights, hybrid_rerank and the evaluation loop's order, with batch-wise and with
    running-statistics normalisation."""
    from oracle import router_oracle as ro
    g2 = json.load(open(os.path.join(golden_dir, "g2_router.json")))
    assert g2["config"] == dict(hidden_dim=64, dropout=0.1, num_layers=2, use_batch_norm=False)
    for c in g2["cases"]:
        r = ro.RouterOracle(g2["state_dict"])
        if c["stats_initialized"]:
            r.set_stats(**c["stats"])
        w = r.forward(c["bm25"], c["dense"])
        np.testing.assert_allclose(w, np.asarray(c["weights"], np.float32), rtol=0, atol=2e-6)
        s, i = r.hybrid_rerank(c["bm25"], c["dense"], c["top_k"])
        np.testing.assert_allclose(s, np.asarray(c["rerank_scores"], np.float32), rtol=2e-6, atol=2e-6)
        hyb = r.hybrid_scores(c["bm25"], c["dense"])

        def same_order(mine, ref, b):
            # identical up to the order INSIDE a group of exactly tied hybrid scores (the zero padding of
            # get_scores_for_router ties at 0.0; torch.topk / argsort leave that order unspecified)
            return len(mine) == len(ref) and all(m == t or hyb[b][m] == hyb[b][t] for m, t in zip(mine, ref))
        for b in range(c["shape"][0]):
            assert same_order(i[b].tolist(), c["rerank_indices"][b], b)
            if c["shape"][0] == 1:      # the loop of run_evaluation.py:170-184 normalises one question at a time
                assert same_order(r.eval_loop_order(c["bm25"][b], c["dense"][b]), c["eval_loop_order"][b], b)
    for c in json.load(open(os.path.join(golden_dir, "g3_retrieval_metrics.json"))):
        assert ro.recall_at_k(c["retrieved"], c["relevant"], c["k"]) == pytest.approx(c["recall_at_k"])
