"""BASELINE.json configs[2], [3], [4] at their shapes on one MI355X (tests/config_workloads.py).  Wikipedia embeddings,
nomic-embed-text weights and NQ-dev-500 do not exist offline: the workloads are synthetic stand-ins of the same shape
(the result dicts say so under "data") -- what is asserted is parity of the GPU path with the oracle, not retrieval
quality on real data."""
import pytest

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config_workloads as cw  # noqa: E402

pytestmark = pytest.mark.gpu


def test_config2_ten_million_rows_in_eight_shards():
    """10M x 768 fp16 in 8 row shards of 1.25M (15.4 GB, one GPU): set_row_offset, search_device with keys,
    rq_merge_keys_device.  Planted copies of stored rows (first / last row of a shard, both sides of shard boundaries)
    come back at rank 1, top-10 is a prefix of top-50, the merged answer equals one 10M-row index for all 64 queries,
    and a 3-query slice equals the oracle per shard + orc.merge_topk."""
    r = cw.run_config2(steps=4)
    assert r["planted_at_rank_1"] and r["top10_is_prefix_of_top50"], r
    assert r["merged_equals_single_index"], r
    assert r["ids_exact_vs_oracle"] and r["max_abs_score_err_vs_oracle"] <= 1e-6, r
    assert r["repaired_queries"] == 0, r


def test_config3_text_queries_through_nomic_bert_over_one_million_rows():
    """256 raw text queries -> 12-layer NomicBert forward (random init, fp16) -> search_batch over 1M rows: the rows
    returned for the vectors the encoder produced equal the oracle's for all 256 queries."""
    r = cw.run_config3(reps=2)
    assert r["ids_exact_vs_oracle"] and r["max_abs_score_err_vs_oracle"] <= 1e-6, r
    assert r["oracle_queries"] == 256


def test_config4_hybrid_top100_through_router_recall_at_10():
    """500 questions: GPU dense top-100 + CPU BM25 top-100 -> fusion -> router (oracle/router_oracle.py, pinned by
    g2_router.json from the reference) -> Recall@10 (definition pinned by g3).  With the dense side answered by the
    oracle on the same stored vectors the router inputs, the reranked id lists and Recall@10 are identical."""
    r = cw.run_config4()
    assert r["router_inputs_identical"] and r["id_lists_identical"], r
    assert r["recall_at_10_router_gpu_dense"] == r["recall_at_10_router_oracle_dense"]
    # The stand-in task is solvable -- these are sanity bounds on the SYNTHETIC data, not parity assertions (parity is the three
    # lines above).  Dense-only Recall@10 is 0.876 on this seed, deterministically: a 12-word question shares 8 words with its
    # 40-word passage and carries 4 distractors, and RandomProjectionEmbedder is a bag of random word directions, so 12 % of the
    # answers rank below 10 neighbours that happen to share frequent words.  (Round 2 first wrote "> 0.9" from a guess, saw 0.876 in
    # gpurun_out/r02_t5.log and dropped the bound to 0.5; 0.85 is the measured value with a margin for a different numpy RNG stream.)
    assert r["recall_at_10_dense_only"] > 0.85 and r["recall_at_10_fusion_only"] > 0.9, r
