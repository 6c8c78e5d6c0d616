/* rq_bm25.h -- C ABI of librq_bm25.so: batched BM25 scoring + top-k selection on the HOST cores.
 *
 * Replaces, for a whole batch of queries at once, what the reference obtains one query at a time from
 * rank_bm25.BM25Okapi.get_scores + np.argsort (reference rag_uq/streaming_index.py:168-177, BM25Index.search).
 * BASELINE.json configs[4] keeps the sparse side on the CPU ("GPU dense top-100 + CPU BM25 top-100"); this library
 * holds no GPU code and is built with g++ (csrc/rq_bm25.cpp).  The per-posting contributions
 *     idf(t) * f * (k1 + 1) / (f + k1 * (1 - b + b * dl / avgdl))
 * are computed by the Python side (BM25Index, float64, the same expression its per-query path evaluates); this
 * library only ADDS them, in query-token order, so every score has the same bits as the per-query path, and selects.
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, no global state, thread safe; returns 0 or a
 * negative error code (RQ_BM25_EINVAL).  Called from Python through ctypes (the GIL is released during the call).
 */
#ifndef RQ_BM25_H
#define RQ_BM25_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RQ_BM25_OK 0
#define RQ_BM25_EINVAL (-1)

typedef struct rq_bm25 rq_bm25;

/* Inverted index in CSR form: token t owns postings [indptr[t], indptr[t+1]): rows[] = document rows (each at most once
 * per token), contrib[] = that posting's score contribution.  The arrays are BORROWED: the caller keeps them alive and
 * unchanged until rq_bm25_destroy (BM25Index rebuilds them, and the handle, after an add -- idf and avgdl change with every
 * document).  Operands are validated here, once (a bad row must not become a wild write later).  NULL + no handle on error. */
rq_bm25* rq_bm25_create(const int64_t* indptr, const int32_t* rows, const double* contrib, int64_t n_tokens, int64_t n_docs);
void rq_bm25_destroy(rq_bm25* h);

/* Queries: query q owns q_tokens[q_indptr[q] .. q_indptr[q+1]), token ids in query order WITH repetition (tokens unknown
 * to the index are left out by the caller).
 *
 * score(q, d) = sum of contrib over the query's tokens, added in query order (float64).
 * Selection = reference :172-177 with a deterministic tie rule: documents with score > 0 only, best first, equal scores
 * ordered by DESCENDING row (what np.argsort(scores, kind="stable")[::-1] yields; the reference's default argsort leaves
 * the order of exact ties unspecified), at most k per query; out_rows is -1 padded, out_scores 0 padded.
 * n_threads <= 0: one thread per host core, at most 16.  Thread safe (no state in the handle changes). */
int rq_bm25_topk(const rq_bm25* h, const int64_t* q_indptr, const int32_t* q_tokens, int n_queries, int k,
                 int32_t* out_rows, double* out_scores, int n_threads);

const char* rq_bm25_version(void);

#ifdef __cplusplus
}
#endif
#endif
