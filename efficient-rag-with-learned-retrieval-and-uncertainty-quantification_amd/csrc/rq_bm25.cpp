// rq_bm25.cpp -- librq_bm25.so (include/rq_bm25.h): batched BM25 scoring and top-k selection on the host cores.
// The sparse side of reference rag_uq/streaming_index.py:168-177 for a whole batch of queries (BASELINE.json configs[4] keeps
// BM25 on the CPU).  Term-at-a-time: a thread owns one dense float64 accumulator over the documents, adds the posting
// contributions of a query's tokens in query order (same additions, same order, same bits as BM25Index.get_scores), then
// walks the same postings once more to pick the k best and put the accumulator back to zero.  No GPU code in here.
#include "../../include/rq_bm25.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace {

struct Cand {
    double score;
    int32_t row;
};
// "a ranks before b": higher score first, equal scores by DESCENDING row (np.argsort(kind="stable")[::-1])
inline bool before(const Cand& a, const Cand& b) { return a.score > b.score || (a.score == b.score && a.row > b.row); }
// heap with the WORST kept candidate on top
struct WorstOnTop {
    bool operator()(const Cand& a, const Cand& b) const { return before(a, b); }
};

void worker(const int64_t* indptr, const int32_t* rows, const double* contrib, int64_t n_docs, const int64_t* q_indptr,
            const int32_t* q_tokens, int n_queries, int k, int32_t* out_rows, double* out_scores, std::atomic<int>* next) {
    std::vector<double> acc((size_t)n_docs, 0.0);
    std::vector<Cand> heap;
    heap.reserve((size_t)k + 1);
    for (;;) {
        const int q = next->fetch_add(1, std::memory_order_relaxed);
        if (q >= n_queries) break;
        const int64_t t0 = q_indptr[q], t1 = q_indptr[q + 1];
        for (int64_t t = t0; t < t1; ++t) {
            const int32_t tok = q_tokens[t];
            for (int64_t p = indptr[tok], e = indptr[tok + 1]; p < e; ++p) acc[(size_t)rows[p]] += contrib[p];
        }
        heap.clear();
        for (int64_t t = t0; t < t1; ++t) {
            const int32_t tok = q_tokens[t];
            for (int64_t p = indptr[tok], e = indptr[tok + 1]; p < e; ++p) {
                const int32_t d = rows[p];
                const double s = acc[(size_t)d];
                // a document reached a second time (another token, a repeated token) reads 0 (or, when every contribution
                // cancelled exactly, was never a candidate): it was taken -- and reset -- the first time
                if (s == 0.0) continue;
                acc[(size_t)d] = 0.0;
                if (!(s > 0.0)) continue;               // reference :177 keeps score > 0 only (NaN fails the test too)
                const Cand c{s, d};
                if ((int)heap.size() < k) {
                    heap.push_back(c);
                    std::push_heap(heap.begin(), heap.end(), WorstOnTop());
                } else if (before(c, heap.front())) {
                    std::pop_heap(heap.begin(), heap.end(), WorstOnTop());
                    heap.back() = c;
                    std::push_heap(heap.begin(), heap.end(), WorstOnTop());
                }
            }
        }
        std::sort(heap.begin(), heap.end(), before);
        int32_t* orow = out_rows + (size_t)q * k;
        double* osc = out_scores + (size_t)q * k;
        const int have = (int)heap.size();
        for (int i = 0; i < k; ++i) {
            orow[i] = i < have ? heap[(size_t)i].row : -1;
            osc[i] = i < have ? heap[(size_t)i].score : 0.0;
        }
    }
}

}  // namespace

extern "C" const char* rq_bm25_version(void) { return "rq-bm25 0.1 (host)"; }

struct rq_bm25 {
    const int64_t* indptr;
    const int32_t* rows;
    const double* contrib;
    int64_t n_tokens, n_docs;
};

extern "C" rq_bm25* rq_bm25_create(const int64_t* indptr, const int32_t* rows, const double* contrib, int64_t n_tokens, int64_t n_docs) {
    if (!indptr || n_tokens < 0 || n_docs < 0 || indptr[0] != 0) return nullptr;
    for (int64_t t = 0; t < n_tokens; ++t)
        if (indptr[t] > indptr[t + 1]) return nullptr;
    const int64_t nnz = indptr[n_tokens];
    if (nnz > 0 && (!rows || !contrib)) return nullptr;
    for (int64_t p = 0; p < nnz; ++p)
        if (rows[p] < 0 || (int64_t)rows[p] >= n_docs) return nullptr;
    return new rq_bm25{indptr, rows, contrib, n_tokens, n_docs};
}

extern "C" void rq_bm25_destroy(rq_bm25* h) { delete h; }

extern "C" int rq_bm25_topk(const rq_bm25* h, const int64_t* q_indptr, const int32_t* q_tokens, int n_queries, int k, int32_t* out_rows,
                            double* out_scores, int n_threads) {
    if (!h || !q_indptr || !out_rows || !out_scores || n_queries < 0 || k < 1) return RQ_BM25_EINVAL;
    if (n_queries == 0) return RQ_BM25_OK;
    if (q_indptr[0] < 0 || (q_indptr[n_queries] > 0 && !q_tokens)) return RQ_BM25_EINVAL;
    for (int q = 0; q < n_queries; ++q)
        if (q_indptr[q] > q_indptr[q + 1]) return RQ_BM25_EINVAL;
    for (int64_t t = q_indptr[0], e = q_indptr[n_queries]; t < e; ++t)
        if (q_tokens[t] < 0 || (int64_t)q_tokens[t] >= h->n_tokens) return RQ_BM25_EINVAL;
    int nthr = n_threads > 0 ? n_threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    nthr = std::max(1, std::min(nthr, n_queries));
    std::atomic<int> next(0);
    if (nthr == 1) {
        worker(h->indptr, h->rows, h->contrib, h->n_docs, q_indptr, q_tokens, n_queries, k, out_rows, out_scores, &next);
        return RQ_BM25_OK;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)nthr);
    for (int i = 0; i < nthr; ++i)
        th.emplace_back(worker, h->indptr, h->rows, h->contrib, h->n_docs, q_indptr, q_tokens, n_queries, k, out_rows, out_scores, &next);
    for (auto& t : th) t.join();
    return RQ_BM25_OK;
}
