// rq_multi.hip -- the multi-device parent index (include/rq.h rq_index_create with n_devices > 1): row shards per device
// inside the library.  The reference has no counterpart (its dense store is one ChromaDB collection,
// rag_uq/streaming_index.py:252-263); this is the north star's row sharding, for callers that want several GPUs behind
// ONE index handle instead of one process per GPU (rag_uq_amd.distributed).
//
// Layout: global rows are dealt to the devices in STRIPES of `stripe` rows (default 65 536 = 100 MB of fp16 rows), round
// robin: global row g lives in stripe s = g / stripe, on device slot s % G, at local row (s / G) * stripe + g % stripe.
// Appending in global order fills one stripe after the other, so every child's local rows stay dense, local order is
// monotone in the global id (ties keep their canonical order through the merge), and the mapping is arithmetic -- no
// segment table.  A streaming build that appends 100 documents per call (reference StreamingIndex, :606-660) touches ONE
// device per append (two when the block crosses a stripe end); round 2 cut every appended block into G pieces (12-row
// pieces and one table entry per append and device).
#include "rq_index.h"

#include <thread>

rq_index* rq_multi_create(int dim, int n_devices, const int* device_ids) {
    rq_index* p = new rq_index();
    p->dim = dim;
    p->device = device_ids[0];
    for (int j = 0; j < n_devices; ++j) {
        rq_index* c = rq_index_create(dim, 1, device_ids + j);   // (the same device may be named several times)
        if (!c) { rq_index_destroy(p); return nullptr; }
        p->shards.push_back(c);
    }
    return p;
}

// rows of child j when the parent holds n rows
static int64_t child_rows(const rq_index* idx, size_t j, int64_t n) {
    const int64_t S = idx->stripe, G = (int64_t)idx->shards.size();
    const int64_t full = n / S, rem = n % S;          // complete stripes, rows of the open one
    int64_t r = (full / G) * S + ((int64_t)j < full % G ? S : 0);
    if ((int64_t)j == full % G) r += rem;
    return r;
}

int rq_multi_reserve(rq_index* idx, int64_t n_rows) {
    for (size_t j = 0; j < idx->shards.size(); ++j)
        if (int r = rq_index_reserve(idx->shards[j], child_rows(idx, j, n_rows))) return r;
    return RQ_OK;
}

int rq_multi_add(rq_index* idx, const void* rows, int64_t n_rows, bool is_f32, int normalize) {
    if (idx->poisoned) return set_err(RQ_EHIP, "multi-device index is unusable: an earlier append failed part-way (%s)", "rq_index_destroy it and rebuild");
    if ((uint64_t)idx->row_offset + (uint64_t)idx->n + (uint64_t)n_rows >= 0xffffffffull)
        return set_err(RQ_EUNSUPPORTED, "row ids beyond 2^32-1 are not supported");
    const int64_t S = idx->stripe, G = (int64_t)idx->shards.size();
    const size_t esz = is_f32 ? 4 : 2;
    for (int64_t done = 0; done < n_rows;) {
        const int64_t g0 = idx->n, s = g0 / S;
        const int64_t m = std::min(n_rows - done, (s + 1) * S - g0);   // up to the end of the open stripe
        rq_index* c = idx->shards[(size_t)(s % G)];
        if (c->n != (s / G) * S + g0 % S)
            return set_err(RQ_EHIP, "internal: device slot %lld holds %lld rows, the stripe layout expects %lld", (long long)(s % G), (long long)c->n, (long long)((s / G) * S + g0 % S));
        if (int r = rq_add_host_common(c, (const char*)rows + (size_t)done * idx->dim * esz, m, is_f32, normalize)) {
            // The child may hold part of the piece (host rows are streamed in chunks): the parent can no longer tell which
            // global rows exist.  Rows of EARLIER pieces of this call are complete and counted; the index refuses further use.
            idx->poisoned = c->n != (s / G) * S + g0 % S;
            return set_err(r, "multi-device append failed on device slot %lld after %lld of %lld rows of the block were stored%s: %s",
                           (long long)(s % G), (long long)done, (long long)n_rows, idx->poisoned ? " (index unusable)" : "", std::string(rq_err_text()).c_str());
        }
        idx->n += m;
        done += m;
    }
    return RQ_OK;
}

int rq_multi_get_rows(const rq_index* idx, int64_t row_begin, int64_t n_rows, uint16_t* out) {
    const int64_t S = idx->stripe, G = (int64_t)idx->shards.size();
    for (int64_t g = row_begin, end = row_begin + n_rows; g < end;) {
        const int64_t s = g / S, m = std::min(end - g, (s + 1) * S - g);
        if (int r = rq_index_get_rows_f16(idx->shards[(size_t)(s % G)], (s / G) * S + g % S, m, out + (size_t)(g - row_begin) * idx->dim)) return r;
        g += m;
    }
    return RQ_OK;
}

// Per query: the children's k best (score, local row) pairs arrive best first; local order is monotone in the global id, so
// each list is already in canonical order after the mapping and a k-way merge of the list heads yields the global top-k.
static void merge_queries(const rq_index* idx, const std::vector<size_t>& live, int q_lo, int q_hi, int k, float* out_scores, int64_t* out_rows) {
    const int64_t S = idx->stripe, G = (int64_t)idx->shards.size();
    const size_t nl = live.size();
    size_t head[64];
    for (int q = q_lo; q < q_hi; ++q) {
        for (size_t a = 0; a < nl; ++a) head[a] = 0;
        for (int i = 0; i < k; ++i) {
            uint64_t best = 0;
            size_t who = nl;
            for (size_t a = 0; a < nl; ++a) {
                if (head[a] >= (size_t)k) continue;
                const size_t at = (a * (size_t)idx->m_B + (size_t)q) * (size_t)k + head[a];
                const int64_t lr = idx->m_rows[at];
                if (lr < 0) { head[a] = (size_t)k; continue; }          // padding: this child's list is exhausted
                const int64_t j = (int64_t)live[a];
                const int64_t grow = ((lr / S) * G + j) * S + lr % S;
                const uint64_t key = rq_make_key(idx->m_scores[at], (uint32_t)grow);
                if (key > best) { best = key; who = a; }
            }
            if (who == nl) { out_scores[(size_t)q * k + i] = 0.f; out_rows[(size_t)q * k + i] = -1; continue; }
            head[who]++;
            out_scores[(size_t)q * k + i] = rq_key_score(best);
            out_rows[(size_t)q * k + i] = idx->row_offset + (int64_t)rq_key_index(best);
        }
    }
}

int rq_multi_search(rq_index* idx, const float* queries, int B, int k, int metric, float* out_scores, int64_t* out_rows) {
    if (idx->poisoned) return set_err(RQ_EHIP, "multi-device index is unusable: an earlier append failed part-way");
    const size_t g = idx->shards.size();
    idx->t.searches++;
    idx->t.queries += B;
    std::vector<size_t>& live = idx->m_live;
    live.clear();
    int rc = RQ_OK;
    for (size_t j = 0; j < g && rc == RQ_OK; ++j)
        if (idx->shards[j]->n > 0) {
            rc = rq_search_begin(idx->shards[j], queries, B, k, metric);   // every device is busy before any is waited for
            if (rc == RQ_OK) live.push_back(j);
        }
    // result staging [live child][B][k], kept between calls
    const size_t need = live.size() * (size_t)B * (size_t)k;
    if (idx->m_scores.size() < need) { idx->m_scores.resize(need); idx->m_rows.resize(need); }
    idx->m_B = B;
    std::string first_err;
    if (rc != RQ_OK) first_err = rq_err_text();
    for (size_t a = 0; a < live.size(); ++a) {
        // (after a failure the searches already enqueued are still waited for and their staging released; their results are dropped)
        const int r = rq_search_end(idx->shards[live[a]], B, k, metric, idx->m_scores.data() + a * (size_t)B * k, idx->m_rows.data() + a * (size_t)B * k);
        if (r != RQ_OK && rc == RQ_OK) { rc = r; first_err = rq_err_text(); }
    }
    if (rc != RQ_OK) return set_err(rc, "%s", first_err.c_str());
    // host merge: O(B k G); calls with many queries split them over a few threads
    const size_t work = (size_t)B * (size_t)k * live.size();
    const int nthr = work < 200000 ? 1 : (int)std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency() / 2));
    if (nthr <= 1) merge_queries(idx, live, 0, B, k, out_scores, out_rows);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t)
            th.emplace_back(merge_queries, idx, std::cref(live), (int)((int64_t)B * t / nthr), (int)((int64_t)B * (t + 1) / nthr), k, out_scores, out_rows);
        for (auto& x : th) x.join();
    }
    return RQ_OK;
}
