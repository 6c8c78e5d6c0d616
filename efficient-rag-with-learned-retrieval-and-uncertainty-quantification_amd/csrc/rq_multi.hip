// rq_multi.hip -- the multi-device parent index (include/rq.h rq_index_create with n_devices > 1): contiguous row blocks
// per device inside the library.  The reference has no counterpart (its dense store is one ChromaDB collection,
// rag_uq/streaming_index.py:252-263); this is the north star's row sharding, for callers that want several GPUs behind
// ONE index handle instead of one process per GPU (rag_uq_amd.distributed).
#include "rq_index.h"

// ---------------------------------------------------------------------------------------------
// multi-device parent: contiguous row blocks per device inside the library
// ---------------------------------------------------------------------------------------------
rq_index* rq_multi_create(int dim, int n_devices, const int* device_ids) {
    rq_index* p = new rq_index();
    p->dim = dim;
    p->device = device_ids[0];
    for (int j = 0; j < n_devices; ++j) {
        rq_index* c = rq_index_create(dim, 1, device_ids + j);   // (the same device may be named several times)
        if (!c) { rq_index_destroy(p); return nullptr; }
        p->shards.push_back(c);
        p->seg_local.push_back({0});
        p->seg_global.push_back({});
    }
    return p;
}

int rq_multi_add(rq_index* idx, const void* rows, int64_t n_rows, bool is_f32, int normalize) {
    if ((uint64_t)idx->row_offset + (uint64_t)idx->n + (uint64_t)n_rows >= 0xffffffffull)
        return set_err(RQ_EUNSUPPORTED, "row ids beyond 2^32-1 are not supported");
    const int64_t g = (int64_t)idx->shards.size(), per = (n_rows + g - 1) / g;
    const size_t esz = is_f32 ? 4 : 2;
    for (int64_t j = 0; j < g; ++j) {
        const int64_t lo = std::min(j * per, n_rows), hi = std::min((j + 1) * per, n_rows);
        if (hi <= lo) continue;
        if (int r = rq_add_host_common(idx->shards[j], (const char*)rows + (size_t)lo * idx->dim * esz, hi - lo, is_f32, normalize)) {
            // pieces 0..j-1 of this block are already stored: the parent is no longer a prefix of what the caller sent
            return set_err(r, "multi-device append failed on device slot %lld after %lld of %lld rows of the block were stored: %s",
                           (long long)j, (long long)lo, (long long)n_rows, std::string(rq_err_text()).c_str());
        }
        idx->seg_global[j].push_back(idx->n + lo);
        idx->seg_local[j].push_back(idx->seg_local[j].back() + (hi - lo));
    }
    idx->n += n_rows;
    return RQ_OK;
}

int rq_multi_get_rows(const rq_index* idx, int64_t row_begin, int64_t n_rows, uint16_t* out) {
    const int64_t row_end = row_begin + n_rows;
    for (size_t j = 0; j < idx->shards.size(); ++j)
        for (size_t sgi = 0; sgi < idx->seg_global[j].size(); ++sgi) {
            const int64_t g0 = idx->seg_global[j][sgi], len = idx->seg_local[j][sgi + 1] - idx->seg_local[j][sgi];
            const int64_t lo = std::max(g0, row_begin), hi = std::min(g0 + len, row_end);
            if (hi <= lo) continue;
            if (int r = rq_index_get_rows_f16(idx->shards[j], idx->seg_local[j][sgi] + (lo - g0), hi - lo, out + (size_t)(lo - row_begin) * idx->dim)) return r;
        }
    return RQ_OK;
}

int rq_multi_search(rq_index* idx, const float* queries, int B, int k, int metric, float* out_scores, int64_t* out_rows) {
    const size_t g = idx->shards.size();
    idx->t.searches++;
    idx->t.queries += B;
    std::vector<size_t> live;
    for (size_t j = 0; j < g; ++j)
        if (idx->shards[j]->n > 0) {
            if (int r = rq_search_begin(idx->shards[j], queries, B, k, metric)) return r;   // every device is busy before any is waited for
            live.push_back(j);
        }
    std::vector<float> sc((size_t)B * k);
    std::vector<int64_t> rw((size_t)B * k);
    std::vector<std::vector<uint64_t>> keys((size_t)B);
    for (size_t j : live) {
        if (int r = rq_search_end(idx->shards[j], B, k, metric, sc.data(), rw.data())) return r;
        const std::vector<int64_t>& sl = idx->seg_local[j];
        const std::vector<int64_t>& sg = idx->seg_global[j];
        for (int q = 0; q < B; ++q)
            for (int i = 0; i < k; ++i) {
                const int64_t lr = rw[(size_t)q * k + i];
                if (lr < 0) continue;
                const size_t sgi = (size_t)(std::upper_bound(sl.begin(), sl.end() - 1, lr) - sl.begin()) - 1;   // segment that holds the local row
                const int64_t grow = sg[sgi] + (lr - sl[sgi]);
                keys[(size_t)q].push_back(rq_make_key(sc[(size_t)q * k + i], (uint32_t)grow));
            }
    }
    for (int q = 0; q < B; ++q) {   // canonical order: score descending, then global row ascending = key descending
        std::vector<uint64_t>& kq = keys[(size_t)q];
        const size_t m = std::min<size_t>((size_t)k, kq.size());
        std::partial_sort(kq.begin(), kq.begin() + m, kq.end(), std::greater<uint64_t>());
        for (int i = 0; i < k; ++i) {
            const bool v = (size_t)i < m;
            out_scores[(size_t)q * k + i] = v ? rq_key_score(kq[(size_t)i]) : 0.f;
            out_rows[(size_t)q * k + i] = v ? idx->row_offset + (int64_t)rq_key_index(kq[(size_t)i]) : -1;
        }
    }
    return RQ_OK;
}

