// Host-side check of the multi-device parent (csrc/rq_multi.hip: stripe arithmetic, row read-back, k-way merge, merge threads,
// error paths) behind STUBBED shards: the children here are plain host arrays with a brute-force search, so the whole parent
// runs on the CPU and can be built with -fsanitize=address,undefined (the GPU boxes of this pool cannot run sanitizers).
// The stub is test scaffolding, not a CPU backend: nothing in the product links it.
//   hipcc -O1 -g -std=c++17 --offload-host-only -fsanitize=address,undefined -I <csrc> tests/native/multi_check.cpp -o multi_check
#include "rq_multi.hip"

#include <cstdlib>
#include <random>

static thread_local char g_err[512] = "";
int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* rq_err_text() { return g_err; }

static int g_fail_add_at = -1, g_adds = 0, g_fail_begin_child = -1, g_ends = 0;
static std::vector<rq_index*> g_children;

extern "C" rq_index* rq_index_create(int dim, int n_devices, const int* device_ids) {
    if (n_devices > 1) return rq_multi_create(dim, n_devices, device_ids);
    rq_index* c = new rq_index();
    c->dim = dim;
    c->device = device_ids[0];
    g_children.push_back(c);
    return c;
}
extern "C" void rq_index_destroy(rq_index* idx) {
    if (!idx) return;
    for (rq_index* c : idx->shards) rq_index_destroy(c);
    std::free(idx->x);
    delete idx;
}
extern "C" int rq_index_reserve(rq_index* idx, int64_t n_rows) {
    if (!idx->shards.empty()) return rq_multi_reserve(idx, n_rows);
    if (n_rows > idx->cap) {
        idx->x = (char*)std::realloc(idx->x, (size_t)n_rows * idx->dim * sizeof(float));
        idx->cap = n_rows;
    }
    return RQ_OK;
}
// stub child: rows kept as fp32 on the host (is_f32 appends only)
int rq_add_host_common(rq_index* idx, const void* rows, int64_t n_rows, bool is_f32, int) {
    if (!idx->shards.empty()) return rq_multi_add(idx, rows, n_rows, is_f32, 0);
    if (!is_f32) return set_err(RQ_EINVAL, "stub: f32 rows only");
    if (g_adds++ == g_fail_add_at) {
        // a failure AFTER part of the piece was stored (what a chunked host append can do)
        const int64_t half = n_rows / 2;
        rq_index_reserve(idx, std::max<int64_t>(idx->cap, (idx->n + half) * 2));
        std::memcpy(idx->x + (size_t)idx->n * idx->dim * 4, rows, (size_t)half * idx->dim * 4);
        idx->n += half;
        return set_err(RQ_EHIP, "stub: injected append failure");
    }
    if (idx->n + n_rows > idx->cap) rq_index_reserve(idx, std::max<int64_t>(idx->cap * 2, idx->n + n_rows));
    std::memcpy(idx->x + (size_t)idx->n * idx->dim * 4, rows, (size_t)n_rows * idx->dim * 4);
    idx->n += n_rows;
    return RQ_OK;
}
extern "C" int rq_index_get_rows_f16(const rq_index* idx, int64_t row_begin, int64_t n_rows, uint16_t* out) {
    if (!idx->shards.empty()) return rq_multi_get_rows(idx, row_begin, n_rows, out);
    if (row_begin < 0 || row_begin + n_rows > idx->n) return set_err(RQ_EINVAL, "stub: row range");
    const float* x = (const float*)idx->x;
    for (int64_t i = 0; i < n_rows * idx->dim; ++i) out[i] = (uint16_t)(int)(x[(size_t)row_begin * idx->dim + i] * 100.f + 30000.f);   // (a recognisable code, not fp16)
    return RQ_OK;
}
struct Pending { std::vector<float> q; };
static std::map<rq_index*, Pending> g_pending;
int rq_search_begin(rq_index* idx, const float* queries, int B, int, int) {
    const int which = (int)(std::find(g_children.begin(), g_children.end(), idx) - g_children.begin());
    if (which == g_fail_begin_child) return set_err(RQ_EHIP, "stub: injected search failure on child %d", which);
    g_pending[idx].q.assign(queries, queries + (size_t)B * idx->dim);
    return RQ_OK;
}
int rq_search_end(rq_index* idx, int B, int k, int, float* out_scores, int64_t* out_rows) {
    ++g_ends;
    const std::vector<float>& q = g_pending[idx].q;
    const float* x = (const float*)idx->x;
    for (int b = 0; b < B; ++b) {
        std::vector<std::pair<uint64_t, int64_t>> keys;
        for (int64_t r = 0; r < idx->n; ++r) {
            double dot = 0;
            for (int j = 0; j < idx->dim; ++j) dot += (double)q[(size_t)b * idx->dim + j] * x[(size_t)r * idx->dim + j];
            keys.push_back({rq_make_key((float)dot, (uint32_t)r), r});
        }
        std::sort(keys.begin(), keys.end(), [](auto& a, auto& c) { return a.first > c.first; });
        for (int i = 0; i < k; ++i) {
            const bool v = (size_t)i < keys.size();
            out_scores[(size_t)b * k + i] = v ? rq_key_score(keys[(size_t)i].first) : 0.f;
            out_rows[(size_t)b * k + i] = v ? keys[(size_t)i].second : -1;
        }
    }
    g_pending.erase(idx);
    return RQ_OK;
}

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { std::printf("FAIL %s:%d %s  ", __FILE__, __LINE__, #c); std::printf(__VA_ARGS__); std::printf("\n"); } ++fails; } } while (0)

int main() {
    const int dim = 6;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd;
    for (int G : {2, 3, 5}) {
        for (int64_t stripe : {64, 192}) {
            g_children.clear();
            int devs[5] = {0, 0, 0, 0, 0};
            rq_index* multi = rq_index_create(dim, G, devs);
            multi->stripe = stripe;
            int one = 0;
            rq_index* single = rq_index_create(dim, 1, &one);
            g_children.pop_back();                                   // (only the multi's children are numbered for failure injection)
            const int64_t total = 5 * stripe * G + 37;
            std::vector<float> rows((size_t)total * dim);
            for (auto& v : rows) v = std::round(nd(rng) * 8.f) / 8.f;   // coarse values: plenty of exactly tied scores
            for (int64_t i = 0; i < dim; ++i) rows[(size_t)(total - 1) * dim + i] = rows[(size_t)3 * dim + i];   // a duplicate far apart: tie across children
            // ragged appends: 1 row, blocks that end exactly on a stripe end, blocks that span several stripes
            int64_t done = 0;
            const int64_t sizes[] = {1, stripe - 1, stripe, 2 * stripe + 5, 7, 3 * stripe, 50};
            for (int i = 0; done < total; ++i) {
                const int64_t m = std::min(sizes[i % 7], total - done);
                CHECK(rq_add_host_common(multi, rows.data() + (size_t)done * dim, m, true, 0) == RQ_OK, "append: %s", rq_err_text());
                done += m;
            }
            CHECK(rq_add_host_common(single, rows.data(), total, true, 0) == RQ_OK, "single append");
            CHECK(multi->n == total, "size %lld", (long long)multi->n);
            int64_t sum = 0;
            for (rq_index* c : multi->shards) sum += c->n;
            CHECK(sum == total, "children hold %lld rows", (long long)sum);
            std::vector<uint16_t> a((size_t)total * dim), b((size_t)total * dim);
            CHECK(rq_index_get_rows_f16(multi, 0, total, a.data()) == RQ_OK && rq_index_get_rows_f16(single, 0, total, b.data()) == RQ_OK && a == b, "read-back differs");
            CHECK(rq_index_get_rows_f16(multi, stripe - 3, 2 * stripe + 9, a.data()) == RQ_OK &&
                  std::memcmp(a.data(), b.data() + (size_t)(stripe - 3) * dim, (size_t)(2 * stripe + 9) * dim * 2) == 0, "ranged read-back differs");
            for (int B : {1, 9, 700}) {                               // 700 x 40 x G keys: the threaded merge
                const int k = B == 700 ? 40 : 13;
                std::vector<float> q((size_t)B * dim);
                for (auto& v : q) v = std::round(nd(rng) * 4.f) / 4.f;
                for (int j = 0; j < dim; ++j) q[j] = rows[(size_t)3 * dim + j];
                std::vector<float> s1((size_t)B * k), s2((size_t)B * k);
                std::vector<int64_t> r1((size_t)B * k), r2((size_t)B * k);
                multi->row_offset = 1000;
                CHECK(rq_multi_search(multi, q.data(), B, k, 0, s1.data(), r1.data()) == RQ_OK, "search: %s", rq_err_text());
                CHECK(rq_search_begin(single, q.data(), B, k, 0) == RQ_OK && rq_search_end(single, B, k, 0, s2.data(), r2.data()) == RQ_OK, "single search");
                for (auto& r : r2) r += 1000;
                CHECK(r1 == r2 && s1 == s2, "G=%d stripe=%lld B=%d: merged result differs from the single index (first rows %lld vs %lld)", G, (long long)stripe, B,
                      (long long)r1[0], (long long)r2[0]);
                multi->row_offset = 0;
            }
            // k larger than the rows of a child: -1 padding inside the children's lists
            {
                const int k = (int)stripe + 20;
                std::vector<float> q(dim, 1.f), s1((size_t)k), s2((size_t)k);
                std::vector<int64_t> r1((size_t)k), r2((size_t)k);
                rq_index* tiny = rq_index_create(dim, G, devs);
                tiny->stripe = stripe;
                CHECK(rq_add_host_common(tiny, rows.data(), stripe + 3, true, 0) == RQ_OK, "tiny append");
                CHECK(rq_multi_search(tiny, q.data(), 1, k, 0, s1.data(), r1.data()) == RQ_OK, "tiny search");
                CHECK(r1[(size_t)stripe + 2] >= 0 && r1[(size_t)stripe + 3] == -1 && s1[(size_t)stripe + 3] == 0.f, "padding after %lld rows", (long long)stripe + 3);
                rq_index_destroy(tiny);
            }
            // a search that fails on one child still ends the searches already enqueued on the others
            g_fail_begin_child = G - 1;
            g_ends = 0;
            {
                std::vector<float> q(dim, 0.5f), s1(5);
                std::vector<int64_t> r1(5);
                CHECK(rq_multi_search(multi, q.data(), 1, 5, 0, s1.data(), r1.data()) == RQ_EHIP, "failure not reported");
                CHECK(g_ends == G - 1 && g_pending.empty(), "%d searches ended, %zu left pending", g_ends, g_pending.size());
                CHECK(std::strstr(rq_err_text(), "injected search failure") != nullptr, "message: %s", rq_err_text());
            }
            g_fail_begin_child = -1;
            // an append that fails part-way inside a child poisons the parent (the global <-> local mapping is gone)
            g_adds = 0;
            g_fail_add_at = 1;
            CHECK(rq_add_host_common(multi, rows.data(), 2 * stripe, true, 0) == RQ_EHIP, "append failure not reported");
            CHECK(multi->poisoned, "parent not poisoned: %s", rq_err_text());
            g_fail_add_at = -1;
            CHECK(rq_add_host_common(multi, rows.data(), 1, true, 0) == RQ_EHIP, "append accepted on a poisoned index");
            {
                std::vector<float> q(dim, 0.5f), s1(5);
                std::vector<int64_t> r1(5);
                CHECK(rq_multi_search(multi, q.data(), 1, 5, 0, s1.data(), r1.data()) == RQ_EHIP, "search accepted on a poisoned index");
            }
            rq_index_destroy(multi);
            rq_index_destroy(single);
        }
    }
    std::printf(fails ? "FAILED: %d checks\n" : "ok: multi-device parent over stubbed shards (stripes, read-back, merge, error paths), 0 failures\n", fails);
    return fails ? 1 : 0;
}
