// librq_bm25's scorer (csrc/rq_bm25.cpp, compiled INTO this program so that -fsanitize=address,undefined sees it) against a
// straightforward per-document reference: same scores bit for bit (same additions in the same order), same selection
// (score > 0, best first, ties by descending row), padding, threads, refused operands.
//   g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined tests/native/bm25_check.cpp <csrc>/rq_bm25.cpp -I include -o bm25_check
#include "rq_bm25.h"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { std::printf("FAIL %s:%d %s  ", __FILE__, __LINE__, #c); std::printf(__VA_ARGS__); std::printf("\n"); } ++fails; } } while (0)

int main() {
    std::mt19937 g(3);
    const int64_t T = 300, N = 2000;
    std::vector<int64_t> ip(T + 1, 0);
    std::vector<int32_t> rows;
    std::vector<double> c;
    for (int64_t t = 0; t < T; ++t) {
        const double p = 0.4 / (1 + t * 0.05);
        for (int64_t d = 0; d < N; ++d)
            if ((g() & 0xffff) < p * 65536) { rows.push_back((int32_t)d); c.push_back(t % 17 == 0 ? -0.25 : 0.5 + (g() & 7) / 8.0); }   // few distinct values: ties
        ip[t + 1] = (int64_t)rows.size();
    }
    rq_bm25* h = rq_bm25_create(ip.data(), rows.data(), c.data(), T, N);
    CHECK(h != nullptr, "create");
    const int B = 64;
    std::vector<int64_t> qp(B + 1, 0);
    std::vector<int32_t> qt;
    for (int q = 0; q < B; ++q) {
        const int len = q == 5 ? 0 : 1 + (int)(g() % 9);
        for (int i = 0; i < len; ++i) qt.push_back((int32_t)(g() % (q % 2 ? 30 : T)));
        if (q == 7) { qt.push_back(qt.back()); qt.push_back(qt.back()); }          // a repeated token
        qp[q + 1] = (int64_t)qt.size();
    }
    for (int k : {1, 10, 3000}) {
        for (int nt : {1, 3, 0}) {
            std::vector<int32_t> orow((size_t)B * k);
            std::vector<double> osc((size_t)B * k);
            CHECK(rq_bm25_topk(h, qp.data(), qt.data(), B, k, orow.data(), osc.data(), nt) == RQ_BM25_OK, "topk");
            for (int q = 0; q < B; ++q) {
                std::vector<double> acc((size_t)N, 0.0);
                for (int64_t t = qp[q]; t < qp[q + 1]; ++t)
                    for (int64_t p = ip[qt[t]]; p < ip[qt[t] + 1]; ++p) acc[(size_t)rows[p]] += c[p];
                std::vector<int32_t> order;
                for (int32_t d = 0; d < N; ++d) if (acc[(size_t)d] > 0.0) order.push_back(d);
                std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return acc[(size_t)a] > acc[(size_t)b] || (acc[(size_t)a] == acc[(size_t)b] && a > b); });
                for (int i = 0; i < k; ++i) {
                    const bool v = (size_t)i < order.size();
                    CHECK(orow[(size_t)q * k + i] == (v ? order[(size_t)i] : -1), "q %d rank %d: row %d", q, i, orow[(size_t)q * k + i]);
                    CHECK(osc[(size_t)q * k + i] == (v ? acc[(size_t)order[(size_t)i]] : 0.0), "q %d rank %d: score", q, i);
                }
            }
        }
    }
    std::vector<int32_t> one(4);
    std::vector<double> onesc(4);
    const int64_t badp[2] = {0, 1};
    const int32_t badt[1] = {(int32_t)T};
    CHECK(rq_bm25_topk(h, badp, badt, 1, 4, one.data(), onesc.data(), 1) == RQ_BM25_EINVAL, "token id outside the index accepted");
    CHECK(rq_bm25_topk(h, qp.data(), qt.data(), B, 0, one.data(), onesc.data(), 1) == RQ_BM25_EINVAL, "k = 0 accepted");
    CHECK(rq_bm25_topk(h, qp.data(), qt.data(), 0, 4, one.data(), onesc.data(), 1) == RQ_BM25_OK, "empty batch refused");
    rq_bm25_destroy(h);
    std::vector<int32_t> badrows = rows;
    badrows[rows.size() / 2] = (int32_t)N;
    CHECK(rq_bm25_create(ip.data(), badrows.data(), c.data(), T, N) == nullptr, "row outside the corpus accepted");
    std::printf(fails ? "FAILED: %d checks\n" : "ok: batched BM25 scorer equals the per-document reference, 0 failures\n", fails);
    return fails ? 1 : 0;
}
