// Host-side property check of the bit-level helpers in csrc/rq_device.h (compiled with hipcc, runs without a GPU).
// Every decoded field of a scan record must be an UPPER bound of what was encoded and the codes must preserve order:
// that is what the exactness argument of the tail rests on.
#include "rq_device.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { std::printf("FAIL %s:%d %s  ", __FILE__, __LINE__, #c); std::printf(__VA_ARGS__); std::printf("\n"); } ++fails; } } while (0)

static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main() {
    std::mt19937_64 rng(12345);
    std::vector<float> v = {0.f, -0.f, 1.f, -1.f, 0.15f, -0.15f, 1e-30f, -1e-30f, 1e-40f, -1e-40f, 3.4e38f, -3.4e38f,
                            std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(),
                            from_bits(0x3e19999au), from_bits(0x3e19ffffu), from_bits(0x3e1a0000u), from_bits(0xbe19ffffu)};
    for (int i = 0; i < 200000; ++i) {
        const uint32_t u = (uint32_t)rng();
        const float f = from_bits(u);
        if (f == f) v.push_back(f);                                   // no NaN: the scan drops NaN scores before encoding
        v.push_back(((int64_t)(rng() % 2000001) - 1000000) * 1e-6f);  // the range cosine scores live in
    }
    for (float f : v) {
        // mono32: order preserving and invertible
        CHECK(rq_unmono32(rq_mono32(f)) == f || (f == 0.f), "f=%g", f);
        // up16 / up26: upper bounds with the low bits cleared
        const float u16 = from_bits(rq_up16(f)), u26 = rq_rec_m1(rq_up26(f) | 37u);
        CHECK(u16 >= f && (rq_up16(f) & 0xffffu) == 0, "f=%g up16=%g", f, u16);
        CHECK(u26 >= f && u26 <= u16, "f=%g up26=%g up16=%g", f, u26, u16);
        CHECK((rq_up26(f) & 63u) == 0, "f=%g", f);
        // code16: decodes to exactly the 16-bit rounded-up value
        const uint32_t c = rq_code16(f);
        CHECK(c < 65536u, "f=%g", f);
        CHECK(rq_code16_value(c) == u16 || (u16 == 0.f && rq_code16_value(c) == 0.f), "f=%g code=%u value=%g up16=%g", f, c, rq_code16_value(c), u16);
    }
    // monotonicity on random pairs
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
        float a = v[i], b = v[i + 1];
        if (a > b) std::swap(a, b);
        if (a < b) CHECK(rq_mono32(a) < rq_mono32(b), "a=%g b=%g", a, b);
        CHECK(rq_code16(a) <= rq_code16(b), "a=%g b=%g", a, b);
        CHECK(rq_rec_m1(rq_up26(a)) <= rq_rec_m1(rq_up26(b)), "a=%g b=%g", a, b);
        // the record's third-score field: c3 <= c2, d = min(c2 - c3, 1023), decode(c2 - d) >= third score
        const uint32_t c2 = rq_code16(b), c3 = rq_code16(a), d = (c2 - c3) < 1023u ? (c2 - c3) : 1023u;
        CHECK(c3 <= c2 && rq_code16_value(c2 - d) >= a, "a=%g b=%g", a, b);
        // keys: (score desc, row asc) == key desc
        const uint32_t r1 = (uint32_t)(rng() % 1000000), r2 = (uint32_t)(rng() % 1000000);
        const uint64_t k1 = rq_make_key(a, r1), k2 = rq_make_key(b, r2);
        if (a < b) CHECK(k1 < k2, "a=%g b=%g", a, b);
        if (a == b && r1 != r2) CHECK((k1 > k2) == (r1 < r2), "a=%g r1=%u r2=%u", a, r1, r2);
        CHECK(rq_key_index(k1) == r1 && (rq_key_score(k1) == a || a == 0.f), "a=%g r1=%u", a, r1);
    }
    std::printf("%s: %zu values, %d failures\n", fails ? "FAILED" : "ok", v.size(), fails);
    return fails ? 1 : 0;
}
