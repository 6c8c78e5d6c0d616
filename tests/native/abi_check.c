/* include/rq.h from plain C (gcc -std=c99, no HIP headers): the boundary is a C ABI.
 * Without a GPU every data call must fail loudly (RQ_ENODEVICE / NULL + message); with one, a tiny search must come
 * back exact.  Exit code 0 = as expected, and the last line says which branch ran. */
#define _POSIX_C_SOURCE 199309L
#include "rq.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

int main(void) {
    printf("%s\n", rq_version());
    const int ndev = rq_device_count();
    int dev = 0;
    rq_index* idx = rq_index_create(8, 1, &dev);
    if (ndev <= 0) {
        if (idx != NULL) { printf("index created without a device\n"); return 1; }
        if (strlen(rq_last_error()) == 0) { printf("no error message\n"); return 1; }
        printf("no device: %s\nOK (no device)\n", rq_last_error());
        return 0;
    }
    if (!idx) { printf("create failed: %s\n", rq_last_error()); return 1; }
    float rows[100][8];
    for (int i = 0; i < 100; ++i)
        for (int j = 0; j < 8; ++j) rows[i][j] = sinf((float)(i * 8 + j) * 0.37f) + (j == i % 8 ? 2.0f : 0.0f);
    if (rq_index_add_f32(idx, &rows[0][0], 100, 1) != RQ_OK || rq_index_size(idx) != 100) { printf("add failed: %s\n", rq_last_error()); return 1; }
    float q[2][8];
    memcpy(q[0], rows[42], sizeof q[0]);
    memcpy(q[1], rows[7], sizeof q[1]);
    float scores[2][3];
    int64_t ids[2][3];
    if (rq_search(idx, &q[0][0], 2, 3, RQ_METRIC_COSINE, &scores[0][0], &ids[0][0]) != RQ_OK) { printf("search failed: %s\n", rq_last_error()); return 1; }
    if (ids[0][0] != 42 || ids[1][0] != 7 || fabsf(scores[0][0] - 1.0f) > 1e-3f || scores[0][1] > scores[0][0] || scores[0][2] > scores[0][1]) {
        printf("wrong result: %lld %f %lld %f\n", (long long)ids[0][0], scores[0][0], (long long)ids[1][0], scores[1][0]);
        return 1;
    }
    if (rq_search(idx, &q[0][0], 0, 3, RQ_METRIC_COSINE, &scores[0][0], &ids[0][0]) != RQ_EINVAL) { printf("B = 0 accepted\n"); return 1; }
    /* the same rows sharded across three device slots inside the library (rq_index_create(dim, n_devices, ids));
     * here the one GPU named three times.  Stripes of 64 rows (rq_set_option stripe_rows), appended in two blocks. */
    {
        int devs[3] = {0, 0, 0};
        rq_index* multi = rq_index_create(8, 3, devs);
        if (!multi) { printf("multi-device create failed: %s\n", rq_last_error()); return 1; }
        if (rq_set_option(multi, "stripe_rows", 64.0) != RQ_OK) { printf("stripe_rows refused: %s\n", rq_last_error()); return 1; }
        if (rq_index_add_f32(multi, &rows[0][0], 40, 1) != RQ_OK || rq_index_add_f32(multi, &rows[40][0], 60, 1) != RQ_OK || rq_index_size(multi) != 100) {
            printf("multi-device add failed: %s\n", rq_last_error()); return 1;
        }
        float ms[2][3];
        int64_t mi[2][3];
        if (rq_search(multi, &q[0][0], 2, 3, RQ_METRIC_COSINE, &ms[0][0], &mi[0][0]) != RQ_OK) { printf("multi-device search failed: %s\n", rq_last_error()); return 1; }
        if (memcmp(mi, ids, sizeof mi) != 0 || memcmp(ms, scores, sizeof ms) != 0) {
            printf("multi-device result differs: %lld %f vs %lld %f\n", (long long)mi[0][0], ms[0][0], (long long)ids[0][0], scores[0][0]);
            return 1;
        }
        uint16_t one[8], ref[8];
        if (rq_index_get_rows_f16(multi, 57, 1, one) != RQ_OK || rq_index_get_rows_f16(idx, 57, 1, ref) != RQ_OK || memcmp(one, ref, sizeof one) != 0) {
            printf("multi-device row read-back differs\n"); return 1;
        }
        if (rq_search_device(multi, &q[0][0], 2, 3, RQ_METRIC_COSINE, &ms[0][0], &mi[0][0], NULL, (int*)mi, NULL) != RQ_EUNSUPPORTED) {
            printf("device-pointer search accepted on a multi-device index\n"); return 1;
        }
        rq_index_destroy(multi);
    }
    /* a streaming build (reference StreamingIndex: 100 documents per add): 2 000 appends of 50 rows over three device slots.
     * Stripes of 4 096 rows: an append touches one slot (two when it crosses a stripe end), the global <-> local mapping is
     * arithmetic.  Rows read back and a search must equal the single-device index built from the same rows. */
    {
        enum { APPENDS = 2000, PER = 50, DIM = 8, TOTAL = APPENDS * PER };
        static float big[TOTAL][DIM];
        static uint16_t back_m[TOTAL][DIM], back_s[TOTAL][DIM];
        for (int i = 0; i < TOTAL; ++i)
            for (int j = 0; j < DIM; ++j) big[i][j] = sinf((float)i * 0.013f + (float)j * 1.7f) + cosf((float)(i % 977) * 0.11f * (float)(j + 1));
        int devs[3] = {0, 0, 0};
        rq_index* multi = rq_index_create(DIM, 3, devs);
        rq_index* single = rq_index_create(DIM, 1, &dev);
        if (!multi || !single) { printf("create failed: %s\n", rq_last_error()); return 1; }
        if (rq_set_option(multi, "stripe_rows", 4096.0) != RQ_OK) { printf("stripe_rows refused: %s\n", rq_last_error()); return 1; }
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int a = 0; a < APPENDS; ++a)
            if (rq_index_add_f32(multi, &big[a * PER][0], PER, 1) != RQ_OK) { printf("append %d failed: %s\n", a, rq_last_error()); return 1; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double ms_multi = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int a = 0; a < APPENDS; ++a)
            if (rq_index_add_f32(single, &big[a * PER][0], PER, 1) != RQ_OK) { printf("append %d failed: %s\n", a, rq_last_error()); return 1; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double ms_single = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
        if (rq_index_size(multi) != TOTAL || rq_set_option(multi, "stripe_rows", 64.0) == RQ_OK) { printf("size / late stripe_rows\n"); return 1; }
        if (rq_index_get_rows_f16(multi, 0, TOTAL, &back_m[0][0]) != RQ_OK || rq_index_get_rows_f16(single, 0, TOTAL, &back_s[0][0]) != RQ_OK ||
            memcmp(back_m, back_s, sizeof back_m) != 0) { printf("streaming build: rows differ from the single index\n"); return 1; }
        float qs[4][DIM], s1[4][7], s2[4][7];
        int64_t r1[4][7], r2[4][7];
        for (int i = 0; i < 4; ++i) memcpy(qs[i], big[4095 + i * 31337 % TOTAL], sizeof qs[i]);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int rep = 0; rep < 20; ++rep)
            if (rq_search(multi, &qs[0][0], 4, 7, RQ_METRIC_COSINE, &s1[0][0], &r1[0][0]) != RQ_OK) { printf("search failed: %s\n", rq_last_error()); return 1; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double us_search = ((double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3) / 20.0;
        if (rq_search(single, &qs[0][0], 4, 7, RQ_METRIC_COSINE, &s2[0][0], &r2[0][0]) != RQ_OK) { printf("search failed: %s\n", rq_last_error()); return 1; }
        if (memcmp(r1, r2, sizeof r1) != 0 || memcmp(s1, s2, sizeof s1) != 0) {
            printf("streaming build: search differs (%lld vs %lld)\n", (long long)r1[0][0], (long long)r2[0][0]); return 1;
        }
        printf("streaming build: %d appends of %d rows: 3 device slots %.1f ms (%.1f us per append), one device %.1f ms; search of 4 queries over 3 slots %.1f us\n",
               APPENDS, PER, ms_multi, ms_multi * 1e3 / APPENDS, ms_single, us_search);
        rq_index_destroy(multi);
        rq_index_destroy(single);
    }
    if (rq_stream_release(idx, NULL) != RQ_OK) { printf("stream release failed: %s\n", rq_last_error()); return 1; }
    rq_index_destroy(idx);
    printf("OK (device search)\n");
    return 0;
}
