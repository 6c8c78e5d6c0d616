/* include/rq.h from plain C (gcc -std=c99, no HIP headers): the boundary is a C ABI.
 * Without a GPU every data call must fail loudly (RQ_ENODEVICE / NULL + message); with one, a tiny search must come
 * back exact.  Exit code 0 = as expected, and the last line says which branch ran. */
#include "rq.h"
#include <math.h>
#include <stdio.h>
#include <string.h>

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
     * here the one GPU named three times.  Appended in two blocks, so every slot holds two segments of global ids. */
    {
        int devs[3] = {0, 0, 0};
        rq_index* multi = rq_index_create(8, 3, devs);
        if (!multi) { printf("multi-device create failed: %s\n", rq_last_error()); return 1; }
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
    if (rq_stream_release(idx, NULL) != RQ_OK) { printf("stream release failed: %s\n", rq_last_error()); return 1; }
    rq_index_destroy(idx);
    printf("OK (device search)\n");
    return 0;
}
