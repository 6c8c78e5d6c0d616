// rq_final_body.h -- final top-k of one query's candidate keys + exactness certificate.
// Shared by rq_final_fast_kernel (rq_select.hip) and the last-arriving workgroup of rq_tail_kernel (rq_tail.hip).
// Called by all 256 threads of a workgroup.
//
// The candidates are cnt bins x binrows keys.  Only ~k of them matter, so:
//   1. best key of every bin (staged through LDS, 1024 keys per pass),
//   2. t2 = k-th largest bin-best: k distinct keys are >= t2, so keys < t2 cannot be in the top-k,
//   3. survivors (keys >= t2, about k of them) are ranked against each other and written by rank.
// If more than 1024 keys survive (massive exact ties) the k-round "extract the maximum" loop runs instead.
#pragma once
#include "rq_device.h"

struct RqFinalLds {
    uint64_t skeys[1024];
    uint64_t bb[256];
    uint64_t wbest[2][4];
    uint64_t skth, t2;
    int ns;
};

struct RqFinalCore {
    const uint64_t* cand;   // this query's candidate keys, cnt * binrows of them are valid
    int rmax, binrows, metric, k;
    float eps, max_row_norm;
    int64_t row_offset, n_rows;
    float* out_scores;      // [k] of this query
    int64_t* out_rows;
    uint64_t* out_keys;     // may be null
    int* out_status;        // this query's slot
};

__device__ __forceinline__ void rq_emit(const RqFinalCore& a, int rank, uint64_t key) {
    const float s = rq_key_score(key);
    const int64_t grow = a.row_offset + (int64_t)rq_key_index(key);
    a.out_scores[rank] = s;
    a.out_rows[rank] = grow;
    if (a.out_keys) a.out_keys[rank] = rq_make_key(s, (uint32_t)grow);
}

// total: bins that wanted a slot (may exceed rmax = overflow); overflow: any other overflow seen for this query;
// T: every bin outside the candidate list has approximate score < T;  qn: fp64 norm of the query.
// COHERENT: the keys were published by other workgroups of the SAME launch with sc1 (write-through) stores and a
// ticket; every load of them must then be an sc1 load (relaxed agent-scope atomic load), never a plain one.
template <bool COHERENT>
__device__ __forceinline__ void rq_final_body(const RqFinalCore& a, int total, int overflow, float T, double qn, RqFinalLds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t kk = a.k < a.n_rows ? a.k : a.n_rows;
    if (qn == 0.0) {   // every score is exactly 0: rows 0 .. kk-1 (score desc, row asc)
        for (int j = tid; j < a.k; j += 256) {
            const bool v = j < kk;
            a.out_scores[j] = 0.f;
            a.out_rows[j] = v ? a.row_offset + j : -1;
            if (a.out_keys) a.out_keys[j] = v ? rq_make_key(0.f, (uint32_t)(a.row_offset + j)) : 0;
        }
        if (tid == 0) *a.out_status = 0;
        return;
    }
    const int cnt = total < a.rmax ? total : a.rmax;   // <= 256
    const int n = cnt * a.binrows;                     // <= 4096
    uint64_t key[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = i * 256 + tid;
        if (COHERENT) key[i] = j < n ? __hip_atomic_load(a.cand + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        else key[i] = j < n ? a.cand[j] : 0;
    }
    if (tid == 0) { L.skth = 0; L.t2 = 0; L.ns = 0; }
    // 1. bin bests, 1024 keys (= 1024 / binrows bins) per pass
    const int bins_per_pass = 1024 / a.binrows;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if (pass * 1024 < n) {   // uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) L.skeys[i * 256 + tid] = key[pass * 4 + i];
            __syncthreads();
            const int b = pass * bins_per_pass + tid;
            if (tid < bins_per_pass && b < cnt) {
                uint64_t best = 0;
                for (int r = 0; r < a.binrows; ++r) { const uint64_t o = L.skeys[tid * a.binrows + r]; best = o > best ? o : best; }
                L.bb[b] = best;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // 2. t2 = k-th largest bin best (stays 0 = "keep everything" when there are fewer than k bins)
    if (tid < cnt) {
        const uint64_t mine = L.bb[tid];
        int r = 0;
        for (int j = 0; j < cnt; ++j) r += L.bb[j] > mine ? 1 : 0;
        if (r == a.k - 1 && mine != 0) L.t2 = mine;
    }
    __syncthreads();
    const uint64_t t2 = L.t2;
    // 3. survivors
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (key[i] != 0 && key[i] >= t2) {
            const int pos = atomicAdd(&L.ns, 1);
            if (pos < 1024) L.skeys[pos] = key[i];
        }
    __syncthreads();
    const int ns = L.ns;
    int have = 0;
    uint64_t kth = 0;
    if (ns <= 1024) {
        // 4. rank the survivors against each other (keys are unique), write by rank
        for (int s = tid; s < ns; s += 256) {
            const uint64_t mine = L.skeys[s];
            int r = 0;
#pragma unroll 4
            for (int j = 0; j < ns; ++j) r += L.skeys[j] > mine ? 1 : 0;
            if (r < a.k) {
                rq_emit(a, r, mine);
                if (r == kk - 1) L.skth = mine;
            }
        }
        __syncthreads();
        have = ns < a.k ? ns : a.k;
        kth = L.skth;
    } else {
        // massive ties: up to 4096 keys in registers, k rounds of "extract the maximum"
        uint64_t best = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) best = key[i] > best ? key[i] : best;
        for (int j = 0; j < a.k; ++j) {
            const uint64_t wm = rq_wave_max_u64(best);
            if (lane == 0) L.wbest[j & 1][wave] = wm;
            __syncthreads();
            uint64_t win = L.wbest[j & 1][0];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) win = L.wbest[j & 1][w2] > win ? L.wbest[j & 1][w2] : win;
            if (win == 0) break;   // uniform: candidates exhausted
            have = j + 1;
            if (j == kk - 1) kth = win;
            if (tid == 0) rq_emit(a, j, win);
            if (best == win) {   // keys are unique: exactly one thread owns the winner
                best = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) { if (key[i] == win) key[i] = 0; best = key[i] > best ? key[i] : best; }
            }
        }
    }
    for (int j = have + tid; j < a.k; j += 256) { a.out_scores[j] = 0.f; a.out_rows[j] = -1; if (a.out_keys) a.out_keys[j] = 0; }
    if (tid == 0) {
        // Certificate: every row outside the candidate bins has approximate score < T, hence exact score
        // < T + eps (unit-query units).  Exact iff that bound is strictly below the k-th exact score.
        int good;
        if (total > a.rmax || overflow) good = 0;                       // some candidate bin was not re-scored
        else if (T == -__builtin_huge_valf()) good = have >= kk;        // every bin was a candidate
        else if (have < kk || kth == 0) good = 0;
        else {
            const double bound = a.metric == 0 ? (double)T + (double)a.eps
                                               : ((double)T + (double)a.eps * (double)a.max_row_norm) * qn * (1.0 + 1e-6);
            good = (float)bound < rq_key_score(kth);
        }
        *a.out_status = good ? 0 : 1;
    }
}
