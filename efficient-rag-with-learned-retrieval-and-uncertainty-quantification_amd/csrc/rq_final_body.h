// rq_final_body.h -- final top-k of one query's candidate keys + exactness certificate.
// Run by all 256 threads of the workgroup that drew the last ticket of its query in rq_tail_kernel (rq_tail.hip).
//
// The candidates are a compact list of n keys (typically k..3k: one exact re-scored row per candidate bin).
//   n <= 2048: every key is ranked against all others through LDS (keys are unique) and written by rank;
//   n <= 4096: keys stay in registers, k rounds of "extract the maximum" (massive exact ties only).
#pragma once
#include "rq_device.h"
#include "rq_kernels.h"   // RQ_CAND_CAP

// The score definition divides by (|q| |x| + 1e-30); the scan scores the pure cosine.  A stored fp16 row has norm 0 or
// >= 6e-8, so for |q| >= 2e-17 the two differ by less than 1e-6 relative on every row; a query below that is never
// certified from the approximate pass and takes the exact fp64 route (rq_search_fixup_device), which follows the definition.
#define RQ_TINY_QUERY_NORM 2e-17

struct RqFinalLds {
    uint64_t skeys[2048];
    uint64_t wbest[2][4];
    uint64_t skth;
    int snz;
};

struct RqFinalCore {
    const uint64_t* cand;   // this query's candidate keys
    int metric, k;
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

// total: keys that wanted a slot (may exceed RQ_CAND_CAP = overflow); overflow: any other overflow seen for this query;
// T: every row that was NOT re-scored has approximate score < T;  qn: fp64 norm of the query.
// The keys were published by other workgroups of the SAME launch with sc1 (write-through) stores and a ticket; the caller
// (rq_tail_body.h, section D) has run an agent-scope acquire + workgroup barrier before this point, and every load of the
// keys is an sc1 load (relaxed agent-scope atomic load) besides.
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
    const int n = total < RQ_CAND_CAP ? total : RQ_CAND_CAP;
    int have = 0;
    uint64_t kth = 0;
    if (tid == 0) { L.skth = 0; L.snz = 0; }
    if (n <= 2048) {
        int nz = 0;
        for (int j = tid; j < n; j += 256) L.skeys[j] = __hip_atomic_load(a.cand + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (int s = tid; s < n; s += 256) {
            const uint64_t mine = L.skeys[s];
            if (mine == 0) continue;           // a row beyond the shard's end
            nz++;
            int r = 0;
#pragma unroll 4
            for (int j = 0; j < n; ++j) r += L.skeys[j] > mine ? 1 : 0;
            if (r < a.k) {
                rq_emit(a, r, mine);
                if (r == kk - 1) L.skth = mine;
            }
        }
        if (nz) atomicAdd(&L.snz, nz);   // number of non-empty keys
        __syncthreads();                 // all emits done, skth and snz complete
        have = L.snz < a.k ? L.snz : a.k;
        kth = L.skth;
    } else {
        uint64_t key[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i * 256 + tid;
            key[i] = j < n ? __hip_atomic_load(a.cand + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
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
        // Certificate: every row that was not re-scored has approximate score < T, hence exact score < T + eps
        // (unit-query units).  Exact iff that bound is strictly below the k-th exact score.
        int good;
        if (total > RQ_CAND_CAP || overflow) good = 0;                  // some candidate row was not re-scored
        else if (a.metric == 0 && qn < RQ_TINY_QUERY_NORM) good = 0;                     // the 1e-30 of the score definition is no longer negligible
        else if (T == -__builtin_huge_valf()) good = have >= kk;        // every row was a candidate
        else if (have < kk || kth == 0) good = 0;
        else {
            const double bound = a.metric == 0 ? (double)T + (double)a.eps
                                               : ((double)T + (double)a.eps * (double)a.max_row_norm) * qn * (1.0 + 1e-6);
            good = (float)bound < rq_key_score(kth);
        }
        *a.out_status = good ? 0 : 1;
    }
}
