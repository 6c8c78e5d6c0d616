// rq_final_body.h -- final top-k of one query's candidate keys + exactness certificate.
// Run by all 256 threads of the workgroup that drew the last ticket of its query in rq_tail_kernel (rq_tail.hip).
//
// The candidates are a compact list of n keys (typically k..3k: one exact re-scored row per candidate bin).
//   n <= 512:  every key is ranked against all others through LDS (keys are unique) and written by rank;
//   n <= 4096: keys stay in registers (16 per thread); a 64-round radix select over the whole workgroup finds the k-th
//              largest key (per round: 16 ballots per wave, one LDS word per wave, one barrier), the k winners are
//              compacted into LDS and ranked among themselves.  Document-structured corpora (16 consecutive similar
//              passages) re-score whole 64-row bins and reach 1500-3000 candidates per query at k = 100: there the
//              earlier forms (O(n^2) ranking up to 2048 keys, k rounds of extract-the-maximum beyond) took 260 us of a
//              350 us tail; the select takes under 10 (tools/gpu_tail_docs.py).
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

// The `want` largest of up to 4096 keys held in registers (16 per thread, 0 = empty; non-empty keys are unique), all 256
// threads: a radix select over the whole workgroup (per round 16 ballots per wave, one LDS word per wave, one barrier).  The
// keys of one query share their leading bits (scores within a few percent of each other): those rounds are skipped -- one
// reduction of the keys' OR and AND finds the first bit they differ in -- and the select stops at the first threshold that
// exactly `have` keys reach.  Returns have = min(want, non-empty keys); L.skeys[0 .. have) then hold the winners, unordered.
__device__ __forceinline__ int rq_select_winners(const uint64_t (&key)[16], const int want, RqFinalLds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int round = 0;
    auto count_ge = [&](uint64_t t) -> int {
        int c = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c += __popcll(__ballot(key[i] >= t));
        if (lane == 0) L.wbest[round & 1][wave] = (uint64_t)c;
        __syncthreads();
        const int tot = (int)(L.wbest[round & 1][0] + L.wbest[round & 1][1] + L.wbest[round & 1][2] + L.wbest[round & 1][3]);
        ++round;
        return tot;   // uniform over the workgroup
    };
    const int nzt = count_ge(1);                         // non-empty keys
    const int have = nzt < want ? nzt : want;
    uint64_t prefix = 1;                                 // have == nzt: every non-empty key is a winner
    if (nzt > have) {                                    // the have-th largest key: the largest t with count(key >= t) >= have
        uint64_t kor = 0, kand = ~0ull;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (key[i] != 0) { kor |= key[i]; kand &= key[i]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { kor |= __shfl_xor(kor, off, 64); kand &= __shfl_xor(kand, off, 64); }
        if (lane == 0) { L.wbest[round & 1][wave] = kor; L.skeys[wave] = kand; }
        __syncthreads();
        kor = L.wbest[round & 1][0] | L.wbest[round & 1][1] | L.wbest[round & 1][2] | L.wbest[round & 1][3];
        kand = L.skeys[0] & L.skeys[1] & L.skeys[2] & L.skeys[3];
        ++round;
        const uint64_t diff = kor ^ kand;                // != 0: the keys are unique and there are at least two
        const int top = 63 - __builtin_clzll(diff | 1ull);   // highest bit in which two keys differ
        prefix = top == 63 ? 0 : (kand >> (top + 1)) << (top + 1);
        for (int bit = top; bit >= 0; --bit) {
            const uint64_t t = prefix | (1ull << bit);
            const int c = count_ge(t);                   // uniform
            if (c >= have) {
                prefix = t;
                if (c == have) break;                    // exactly the winners are >= t: the remaining bits cannot change the set
            }
        }
    }
    // keys are unique, so exactly `have` keys are >= prefix
    if (tid == 0) L.snz = 0;
    __syncthreads();                                     // (also: every thread is past its reads of L.skeys[0..3] above)
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (key[i] != 0 && key[i] >= prefix) L.skeys[atomicAdd(&L.snz, 1)] = key[i];
    __syncthreads();
    return have;
}

// total: keys that wanted a slot (may exceed RQ_CAND_CAP = overflow); overflow: any other overflow seen for this query;
// T: every row that was NOT re-scored has approximate score < T;  qn: fp64 norm of the query.
// The keys were published by other workgroups of the SAME launch with sc1 (write-through) stores and a ticket; the caller
// (rq_tail_body.h, section D) has run an agent-scope acquire + workgroup barrier before this point, and every load of the
// keys is an sc1 load (relaxed agent-scope atomic load) besides.
__device__ __forceinline__ void rq_final_body(const RqFinalCore& a, int total, int overflow, float T, double qn, RqFinalLds& L) {
    const int tid = threadIdx.x;
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
    if (n <= 512) {
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
        have = rq_select_winners(key, a.k, L);           // winners compacted into L.skeys[0 .. have): rank them among themselves
        for (int s2 = tid; s2 < have; s2 += 256) {
            const uint64_t mine = L.skeys[s2];
            int r = 0;
            for (int j = 0; j < have; ++j) r += L.skeys[j] > mine ? 1 : 0;
            rq_emit(a, r, mine);
            if (r == kk - 1) L.skth = mine;
        }
        __syncthreads();
        kth = L.skth;
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
