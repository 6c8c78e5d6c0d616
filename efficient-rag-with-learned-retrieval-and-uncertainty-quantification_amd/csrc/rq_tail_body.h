// rq_tail_body.h -- the work of one tail workgroup (see rq_tail.hip for the algorithm), shared by the stand-alone
// tail kernel (rq_tail.hip) and the fused scan + tail kernel (rq_scan.hip).
#pragma once
#include "rq_device.h"
#include "rq_kernels.h"
#include "rq_final_body.h"

#define RQ_TAIL_HITCAP 768     // candidate bins one workgroup can hold
#define RQ_TAIL_JOBCAP 2048    // row jobs one workgroup can hold

// LDS of one tail workgroup (30.8 KB: query 3 KB, hit / job lists 11 KB, 2048-key ranking buffer 16 KB): static in
// rq_tail_kernel, carved from the scan's LDS in the fused kernel.
struct __attribute__((aligned(16))) RqTailLds {
    float qs[RQ_DPAD];            // the raw query, shared by the four waves
    double qpart[4];
    float thr_s;
    int nhit_s, njob_s, base_s, last_s, total_s, ovf_s;
    int hits[RQ_TAIL_HITCAP];
    int jobs[RQ_TAIL_JOBCAP];     // (hit << 6) | position of the row inside its bin
    RqFinalLds flds;
};

template <int N> struct RqInt { static constexpr int value = N; };

// One tail workgroup: chunk `chunk` (of `nchunks`) of query `q`.  256 threads.
// NV: 16-byte loads (two bin records each) per thread: chunk = 512 * NV bins per workgroup
template <int NV>
__device__ __forceinline__ void rq_tail_body(const RqTailArgs& a, const int chunk, const int q, const int nchunks, RqTailLds& L) {
    float* const qs = L.qs;
    double* const qpart = L.qpart;
    int* const hits = L.hits;
    int* const jobs = L.jobs;
    float& thr_s = L.thr_s;
    // this query's error bound in unit-query units: the shard's (a.eps) alone, or -- int8 scan -- together with the query's own
    // measured quantisation error e_q: |approx - exact| <= e_q + (1 + e_q) e_rows  <=  e_q (1 + eps) + eps
    const float eps_q = a.qeps ? a.qeps[q] * (1.f + a.eps) + a.eps : a.eps;
    int &nhit_s = L.nhit_s, &njob_s = L.njob_s, &base_s = L.base_s, &last_s = L.last_s, &total_s = L.total_s, &ovf_s = L.ovf_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float NEG_INF = -__builtin_huge_valf();
    constexpr int binrows = RQ_BIN_ROWS;
    constexpr int CHUNK = 512 * NV;

    // ---- independent loads first: this chunk of bin records, the query (-> LDS), (wave 0) the partition maxima
    const uint2* p = a.bins + (int64_t)q * a.bins_stride;
    const int64_t cbase = (int64_t)chunk * CHUNK + tid * 2;
    uint4 v4[NV];   // two records each
    float2 be[NV];  // int8 scan: the two bins' worst row errors
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int64_t i = cbase + (int64_t)u * 512;
        v4[u] = (i < a.bins_stride) ? *(const uint4*)(p + i) : make_uint4(0u, 0u, 0u, 0u);   // bins_stride is even
        be[u] = (a.binerr && i < a.bins_stride) ? *(const float2*)(a.binerr + i) : make_float2(0.f, 0.f);
    }
    float qmine[3];
    {
        const float* qp = a.q + (size_t)q * a.dim;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) { const int i = pp * 256 + tid; qmine[pp] = i < a.dim ? qp[i] : 0.f; qs[i] = qmine[pp]; }
    }
    if (tid == 0) { thr_s = NEG_INF; nhit_s = 0; njob_s = 0; last_s = 0; }

    // ---- A. threshold.  P = the m-th largest partition maximum (m = the k rows wanted; partitions = groups of the
    //      scan's per-workgroup maxima, distinct workgroups own distinct bins), found by a ballot radix select in
    //      wave 0 and truncated to the top 20 key bits (a slightly lower value).  At least m rows have an approximate
    //      score >= P, so the k-th exact score is >= P - eps.  T = P - 2 eps - margin (thr_mult 2.25) therefore always satisfies the
    //      certificate T + eps < s_k: the candidate set adapts to how dense the scores are around the k-th one
    //      (about k + 2 rows on Gaussian data, hundreds inside a tight cluster) instead of failing there.
    //      64 partitions (one per lane) for m <= 8, 256 for m <= 64, else 512: more partitions = tighter P.
    if (wave == 0) {
        const float* w = a.wgmax + (int64_t)q * a.wgmax_stride;
        const int nwg = q >= a.nwg_split ? a.nwg2 : a.nwg;
        uint32_t prefix = 0;
        auto select = [&](auto npl_tag) {
            constexpr int NPL = decltype(npl_tag)::value;   // partitions per lane
            uint32_t key[NPL];
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                float v = NEG_INF;
                for (int j = i * 64 + lane; j < nwg; j += 64 * NPL) v = fmaxf(v, w[j]);
                key[i] = rq_mono32(v);
            }
            for (int bit = 31; bit >= 12; --bit) {
                const uint32_t t = prefix | (1u << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < NPL; ++i) c += __popcll(__ballot(key[i] >= t));
                if (c >= a.m) prefix = t;   // uniform
            }
        };
        if (a.m <= 8) select(RqInt<1>{});
        else if (a.m <= 64) select(RqInt<4>{});
        else select(RqInt<8>{});
        // prefix == 0 (fewer than m partitions hold anything): unmono gives NaN -> use -inf = "every bin"
        if (lane == 0) {
            float T0 = NEG_INF;
            if (prefix > rq_mono32(NEG_INF)) {
                const float P = rq_unmono32(prefix);
                const float e = a.metric == 0 ? eps_q : eps_q * a.max_row_norm;
                // T = P - e - (thr_mult - 1) max(e, thr_slack): thr_mult 2.25 certifies by construction (T + e < P - e <= s_k); the int8
                // scan runs with less, the slack then being measured against the bound of a typical one-image query (rq_api.hip)
                const float es = fmaxf(e, a.metric == 0 ? a.thr_slack : a.thr_slack * a.max_row_norm);
                T0 = P - e - (a.thr_mult - 1.f) * es - 4e-6f * fabsf(P);   // margin: the 1e-6 relative slack of the certificate and fp32 rounding
            }
            thr_s = T0;
        }
    }
    __syncthreads();
    const float T = thr_s;
    if (a.stop_after == 1) { if (tid == 0 && chunk == 0) a.out_status[q] = (int)T; return; }

    // ---- B. bins of this chunk that reach the threshold -> row jobs: the arg-max row alone when the bin's second
    //      largest score is below T, the two largest when only the third is, all 64 rows otherwise
    {
        // int8 scan: |approx - exact| <= e_q + (1 + e_q) e_rows for the rows of a bin, e_rows <= the bin's worst row <= the shard's.  T was
        // set with the shard's worst row; a bin whose rows quantise better is tested against Tb = T + (1 + e_q)(shard - bin): a row left
        // out there has approx < Tb, hence exact < Tb + e_q + (1 + e_q) bin = T + eps_q -- the same certificate (rq_final_body).
        const float lift = a.binerr ? (1.f + a.qeps[q]) * (a.metric == 0 ? 1.f : a.max_row_norm) * 0.9999f : 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const uint32_t rx[2] = {v4[u].x, v4[u].z}, ry[2] = {v4[u].y, v4[u].w};
            const float eb[2] = {be[u].x, be[u].y};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t i = cbase + (int64_t)u * 512 + e;
                const float T = a.binerr ? thr_s + lift * fmaxf(a.eps_rows_max - eb[e] * 1.000001f, 0.f) : thr_s;
                if (i < a.nbins && rq_rec_m1(rx[e]) >= T) {
                    const int h = atomicAdd(&nhit_s, 1);
                    if (h < RQ_TAIL_HITCAP) {
                        hits[h] = (int)i;
                        const uint32_t c2 = ry[e] >> 16, d = (ry[e] >> 6) & 1023u;
                        const bool two = rq_code16_value(c2) >= T, whole = two && rq_code16_value(c2 - d) >= T;
                        const int nj = whole ? binrows : (two ? 2 : 1);
                        const int j0 = atomicAdd(&njob_s, nj);
                        const int a1 = (int)(rx[e] & 63u), a2 = (int)(ry[e] & 63u);
                        for (int j = 0; j < nj; ++j)
                            if (j0 + j < RQ_TAIL_JOBCAP) jobs[j0 + j] = (h << 6) | (whole ? j : (j ? a2 : a1));
                    }
                }
            }
        }
    }
    __syncthreads();
    const int nh = nhit_s;
    const int njob_all = njob_s;
    const int njob = njob_all < RQ_TAIL_JOBCAP ? njob_all : RQ_TAIL_JOBCAP;
    // A row that is not among the k best of THIS workgroup's candidates cannot be among the k best of the query: with more
    // than k jobs the exact keys stay in LDS, the workgroup selects its own k best and publishes only those.  The query's
    // list then holds at most k keys per workgroup (the final ranks 40 keys instead of the 60-500 the int8 scan's looser
    // bound collects at k = 10, and lists that used to overflow RQ_CAND_CAP -- clustered corpora -- no longer do).
    const bool local = a.local_topk && njob_all > a.k && njob_all <= RQ_TAIL_JOBCAP;
    if (tid == 0) {
        base_s = njob_all ? atomicAdd(&a.rowcount[q], local ? a.k : njob_all) : 0;
        if (nh > RQ_TAIL_HITCAP || njob_all > RQ_TAIL_JOBCAP) atomicOr(&a.ovf[q], 1);
    }
    // fp64 norm of the query (only workgroups with jobs, and later the last one, need it)
    auto query_norm = [&]() -> double {
        double ssq = 0.0;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) ssq += (double)qmine[pp] * (double)qmine[pp];
        ssq = rq_wave_sum(ssq);
        if (lane == 0) qpart[wave] = ssq;
        __syncthreads();
        const double r = sqrt((qpart[0] + qpart[1]) + (qpart[2] + qpart[3]));
        __syncthreads();
        return r;
    };
    double qn = -1.0;
    if (nh > 0) qn = query_norm();   // uniform branch (nh comes from LDS after a barrier); contains barriers, which
                                     // also make base_s (thread 0's returned atomic) visible to every wave
    else __syncthreads();
    if (a.stop_after == 2) return;

    // ---- C. exact re-score of the job rows: 16 lanes per row (sub = lane & 15 owns elements pp*128 + 8*sub + e),
    //      a wave takes 8 jobs per round (two groups of 4 rows, all 12 loads per lane in flight before the math)
    if (nh > 0 && qn != 0.0) {
        const int base = base_s;
        const int sub = lane & 15, rloc = lane >> 4;
        const char* xb = (const char*)a.x;
        uint64_t* out = a.cand + (int64_t)q * RQ_CAND_CAP;
        for (int j0 = wave * 8; j0 < njob; j0 += 32) {
            rq_half8 xv[2][6];
            int64_t rows[2];
            double rn[2];
            int pos[2];
            bool lv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jb = j0 + u * 4 + rloc;
                const bool live = jb < njob;
                lv[u] = live;
                const int job = jobs[live ? jb : 0];
                pos[u] = (live && base + jb < RQ_CAND_CAP) ? base + jb : -1;   // -1: nothing stored (padding or list full)
                rows[u] = (int64_t)hits[job >> 6] * RQ_BIN_ROWS + (job & 63);
                const int64_t rr = rows[u] < a.n_rows ? rows[u] : 0;
                rn[u] = a.rownorm64[rr];
                const char* r = xb + rr * (RQ_DPAD * 2) + sub * 16;
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) xv[u][pp] = *(const rq_half8*)(r + pp * 256);
            }
            double dot[2] = {0.0, 0.0};
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) {
                const float4 qlo = *(const float4*)&qs[pp * 128 + 8 * sub], qhi = *(const float4*)&qs[pp * 128 + 8 * sub + 4];
                const float qq[8] = {qlo.x, qlo.y, qlo.z, qlo.w, qhi.x, qhi.y, qhi.z, qhi.w};
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) dot[u] += (double)qq[e] * (double)(float)xv[u][pp][e];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                double d = dot[u];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
                if (sub == 0 && (local ? lv[u] : pos[u] >= 0)) {
                    uint64_t key = 0;                                       // stays 0 for a row beyond the shard's end
                    if (rows[u] < a.n_rows) {
                        double sc = d;
                        if (a.metric == 0) sc = d / (qn * rn[u] + 1e-30);
                        key = rq_make_key(rq_sanitize((float)sc), (uint32_t)rows[u]);
                    }
                    // write-through (sc1) store: visible to the last workgroup without a release fence
                    if (local) L.flds.skeys[j0 + u * 4 + rloc] = key;
                    else __hip_atomic_store(&out[pos[u]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (local) {   // uniform over the workgroup
            __syncthreads();
            RqFinalLds& F = L.flds;
            const int want = a.k;
            int have;
            if (njob <= 256) {
                // one key per thread, ranked against all others through LDS; non-empty keys are unique (they carry their row)
                const uint64_t mine = tid < njob ? F.skeys[tid] : 0;
                if (tid == 0) F.snz = 0;
                __syncthreads();
                int r = 0;
                if (mine != 0) {
#pragma unroll 4
                    for (int j = 0; j < njob; ++j) r += F.skeys[j] > mine ? 1 : 0;
                    atomicAdd(&F.snz, 1);
                    if (r < want && base + r < RQ_CAND_CAP) __hip_atomic_store(&out[base + r], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                have = F.snz < want ? F.snz : want;
            } else {
                uint64_t key[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { const int j = i * 256 + tid; key[i] = j < njob ? F.skeys[j] : 0; }
                __syncthreads();                                  // every key is in registers before the select re-uses the buffer
                have = rq_select_winners(key, want, F);
                for (int j = tid; j < have; j += 256)
                    if (base + j < RQ_CAND_CAP) __hip_atomic_store(&out[base + j], F.skeys[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the slots this workgroup reserved but has no key for (rows beyond the shard's end among its jobs): empty
            for (int j = have + tid; j < want; j += 256)
                if (base + j < RQ_CAND_CAP) __hip_atomic_store(&out[base + j], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.stop_after == 3) return;

    // ---- D. publish.  Producer side (every workgroup): all candidate keys were stored write-through (sc1), every
    //      storing wave drains vmcnt(0), workgroup barrier, ONE lane draws the ticket (agent-scope atomic add).
    //      Consumer side (the workgroup whose ticket is the query's last): ONE agent-scope acquire by the lane that drew
    //      the ticket, s_waitcnt vmcnt(0) so that the barrier below is held until the L1 invalidate has completed, workgroup
    //      barrier, then the loads (which stay sc1 as well).  This is MI355X_MICROARCH.md "Valid forms", Consumer bullet,
    //      in its unconditional form: several tail workgroups share a CU here (and, fused, sit beside two scan workgroups),
    //      which is outside the "one per CU" cell the sc1-loads-only form was measured for.  Cost: one buffer_inv per QUERY
    //      (64 per launch, in parallel), not per workgroup.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&a.done[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == nchunks - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            total_s = __hip_atomic_load(&a.rowcount[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ovf_s = __hip_atomic_load(&a.ovf[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // counters back to zero for the next launch on this workspace
            __hip_atomic_store(&a.rowcount[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.ovf[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.done[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = 1;
        }
    }
    __syncthreads();
    if (!last_s || a.stop_after == 4) return;
    if (a.stop_after == 5) { if (tid == 0) a.out_status[q] = total_s; return; }   // development: candidate rows of the query
    if (qn < 0.0) qn = query_norm();   // uniform: this workgroup had no hits of its own

    RqFinalCore c;
    c.cand = a.cand + (int64_t)q * RQ_CAND_CAP; c.metric = a.metric; c.eps = eps_q;
    c.max_row_norm = a.max_row_norm; c.k = a.k; c.row_offset = a.row_offset; c.n_rows = a.n_rows;
    c.out_scores = a.out_scores + (int64_t)q * a.k; c.out_rows = a.out_rows + (int64_t)q * a.k;
    c.out_keys = a.out_keys ? a.out_keys + (int64_t)q * a.k : nullptr; c.out_status = a.out_status + q;
    rq_final_body(c, total_s, ovf_s, T, qn, L.flds);
}
