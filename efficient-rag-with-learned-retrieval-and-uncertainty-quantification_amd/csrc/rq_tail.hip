// rq_tail.hip -- everything after the corpus scan in ONE launch (the common case: <= 128 bins, k <= 128).
//
// grid (chunks of 1024 bins, B queries), 256 threads.  Each workgroup
//   A. derives the threshold T = m-th largest of 256 partition maxima (each the max of a few per-workgroup
//      maxima written by the scan; valid because distinct scan workgroups own distinct bins, so at least m
//      bins reach T) with a ballot radix select in wave 0, and the fp64 norm of its query,
//   B. finds the bins of its chunk with pooled >= T (one float4 per thread),
//   C. re-scores every row of those bins exactly in fp64: one wave per bin, 16 lanes per row, all loads of
//      the bin's 4R rows in flight before the math, 4-step reductions; keys go to the query's candidate
//      list with 8-byte write-through (sc1) stores,
//   D. publishes: every wave drains vmcnt, workgroup barrier, ONE lane draws a ticket (agent-scope atomic
//      add).  The workgroup that draws the last ticket of its query reads the keys with sc1 loads only,
//      selects the exact top-k, evaluates the certificate and resets the counters.  No fences: a release
//      fence per workgroup serialises on the L2 write-back (measured +60 us per launch).
// Replaces reference rag_uq/streaming_index.py:355-368 (collection.query + `1 - distance`) after the scan.
#include "rq_device.h"
#include "rq_kernels.h"
#include "rq_final_body.h"

#define RQ_TAIL_LOCALCAP 64

// NV4: float4 loads of pooled values per thread (chunk = 1024 * NV4 bins per workgroup)
template <int R, int NV4>
__global__ __launch_bounds__(256) void rq_tail_kernel(RqTailArgs a) {   // 168 VGPRs; capping at 128 spills and costs 30 % (measured)
    __shared__ __attribute__((aligned(16))) float qs[RQ_DPAD];   // the raw query, shared by the four waves
    __shared__ double qpart[4];
    __shared__ float thr_s;
    __shared__ int nhit_s, base_s, last_s, total_s, ovf_s;
    __shared__ int hits[RQ_TAIL_LOCALCAP];
    __shared__ RqFinalLds flds;
    const int q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float NEG_INF = -__builtin_huge_valf();
    constexpr int binrows = 4 * R;
    constexpr int CHUNK = 1024 * NV4;

    // ---- independent loads first: this chunk of pooled values, the query (-> LDS), (wave 0) the partition maxima
    const float* p = a.pooled + (int64_t)q * a.pooled_stride;
    const int64_t cbase = (int64_t)blockIdx.x * CHUNK + tid * 4;
    float4 v4[NV4];
#pragma unroll
    for (int u = 0; u < NV4; ++u) {
        const int64_t i = cbase + (int64_t)u * 1024;
        v4[u] = (i < a.pooled_stride) ? *(const float4*)(p + i) : make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
    }
    float qmine[3];
    {
        const float* qp = a.q + (size_t)q * a.dim;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) { const int i = pp * 256 + tid; qmine[pp] = i < a.dim ? qp[i] : 0.f; qs[i] = qmine[pp]; }
    }
    if (tid == 0) { thr_s = NEG_INF; nhit_s = 0; last_s = 0; }

    // ---- A. threshold: ballot radix select (wave 0) of the m-th largest partition maximum, truncated to the
    //      top 20 key bits (a slightly lower, still valid threshold).  m <= 24: 64 partitions (one per lane),
    //      else 256 (four per lane): more partitions = tighter threshold when m is large.
    if (wave == 0) {
        const float* w = a.wgmax + (int64_t)q * a.wgmax_stride;
        uint32_t prefix = 0;
        if (a.m <= 24) {
            float v = NEG_INF;
            for (int j = lane; j < a.nwg; j += 64) v = fmaxf(v, w[j]);
            const uint32_t key = rq_mono32(v);
            for (int bit = 31; bit >= 12; --bit) {
                const uint32_t t = prefix | (1u << bit);
                if (__popcll(__ballot(key >= t)) >= a.m) prefix = t;   // uniform
            }
        } else if (a.m <= 256) {
            uint32_t key[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = NEG_INF;
                for (int j = i * 64 + lane; j < a.nwg; j += 256) v = fmaxf(v, w[j]);
                key[i] = rq_mono32(v);
            }
            for (int bit = 31; bit >= 12; --bit) {
                const uint32_t t = prefix | (1u << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) c += __popcll(__ballot(key[i] >= t));
                if (c >= a.m) prefix = t;
            }
        }
        // prefix == 0 (fewer than m partitions hold anything): unmono gives NaN -> use -inf = "every bin"
        if (lane == 0) thr_s = prefix > rq_mono32(NEG_INF) ? rq_unmono32(prefix) : NEG_INF;
    }
    __syncthreads();
    const float T = thr_s;
    if (a.stop_after == 1) { if (tid == 0 && blockIdx.x == 0) a.out_status[q] = (int)T; return; }

    // ---- B. bins of this chunk that reach the threshold
#pragma unroll
    for (int u = 0; u < NV4; ++u) {
        const float xs[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = cbase + (int64_t)u * 1024 + e;
            if (i < a.nbins && xs[e] >= T) {
                const int h = atomicAdd(&nhit_s, 1);
                if (h < RQ_TAIL_LOCALCAP) hits[h] = (int)i;
            }
        }
    }
    __syncthreads();
    const int nh = nhit_s;
    const int nloc = nh < RQ_TAIL_LOCALCAP ? nh : RQ_TAIL_LOCALCAP;
    if (tid == 0) {
        base_s = nh ? atomicAdd(&a.bincount[q], nh) : 0;
        if (nh > RQ_TAIL_LOCALCAP) atomicOr(&a.ovf[q], 1);
    }
    // fp64 norm of the query (only workgroups with hits, and later the last one, need it)
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

    // ---- C. exact re-score.  Task = (hit, group of 4 rows); 16 lanes per row (sub = lane & 15 owns elements
    //      pp*128 + 8*sub + e).  Every wave takes two tasks per round (8 rows of loads in flight), so a
    //      workgroup clears 2 hits of 16 rows per round.
    if (nh > 0 && qn != 0.0) {
        const int base = base_s;
        const int sub = lane & 15, rloc = lane >> 4;
        const char* xb = (const char*)a.x;
        const int ntask = nloc * R;
        for (int t0 = wave * 2; t0 < ntask; t0 += 8) {
            rq_half8 xv[2][6];
            int64_t rows[2];
            double rn[2];
            int slot[2], grp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = t0 + u < ntask ? t0 + u : t0;   // odd tail: repeat the first task, store is skipped
                const int h = t / R;
                grp[u] = t % R;
                slot[u] = (t0 + u < ntask) ? base + h : a.rmax;   // slot >= rmax: nothing is stored
                rows[u] = rq_bin_row((int64_t)hits[h], R, grp[u] * 4 + rloc);
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
                if (sub == 0 && slot[u] < a.rmax) {     // slot >= rmax: the last workgroup sees total > rmax = overflow
                    uint64_t key = 0;
                    if (rows[u] < a.n_rows) {
                        double sc = d;
                        if (a.metric == 0) sc = d / (qn * rn[u] + 1e-30);
                        key = rq_make_key(rq_sanitize((float)sc), (uint32_t)rows[u]);
                    }
                    // write-through (sc1) store: visible to the last workgroup without a release fence
                    uint64_t* out = a.cand + ((int64_t)q * a.rmax + slot[u]) * binrows;
                    __hip_atomic_store(&out[grp[u] * 4 + rloc], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if (a.stop_after == 3) return;

    // ---- D. publish (MI355X_MICROARCH.md "Valid forms", first table row)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&a.done[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (int)gridDim.x - 1) {
            total_s = __hip_atomic_load(&a.bincount[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ovf_s = __hip_atomic_load(&a.ovf[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // counters back to zero for the next launch on this workspace
            __hip_atomic_store(&a.bincount[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.ovf[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.done[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = 1;
        }
    }
    __syncthreads();
    if (!last_s || a.stop_after == 4) return;
    if (qn < 0.0) qn = query_norm();   // uniform: this workgroup had no hits of its own

    RqFinalCore c;
    c.cand = a.cand + (int64_t)q * a.rmax * binrows; c.rmax = a.rmax; c.binrows = binrows; c.metric = a.metric; c.eps = a.eps;
    c.max_row_norm = a.max_row_norm; c.k = a.k; c.row_offset = a.row_offset; c.n_rows = a.n_rows;
    c.out_scores = a.out_scores + (int64_t)q * a.k; c.out_rows = a.out_rows + (int64_t)q * a.k;
    c.out_keys = a.out_keys ? a.out_keys + (int64_t)q * a.k : nullptr; c.out_status = a.out_status + q;
    rq_final_body<true>(c, total_s, ovf_s, T, qn, flds);
}

template <int NV4>
static hipError_t rq_tail_launch_nv(const RqTailArgs& a, int B, hipStream_t stream) {
    const int64_t chunks = (a.nbins + 1024 * NV4 - 1) / (1024 * NV4);
    if (chunks < 1 || chunks > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)chunks, B);
    switch (a.R) {
        case 4: hipLaunchKernelGGL((rq_tail_kernel<4, NV4>), grid, dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((rq_tail_kernel<2, NV4>), grid, dim3(256), 0, stream, a); break;
        case 1: hipLaunchKernelGGL((rq_tail_kernel<1, NV4>), grid, dim3(256), 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t rq_tail_launch(const RqTailArgs& a, int B, hipStream_t stream) {
    if (a.m < 1 || a.rmax < 1 || a.rmax > RQ_FAST_MAX_BINS || a.k < 1 || a.k > RQ_FAST_MAX_K || a.rmax * 4 * a.R > 4096)
        return hipErrorInvalidValue;
    // about a thousand workgroups: enough to spread the hits, few enough to be one dispatch round
    const int64_t wgs1 = ((a.nbins + 1023) / 1024) * B;
    return wgs1 <= 1536 ? rq_tail_launch_nv<1>(a, B, stream) : rq_tail_launch_nv<4>(a, B, stream);
}
