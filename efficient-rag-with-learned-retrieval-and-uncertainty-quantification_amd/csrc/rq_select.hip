// rq_select.hip -- everything of the search that is not the corpus scan:
//   row statistics at add time, query preparation, bin selection (pass 2), exact fp64 re-score of the
//   selected bins (pass 3), final top-k + exactness certificate (pass 4), cross-shard key merge.
//
// Result contract restated from reference rag_uq/streaming_index.py:355-368 (best first,
// score = 1 - cosine distance) with the tie rule fixed by the oracle (oracle/dense_oracle.py):
// order = (fp32 score desc, row asc).
#include "rq_device.h"
#include "rq_kernels.h"
#include "rq_final_body.h"   // RQ_TINY_QUERY_NORM

// --------------------------------------------------------------------------------------------
// small helpers
// --------------------------------------------------------------------------------------------
// rq_wave_sum / rq_sanitize / rq_wave_max_u64: rq_device.h

// --------------------------------------------------------------------------------------------
// row norms: one wave per row, lane l owns elements p*256 + 4*l + e (p < 3, e < 4).
// The matrix cores flush fp16 SUBNORMAL inputs to zero (measured: tests/test_gpu_parity.py, subnormal rows score 0 in the
// scan), so the scan's score of a row misses the products of its subnormal elements: at most
// sqrt(sum of their squares) * |q| by Cauchy-Schwarz.  stats[1] / stats[2] keep the shard's largest such mass relative to
// the row norm (cosine) and absolute (inner product); rq_api.hip adds them to the certificate's error bound.
// stats[0] = largest row norm.  (Bits of non-negative doubles order as integers.)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rq_rownorm_kernel(const char* x, int64_t row_begin, int64_t row_end, double* norm64,
                                                         unsigned long long* stats) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    double mx_norm = 0.0, mx_rel = 0.0, mx_abs = 0.0;
    for (int64_t row = row_begin + wave; row < row_end; row += nwaves) {
        const char* r = x + row * (RQ_DPAD * 2);
        double acc = 0.0, sub = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const rq_half4 v = *(const rq_half4*)(r + p * 512 + lane * 8);
            const ushort4 b = *(const ushort4*)(r + p * 512 + lane * 8);
            const unsigned short bits[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = (double)(float)v[e];
                acc += d * d;
                if ((bits[e] & 0x7c00u) == 0) sub += d * d;   // exponent field 0: zero or subnormal
            }
        }
        acc = rq_wave_sum(acc);
        sub = rq_wave_sum(sub);
        const double nrm = sqrt(acc);
        if (lane == 0) norm64[row] = nrm;
        if (nrm == nrm && nrm <= 1.7e308) {   // (rows with non-finite elements never certify anyway)
            mx_norm = fmax(mx_norm, nrm);
            if (sub > 0.0) { mx_abs = fmax(mx_abs, sqrt(sub)); mx_rel = fmax(mx_rel, sqrt(sub / acc)); }
        }
    }
    if (lane == 0) {
        atomicMax(&stats[0], (unsigned long long)__double_as_longlong(mx_norm));
        if (mx_rel > 0.0) atomicMax(&stats[1], (unsigned long long)__double_as_longlong(mx_rel));
        if (mx_abs > 0.0) atomicMax(&stats[2], (unsigned long long)__double_as_longlong(mx_abs));
    }
}
hipError_t rq_rownorm_launch(const void* x, int64_t row_begin, int64_t row_end, double* norm64, unsigned long long* stats, hipStream_t stream) {
    if (row_end <= row_begin) return hipSuccess;
    int64_t rows = row_end - row_begin;
    int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    hipLaunchKernelGGL(rq_rownorm_kernel, dim3(grid), dim3(256), 0, stream, (const char*)x, row_begin, row_end, norm64, stats);
    return hipGetLastError();
}

// The scan multiplies its accumulator by row_scale.  Queries reach the matrix cores as fp16(q/|q| * 2^12) (see
// rq_prep_queries_kernel), so both scale arrays carry the factor 2^-12: exact powers of two, the product is unchanged.
__global__ void rq_rowscale_kernel(const double* norm64, int64_t row_begin, int64_t row_end, float* inv_norm) {
    const int64_t i = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < row_end) { const double n = norm64[i]; inv_norm[i] = n > 0.0 ? (float)(1.0 / n) * RQ_QSCALE_INV : 0.f; }
}
hipError_t rq_rowscale_launch(const double* norm64, int64_t row_begin, int64_t row_end, float* inv_norm, hipStream_t stream) {
    if (row_end <= row_begin) return hipSuccess;
    const int64_t rows = row_end - row_begin;
    hipLaunchKernelGGL(rq_rowscale_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, norm64, row_begin, row_end, inv_norm);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// fp32 -> fp16 rows (optionally unit-normalised with the fp64 norm), fp16 -> padded fp16 rows
// one wave per row; source row has `dim` elements, destination RQ_DPAD (zero padded)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rq_convert_f32_kernel(const float* src, int dim, int64_t n, int normalize, _Float16* dst) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < n; row += nwaves) {
        const float* s = src + row * dim;
        float v[12];
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = p * 256 + 4 * lane + e;
                v[p * 4 + e] = i < dim ? s[i] : 0.f;
                const double d = (double)v[p * 4 + e];
                acc += d * d;
            }
        double nrm = 1.0;
        if (normalize) { acc = rq_wave_sum(acc); nrm = sqrt(acc); if (!(nrm > 0.0)) nrm = 1.0; }
        _Float16* d = dst + row * RQ_DPAD;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            rq_half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = normalize ? (float)((double)v[p * 4 + e] / nrm) : v[p * 4 + e];
                o[e] = (_Float16)f;
            }
            *(rq_half4*)(d + p * 256 + 4 * lane) = o;
        }
    }
}
hipError_t rq_convert_f32_launch(const float* src, int dim, int64_t n, int normalize, void* dst, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096);
    hipLaunchKernelGGL(rq_convert_f32_kernel, dim3(grid), dim3(256), 0, stream, src, dim, n, normalize, (_Float16*)dst);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void rq_pad_f16_kernel(const _Float16* src, int dim, int64_t n, _Float16* dst) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < n; row += nwaves) {
        const _Float16* s = src + row * dim;
        _Float16* d = dst + row * RQ_DPAD;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            rq_half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int i = p * 256 + 4 * lane + e; o[e] = i < dim ? s[i] : (_Float16)0.f; }
            *(rq_half4*)(d + p * 256 + 4 * lane) = o;
        }
    }
}
hipError_t rq_pad_f16_launch(const void* src, int dim, int64_t n, void* dst, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096);
    hipLaunchKernelGGL(rq_pad_f16_kernel, dim3(grid), dim3(256), 0, stream, (const _Float16*)src, dim, n, (_Float16*)dst);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// query preparation: one wave-sized pass per query (block = 256 threads, thread t owns 3 elements)
// qh = fp16(q / |q| * 2^12): |element| <= 4096, and an element is flushed by the matrix cores (fp16 subnormal) only below
// 2^-26 of the unit query -- at most sqrt(768) * 2^-26 = 4e-7 of score error, against 1.7e-3 without the scaling.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rq_prep_queries_kernel(RqPrepArgs a) {
    __shared__ double part[16];
    rq_prep_body(a, (int)blockIdx.x, part);
}
hipError_t rq_prep_queries_launch(const RqPrepArgs& a, hipStream_t stream) {
    if (a.nslots <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rq_prep_queries_kernel, dim3(a.nslots), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// int8 image of stored rows (one wave per row), see rq_kernels.h rq_quant_rows_launch.  Symmetric per-row scale from the
// row's largest magnitude: no clipping, every element within s/2 of its image.  The measured relative error of the
// worst row is what the int8 scan's certificate rests on, so it is summed in fp64 from the values the scan will use.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rq_quant_rows_kernel(const char* x, const double* norm64, int64_t row_begin, int64_t row_end,
                                                            signed char* x8, float* scale_cos, float* scale_ip, unsigned long long* stat,
                                                            unsigned* binerr) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    double mx = 0.0;
    for (int64_t row = row_begin + wave; row < row_end; row += nwaves) {
        const char* r = x + row * (RQ_DPAD * 2);
        float f[12];
        float am = 0.f;
        bool bad = false;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const rq_half4 v = *(const rq_half4*)(r + p * 512 + lane * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[p * 4 + e] = (float)v[e];
                bad |= !(fabsf(f[p * 4 + e]) <= 65504.f);
                am = fmaxf(am, fabsf(f[p * 4 + e]));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
        bad = __ballot(bad) != 0ull;
        const double nrm = norm64[row];
        const bool live = !bad && am > 0.f && nrm > 0.0;
        const float sr = live ? am / 127.f : 0.f, inv = live ? 127.f / am : 0.f;
        double err = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = fminf(fmaxf(rintf(f[p * 4 + e] * inv), -127.f), 127.f);
                const double d = live ? (double)f[p * 4 + e] - (double)sr * (double)q : 0.0;
                err += d * d;
                packed |= ((uint32_t)(int)q & 0xffu) << (8 * e);
            }
            *(uint32_t*)(x8 + row * RQ_DPAD + p * 256 + lane * 4) = packed;
        }
        err = rq_wave_sum(err);
        if (lane == 0) {
            scale_cos[row] = live ? (float)((double)sr / nrm) : 0.f;
            scale_ip[row] = sr;
        }
        const double e = bad ? (double)__builtin_huge_valf() : (live ? sqrt(err) / nrm : 0.0);
        mx = fmax(mx, e);
        // the worst row of every BIN (64 rows): the tail tests a bin against a threshold that is as much higher as this bin's rows
        // quantise better than the shard's worst row (rq_tail_body.h).  Non-negative floats order like their bit patterns; the
        // value is rounded UP to fp32.
        if (lane == 0 && e > 0.0) atomicMax(&binerr[row >> 6], __float_as_uint(__double2float_ru(e)));
    }
    if (lane == 0 && mx > 0.0) atomicMax(&stat[0], (unsigned long long)__double_as_longlong(mx));
}
hipError_t rq_quant_rows_launch(const void* x, const double* norm64, int64_t row_begin, int64_t row_end, signed char* x8,
                                float* scale_cos, float* scale_ip, unsigned long long* stat, float* binerr, hipStream_t stream) {
    if (row_end <= row_begin) return hipSuccess;
    const int64_t rows = row_end - row_begin;
    const int grid = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    hipLaunchKernelGGL(rq_quant_rows_kernel, dim3(grid), dim3(256), 0, stream, (const char*)x, norm64, row_begin, row_end, x8, scale_cos, scale_ip, stat, (unsigned*)binerr);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// workgroup top-m selection over n keys (all keys distinct, 0 = empty)
// --------------------------------------------------------------------------------------------
#define RQ_SEL_THREADS 1024
#define RQ_SEL_L 4096

// bitonic sort (descending) of list[0..len), len a power of two <= RQ_SEL_L; all 1024 threads call it
__device__ void rq_bitonic_desc(uint64_t* list, int len) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= len; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (len >> 1); t += RQ_SEL_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const uint64_t x = list[i], y = list[p];
                const bool desc = (i & k) == 0;
                if (desc ? (x < y) : (x > y)) { list[i] = y; list[p] = x; }
            }
            __syncthreads();
        }
    }
}

// Leaves the best min(m, #non-empty keys) keys sorted descending in list[0..), returns that count.
template <class KeyAt>
__device__ int rq_wg_topm(KeyAt keyat, int64_t n, int m, uint64_t* list, int* cnt) {
    const int tid = threadIdx.x;
    if (tid == 0) *cnt = 0;
    __syncthreads();
    uint64_t thr = 0;
    uint64_t cur = (tid < n) ? keyat((int64_t)tid) : 0;
    int have = 0;
    for (int64_t base = 0; base < n; base += RQ_SEL_THREADS) {
        const int64_t inext = base + RQ_SEL_THREADS + tid;
        const uint64_t nxt = (inext < n) ? keyat(inext) : 0;
        if (cur > thr) { const int pos = atomicAdd(cnt, 1); list[pos] = cur; }
        __syncthreads();
        const int c = *cnt;
        __syncthreads();   // nobody may bump *cnt (next iteration) before everyone has read it
        const bool last = base + RQ_SEL_THREADS >= n;
        if (c > RQ_SEL_L - RQ_SEL_THREADS || last) {
            int len = 2;
            while (len < c) len <<= 1;
            for (int i = c + tid; i < len; i += RQ_SEL_THREADS) list[i] = 0;
            __syncthreads();
            rq_bitonic_desc(list, len);
            have = c < m ? c : m;
            thr = (c >= m) ? list[m - 1] : 0;
            __syncthreads();
            if (tid == 0) *cnt = have;
            __syncthreads();
        }
        cur = nxt;
    }
    // n == 0: nothing was appended
    return have;
}

// ---- pass 2 -------------------------------------------------------------------------------
__global__ __launch_bounds__(RQ_SEL_THREADS) void rq_select_bins_kernel(const uint2* bins, int64_t stride, int64_t nbins, int m,
                                                                       uint64_t* binkeys) {
    __shared__ uint64_t list[RQ_SEL_L];
    __shared__ int cnt;
    const int q = blockIdx.x;
    const uint2* p = bins + (int64_t)q * stride;
    const int have = rq_wg_topm([&](int64_t i) { return rq_make_key(rq_sanitize(rq_rec_m1(p[i].x)), (uint32_t)i); }, nbins, m, list, &cnt);
    for (int j = threadIdx.x; j < m; j += RQ_SEL_THREADS) binkeys[(int64_t)q * m + j] = j < have ? list[j] : 0;
}
hipError_t rq_select_bins_launch(const uint2* bins, int64_t bins_stride, int64_t nbins, int B, int m, uint64_t* binkeys,
                                 hipStream_t stream) {
    if (m < 1 || m > RQ_SEL_L - RQ_SEL_THREADS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rq_select_bins_kernel, dim3(B), dim3(RQ_SEL_THREADS), 0, stream, bins, bins_stride, nbins, m, binkeys);
    return hipGetLastError();
}

// ---- pass 3 -------------------------------------------------------------------------------
// grid (nb, B); 4 waves; wave w re-scores rows j = w, w+4, ... of the bin in fp64.
__global__ __launch_bounds__(256) void rq_rescore_kernel(RqRescoreArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.y, slot = blockIdx.x;
    constexpr int binrows = RQ_BIN_ROWS;
    uint64_t* out = a.cand + ((int64_t)q * a.nb + slot) * binrows;
    int64_t bin = slot;   // exact mode (binkeys == nullptr): every bin of the shard is re-scored
    if (a.binkeys) {
        const uint64_t bkey = a.binkeys[(int64_t)q * a.binkeys_stride + slot];
        if (bkey == 0) {
            if (threadIdx.x < binrows) out[threadIdx.x] = 0;
            return;
        }
        bin = rq_key_index(bkey);
    }
    float qv[12];
    const float* qp = a.q32 + (size_t)q * RQ_DPAD;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float4 t = *(const float4*)(qp + p * 256 + 4 * lane);
        qv[p * 4 + 0] = t.x; qv[p * 4 + 1] = t.y; qv[p * 4 + 2] = t.z; qv[p * 4 + 3] = t.w;
    }
    const double qn = a.qnorm64[q];
    const char* xb = (const char*)a.x;
    for (int j = wave; j < binrows; j += 4) {
        const int64_t row = bin * RQ_BIN_ROWS + j;
        if (row >= a.n_rows) { if (lane == 0) out[j] = 0; continue; }
        const char* r = xb + row * (RQ_DPAD * 2);
        double dot = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const rq_half4 v = *(const rq_half4*)(r + p * 512 + lane * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) dot += (double)qv[p * 4 + e] * (double)(float)v[e];
        }
        dot = rq_wave_sum(dot);
        if (lane == 0) {
            double s = dot;
            if (a.metric == 0) s = dot / (qn * a.rownorm64[row] + 1e-30);
            out[j] = rq_make_key(rq_sanitize((float)s), (uint32_t)row);
        }
    }
}
hipError_t rq_rescore_launch(const RqRescoreArgs& a, int B, hipStream_t stream) {
    if (a.nb <= 0 || B <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rq_rescore_kernel, dim3(a.nb, B), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---- pass 4 -------------------------------------------------------------------------------
__global__ __launch_bounds__(RQ_SEL_THREADS) void rq_final_kernel(RqFinalArgs a) {
    __shared__ uint64_t list[RQ_SEL_L];
    __shared__ int cnt;
    const int q = blockIdx.x;
    const uint64_t* c = a.cand + (int64_t)q * a.ncand;
    const int have = rq_wg_topm([&](int64_t i) { return c[i]; }, a.ncand, a.k, list, &cnt);
    for (int j = threadIdx.x; j < a.k; j += RQ_SEL_THREADS) {
        const int64_t o = (int64_t)q * a.k + j;
        if (j < have) {
            const uint64_t key = list[j];
            const float s = rq_key_score(key);
            const int64_t grow = a.row_offset + (int64_t)rq_key_index(key);
            a.out_scores[o] = s;
            a.out_rows[o] = grow;
            if (a.out_keys) a.out_keys[o] = rq_make_key(s, (uint32_t)grow);
        } else {
            a.out_scores[o] = 0.f;
            a.out_rows[o] = -1;
            if (a.out_keys) a.out_keys[o] = 0;
        }
    }
    if (threadIdx.x == 0) {
        // Certificate: every row outside the re-scored bins has approximate score <= b, hence exact
        // score <= b + eps (unit-query units). Exact iff that is strictly below the k-th exact score.
        int ok;
        const int64_t kk = a.k < a.n_rows ? a.k : a.n_rows;   // rows that must be returned
        if (a.nbins <= a.nb) {
            ok = 1;   // the whole shard was re-scored
        } else if (have < kk || (a.metric == 0 && a.qnorm64[q] != 0.0 && a.qnorm64[q] < RQ_TINY_QUERY_NORM)) {
            ok = 0;   // (tiny query: see rq_final_body.h RQ_TINY_QUERY_NORM)
        } else if (kk == 0) {
            ok = 1;
        } else {
            const uint64_t bkey = a.binkeys[(int64_t)q * a.binkeys_stride + a.nb];
            if (bkey == 0) {
                ok = 1;
            } else {
                const double b = (double)rq_key_score(bkey);
                const double qn = a.qnorm64[q];
                const double bound = a.metric == 0 ? b + (double)a.eps
                                                   : (b + (double)a.eps * (double)a.max_row_norm) * qn * (1.0 + 1e-6);
                const float sk = rq_key_score(list[kk - 1]);
                ok = ((float)bound < sk) ? 1 : 0;
                if (qn == 0.0) {
                    // every score is 0: the answer is rows 0..k-1, present iff whole leading quads were re-scored
                    ok = (kk <= (int64_t)a.nb * RQ_BIN_ROWS) ? 1 : 0;
                }
            }
        }
        a.out_status[q] = ok ? 0 : 1;
    }
}
hipError_t rq_final_launch(const RqFinalArgs& a, int B, hipStream_t stream) {
    if (a.k < 1 || a.k > RQ_SEL_L - RQ_SEL_THREADS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rq_final_kernel, dim3(B), dim3(RQ_SEL_THREADS), 0, stream, a);
    return hipGetLastError();
}

// ---- cross-shard merge ----------------------------------------------------------------------
__global__ __launch_bounds__(RQ_SEL_THREADS) void rq_merge_keys_kernel(const uint64_t* keys, int n, int k, float* out_scores,
                                                                      int64_t* out_rows, uint64_t* out_keys) {
    __shared__ uint64_t list[RQ_SEL_L];
    __shared__ int cnt;
    const int q = blockIdx.x;
    const uint64_t* c = keys + (int64_t)q * n;
    const int have = rq_wg_topm([&](int64_t i) { return c[i]; }, n, k, list, &cnt);
    for (int j = threadIdx.x; j < k; j += RQ_SEL_THREADS) {
        const int64_t o = (int64_t)q * k + j;
        const uint64_t key = j < have ? list[j] : 0;
        out_scores[o] = key ? rq_key_score(key) : 0.f;
        out_rows[o] = key ? (int64_t)rq_key_index(key) : -1;
        if (out_keys) out_keys[o] = key;
    }
}
hipError_t rq_merge_keys_launch(const uint64_t* keys, int n_per_query, int B, int k, float* out_scores, int64_t* out_rows,
                                uint64_t* out_keys, hipStream_t stream) {
    if (k < 1 || k > RQ_SEL_L - RQ_SEL_THREADS || n_per_query < 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rq_merge_keys_kernel, dim3(B), dim3(RQ_SEL_THREADS), 0, stream, keys, n_per_query, k, out_scores, out_rows,
                       out_keys);
    return hipGetLastError();
}



// --------------------------------------------------------------------------------------------
// Measurement hook: read the stored shard once with plain 16-byte loads and nothing else (XOR-folded so the loads
// stay).  Gives the streaming-read rate THIS GPU reaches right now, the yardstick the scan is compared with
// (boxes differ by 20 %).
// --------------------------------------------------------------------------------------------
typedef unsigned int rq_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void rq_read_probe_kernel(const rq_u32x4* __restrict__ x, int64_t n16, uint32_t* sink) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    rq_u32x4 acc = {0u, 0u, 0u, 0u};
    for (; i + 3 * stride < n16; i += 4 * stride) {
        rq_u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= x[i];
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x9e3779b9u) sink[0] = r;   // practically never: keeps the loads alive without a store per thread
}
hipError_t rq_read_probe_launch(const void* x, int64_t bytes, bool nt, int grid, uint32_t* sink, hipStream_t stream) {
    if (grid <= 0 || (bytes & 15)) return hipErrorInvalidValue;
    if (nt) hipLaunchKernelGGL(rq_read_probe_kernel<true>, dim3(grid), dim3(256), 0, stream, (const rq_u32x4*)x, bytes / 16, sink);
    else hipLaunchKernelGGL(rq_read_probe_kernel<false>, dim3(grid), dim3(256), 0, stream, (const rq_u32x4*)x, bytes / 16, sink);
    return hipGetLastError();
}
