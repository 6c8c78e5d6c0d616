// rq_device.h -- shared device-side helpers for the dense-retrieval hot path (gfx950 only).
//
// Replaces the arithmetic the reference delegates to ChromaDB's cosine kNN
// (reference rag_uq/streaming_index.py:355-368, collection.query + `1 - distance`).
//
// Vocabulary
//   row      one passage vector of the corpus (fp16, DPAD elements, zero padded)
//   quad     64 consecutive rows = 4 tiles of 16 rows; the scan kernel's work unit
//   bin      = quad: the 64 rows whose largest / second-largest approximate score and arg-max the scan keeps
//   key      64-bit sortable (score, index) pair: larger key = better rank
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RQ_DPAD 768            // padded row length the kernels are specialised for
#define RQ_QBLOCK 64           // queries scored per corpus pass
#define RQ_QUAD_ROWS 64
#define RQ_QSCALE 4096.0f             // unit queries are scaled by 2^12 before their fp16 rounding (fp16 subnormal flush of the
#define RQ_QSCALE_INV 0.000244140625f // matrix cores, rq_select.hip); the scan's row scales carry 2^-12
#define RQ_TILE_ROWS 16

typedef _Float16 rq_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 rq_half4 __attribute__((ext_vector_type(4)));
typedef float rq_float4 __attribute__((ext_vector_type(4)));
typedef int rq_int4 __attribute__((ext_vector_type(4)));

// Order-preserving map float -> uint32 (larger float = larger uint). NaN must be removed first.
__host__ __device__ static inline uint32_t rq_mono32(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float rq_unmono32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// key = (mono(score) << 32) | (0xffffffff - index): sorts by score desc, then index asc.
// key 0 is reserved for "empty".
__host__ __device__ static inline uint64_t rq_make_key(float score, uint32_t index) {
    if (score == 0.f) score = 0.f;   // -0.0 (an underflowed negative) ranks with +0.0: equal scores order by row
    return ((uint64_t)rq_mono32(score) << 32) | (uint64_t)(0xffffffffu - index);
}
__host__ __device__ static inline float rq_key_score(uint64_t key) { return rq_unmono32((uint32_t)(key >> 32)); }
__host__ __device__ static inline uint32_t rq_key_index(uint64_t key) { return 0xffffffffu - (uint32_t)(key & 0xffffffffu); }

// Upper bound of a float in its 16 high bits: positive values round the magnitude up, negative ones truncate
// (toward zero = up); -inf stays -inf.  The low 16 bits of the result are zero.
__host__ __device__ static inline uint32_t rq_up16(float f) {
    if (f == 0.f) f = 0.f;           // -0.0 -> +0.0 so that the codes below stay monotone
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? (u & 0xffff0000u) : ((u + 0xffffu) & 0xffff0000u);
}
// One 8-byte record per (query, bin) written by the scan (x, y):
//   x = [31:6] the bin's largest approximate score m1, fp32 bits rounded UP to 26 bits | [5:0] its row (0..63) in the bin
//   y = [31:16] c2 = 16-bit order-preserving code of the second-largest score m2 rounded up (rq_code16)
//       [15:6]  d  = min(c2 - c3, 1023), c3 the same code of the third-largest score m3: m3 <= decode(c2 - d)
//       [5:0]   row of the second largest
// Every decoded value is an UPPER bound of the true one, which is all the exactness argument needs.
#define RQ_BIN_ROWS 64
__host__ __device__ static inline uint32_t rq_up26(float f) {
    if (f == 0.f) f = 0.f;
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? (u & 0xffffffc0u) : ((u + 63u) & 0xffffffc0u);
}
__host__ __device__ static inline float rq_rec_m1(uint32_t x) { const uint32_t u = x & 0xffffffc0u; float f; __builtin_memcpy(&f, &u, 4); return f; }
// 16-bit code of f rounded up to 16 bits: c(a) <= c(b) whenever a <= b; rq_code16_value(rq_code16(f)) >= f
__host__ __device__ static inline uint32_t rq_code16(float f) {
    const uint32_t u = rq_up16(f);                                   // low 16 bits are zero
    return ((u & 0x80000000u) ? ~u : (u | 0x80000000u)) >> 16;
}
__host__ __device__ static inline float rq_code16_value(uint32_t c) {
    const uint32_t k = (c << 16) | ((c & 0x8000u) ? 0u : 0xffffu);   // the mono32 image of a 16-bit-rounded float
    return rq_unmono32(k);
}

#ifdef __HIPCC__
// ---- selection with the row position inside the score (rq_scan_wide.hip, rq_scan.hip EPI = 1) ----------------------
// A score is clamped to the finite range (an infinite score stays the largest, NaN becomes the smallest), its 6 low
// mantissa bits are REPLACED by the row's position in the quad, and it is inserted into the lane's sorted triple
// m1 >= m2 >= m3 with three VALU instructions: positions ride along, no data-dependent code.  Written as asm because the
// compiler turns fmaxf / fmed3(a, b, +inf) on a value that went through integer bit operations into canonicalise + v_max;
// the operands are never NaN here.  The perturbation (< 64 ulp, 7.6e-6 relative) is part of the scan's error bound.
__device__ __forceinline__ float rq_pos_score(float acc_times_scale, uint32_t pos) {
    const float sc = __builtin_amdgcn_fmed3f(acc_times_scale, -3.4028234664e38f, 3.4028234664e38f);
    return __uint_as_float((__float_as_uint(sc) & 0xffffffc0u) | pos);
}
// The same without the clamp, for scores that are finite or NaN by construction (the int8 scan: an exact int32 sum times a finite row
// scale, NaN only from the pad rows' NaN scale): a NaN needs no special care -- v_max_f32 returns the other operand and v_med3_f32 with a
// NaN operand returns the MINIMUM of the other two, so rq_insert3 leaves a sorted triple m1 >= m2 >= m3 exactly as it was (the position
// bits keep a quiet NaN a quiet NaN).  One VALU less per score where instruction issue is the bound (rq_scan_wide.hip I8).
__device__ __forceinline__ float rq_pos_score_finite(float acc_times_scale, uint32_t pos) {
    return __uint_as_float((__float_as_uint(acc_times_scale) & 0xffffffc0u) | pos);
}
// value part times a positive scale, position bits kept (one more truncation of the 6 low bits: part of the error bound)
__device__ __forceinline__ float rq_scale_pos(float x, float scale) {
    const uint32_t b = __float_as_uint(x);
    const float v = __uint_as_float(b & 0xffffffc0u) * scale;
    return __uint_as_float((__float_as_uint(v) & 0xffffffc0u) | (b & 63u));
}
__device__ __forceinline__ void rq_insert3(float& m1, float& m2, float& m3, float x) {
    float n3, n2, n1;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(n3) : "v"(m2), "v"(m3), "v"(x));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(n2) : "v"(m1), "v"(m2), "v"(x));
    asm("v_max_f32 %0, %1, %2" : "=v"(n1) : "v"(m1), "v"(x));
    m3 = n3; m2 = n2; m1 = n1;
}
// The 8-byte bin record from a merged triple whose values carry complete positions.  Every field is an UPPER bound of the
// UNPERTURBED score: low bits all ones for a positive value, all zeros for a negative one, then rounded up to 26 / 16 bits.
__device__ __forceinline__ uint2 rq_record_from_triple(float x1, float x2, float x3) {
    const uint32_t b1 = __float_as_uint(x1), b2 = __float_as_uint(x2), b3 = __float_as_uint(x3);
    const uint32_t f1 = (b1 & 0xffffffc0u) + ((int32_t)b1 >= 0 ? 64u : 0u);   // = rq_up26 of that bound
    const uint32_t u2 = (int32_t)b2 >= 0 ? (b2 | 63u) : (b2 & 0xffffffc0u);
    const uint32_t u3 = (int32_t)b3 >= 0 ? (b3 | 63u) : (b3 & 0xffffffc0u);
    const uint32_t c2 = rq_code16(__uint_as_float(u2)), c3 = rq_code16(__uint_as_float(u3)), d = c2 - c3;   // c3 <= c2
    return make_uint2(f1 | (b1 & 63u), (c2 << 16) | ((d < 1023u ? d : 1023u) << 6) | (b2 & 63u));
}

__device__ __forceinline__ double rq_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;   // identical in every lane (xor butterfly)
}
// Query preparation of one query slot by one 256-thread workgroup (thread t owns elements t, 256 + t, 512 + t):
// fp64 norm, qh = fp16(q / |q| * 2^12) (the factor: rq_select.hip), zero padded raw copy, slots >= B all zero; and, when
// wanted, the int8 image of the query with its scale and its measured quantisation error (rq_kernels.h RqPrepArgs).
// `part` = 16 doubles of LDS.
template <class PrepArgs>
__device__ __forceinline__ void rq_prep_body(const PrepArgs& a, const int qi, double* part) {
    const int tid = threadIdx.x;
    float v[3];
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int i = p * 256 + tid;
        v[p] = (qi < a.B && i < a.dim) ? a.q[(size_t)qi * a.dim + i] : 0.f;
        acc += (double)v[p] * (double)v[p];
    }
    acc = rq_wave_sum(acc);
    float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if ((tid & 63) == 0) { part[tid >> 6] = acc; part[4 + (tid >> 6)] = (double)am; }
    __syncthreads();
    const double nrm = sqrt((part[0] + part[1]) + (part[2] + part[3]));
    if (tid == 0 && qi < a.B) a.qnorm64[qi] = nrm;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int i = p * 256 + tid;
        const float f = nrm > 0.0 ? (float)((double)v[p] / nrm) * RQ_QSCALE : 0.f;
        a.qh[(size_t)qi * RQ_DPAD + i] = (_Float16)f;
        a.q32pad[(size_t)qi * RQ_DPAD + i] = v[p];
    }
    if (a.q8) {
        const bool finite = nrm == nrm && nrm <= 1.7e308;   // (a NaN element makes the norm NaN, an infinite one infinite)
        const float amax = (float)fmax(fmax(part[4], part[5]), fmax(part[6], part[7]));
        const bool live = finite && amax > 0.f;
        const float sq = live ? amax / 127.f : 0.f, inv = live ? 127.f / amax : 0.f;
        const float slo = sq / 254.f, invlo = live ? 254.f / sq : 0.f;   // residual step: |q - sq r| <= sq / 2 = 127 slo
        double err = 0.0, errs = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float r = fminf(fmaxf(rintf(v[p] * inv), -127.f), 127.f);
            const double d = live ? (double)v[p] - (double)sq * (double)r : 0.0;
            err += d * d;
            const float rl = fminf(fmaxf(rintf((float)d * invlo), -127.f), 127.f);
            const double ds = live ? d - (double)slo * (double)rl : 0.0;
            errs += ds * ds;
            a.q8[(size_t)qi * RQ_DPAD + p * 256 + tid] = (signed char)(int)r;
            a.q8lo[(size_t)qi * RQ_DPAD + p * 256 + tid] = (signed char)(int)rl;
        }
        err = rq_wave_sum(err);
        errs = rq_wave_sum(errs);
        if ((tid & 63) == 0) { part[8 + (tid >> 6)] = err; part[12 + (tid >> 6)] = errs; }
        __syncthreads();
        if (tid == 0) {
            const double e = sqrt((part[8] + part[9]) + (part[10] + part[11]));
            const double es = sqrt((part[12] + part[13]) + (part[14] + part[15]));
            a.qscale8[qi] = live ? (float)((double)sq / nrm) : 1.f;
            // rounded up to fp32; +inf for a non-finite query (never certified from the approximate pass)
            a.qeps8[qi] = !finite ? __builtin_huge_valf() : (live ? (float)(e / nrm) * 1.000001f + 1e-30f : 0.f);
            a.qeps8s[qi] = !finite ? __builtin_huge_valf() : (live ? (float)(es / nrm) * 1.000001f + 1e-30f : 0.f);
        }
    }
}

__device__ __forceinline__ float rq_sanitize(float f) { return (f != f) ? -__builtin_huge_valf() : f; }
__device__ __forceinline__ uint64_t rq_wave_max_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
#endif
