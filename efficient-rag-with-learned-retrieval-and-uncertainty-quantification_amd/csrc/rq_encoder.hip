// rq_encoder.hip -- the memory-bound pieces of the NomicBert (nomic-embed-text) forward pass, fused for gfx950.
//
// SURVEY 8(a1/a2, f1): the reference obtains embeddings from Ollama (rag_uq/streaming_index.py:267-288); here the encoder runs
// in-process on PyTorch-ROCm (embedders.NomicBertEmbedder).  rocprofv3 of the stock `transformers` forward at configs[3]'s
// shape (256 queries x 68 tokens, 12 layers, fp16): 10.3 ms, of which the seven GEMMs per layer are 4.1 ms and everything
// else -- rotary (cat / neg / mul / add), SDPA on 68-token sequences, SiLU, gate * up, residual adds, LayerNorm -- is 6 ms
// of small memory-bound kernels.  The GEMMs stay with hipBLASLt (plain library GEMMs); the rest is four kernels here:
//
//   rq_nb_attention_f16      rotary + softmax(QK^T / 8 + prefix mask) V for sequences of <= 512 tokens, on the matrix cores
//   rq_nb_add_layernorm_f16  LayerNorm(x + residual) * gamma + beta (post-LN block of NomicBertLayer)
//   rq_nb_swiglu_f16         silu(gate) * up on the fused [gate | up] GEMM output
//   rq_nb_mean_pool_f16      masked mean over the valid tokens of a sequence (fp32 out)
//
// Architecture facts used (transformers/models/nomic_bert/modeling_nomic_bert.py of the installed package): 12 heads x 64,
// rotate-half rotary over the whole head dimension with theta from the config, no biases in the projections, post-LN,
// SwiGLU MLP, right-padded batches (valid tokens first).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rq_device.h"
#include "rq_index.h"   // set_err, HIPCHK

#define NB_HEAD_DIM 64
#define NB_MAX_SEQ 512      // keys one attention workgroup stages in LDS (137 KB at 512: one workgroup per CU)

// ---------------------------------------------------------------------------------------------------------------------
// Attention.  Grid (heads, batch), NW waves: a workgroup stages the keys and values of one (sequence, head) in LDS once; its waves
// then walk over the queries in tiles of 16 QG (wave w takes tiles w, w + NW, ..), each wave on its own -- no barrier after the
// staging.  qkv: [batch * L][3 * H] fp16 with H = heads * 64 (q | k | v), ctx: [batch * L][H] fp16.  len[b] = valid tokens of
// sequence b (the first len[b] positions); rows >= len[b] of ctx are written as zeros.
//
// Everything is computed transposed so that no operand ever needs a cross-lane transpose:
//   S^T = K Q^T      A = K rows (lane: key r16, dims 8 kg .. and 32 + 8 kg ..), B = Q rows (lane: query r16, the same dims): a lane's
//                    two Q fragments are exactly one rotary pair group (d, d + 32), so the queries are rotated in registers
//                    straight from global memory -- they never pass through LDS (round 3; before: a 64-query block staged per
//                    barrier pair)
//                    D[key 4 kg + i][query r16]: a lane holds 4 consecutive keys of ONE query -> the softmax statistics of a
//                    query are lane-local sums plus the 3 other lanes of its quad {r16 + 16 kg}: v_permlane16_swap /
//                    v_permlane32_swap (gfx950) + one max / add each, no LDS crossbar traffic (ds_bpermute before)
//   O^T = V^T P^T    B = P^T: the lane's own D registers of two key tiles (keys 4 kg + i of tile t0, then of tile t1) -- the
//                    k index of an MFMA is only a label, so A = V^T is read with the SAME key order: Vt[dim][key] in LDS,
//                    two 8-byte reads per fragment
//                    D[dim 4 kg + i][query r16]: 4 consecutive dims of one query -> one 8-byte store
// QG = 2: a wave carries two 16-query groups through every key step -- the K and V^T fragments are read from LDS once for
// both (LDS reads were 70 % of their issue rate at 512 tokens) and the two QK^T -> softmax -> PV chains overlap each other.
// exp(x - m) is exp2 of scores pre-multiplied by log2(e) / 8 (v_exp_f32 IS exp2); the running output is rescaled only in
// key steps where some query's maximum moved.
// ---------------------------------------------------------------------------------------------------------------------
// LDS of one workgroup, carved at run time for the batch's padded sequence length nkmax = round32(L):
//   k  [nkmax][72] fp16   rotated keys, row-major (rows 144 B apart: conflict-free 16-byte reads)
//   vt [64][nkmax + 8]    values transposed
// 27 KB at L = 68 (4+ workgroups of 4 waves per CU), 69 KB at L = 256 (two), 137 KB at L = 512 (one, of 8 waves).
#define NB_KSTRIDE (NB_HEAD_DIM + 8)
static inline size_t nb_attn_lds_bytes(int nkmax) { return ((size_t)nkmax * NB_KSTRIDE + (size_t)NB_HEAD_DIM * (nkmax + 8)) * 2; }

// rope[pos][0..31] = cos(pos * theta^(-d / 32)), rope[pos][32..63] = sin(...)  (fp32; the reference module rounds them to fp16)
__global__ void rq_nb_rope_table_kernel(float* rope, int seq, float theta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= seq * 32) return;
    const int pos = i >> 5, d = i & 31;
    const float ang = (float)pos * powf(theta, -(float)d * (1.f / 32.f));
    rope[pos * 64 + d] = cosf(ang);
    rope[pos * 64 + 32 + d] = sinf(ang);
}

// rotate the pair of 8-element groups (dims 8 j .., 32 + 8 j ..) of one row: x'[d] = x[d] c - x[d + 32] s, x'[d + 32] = x[d + 32] c + x[d] s
__device__ __forceinline__ void nb_rotate8(const rq_half8 lo, const rq_half8 hi, const float* __restrict__ cs, rq_half8& olo, rq_half8& ohi) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float c = cs[e], sn = cs[32 + e], a = (float)lo[e], b = (float)hi[e];
        olo[e] = (_Float16)(a * c - b * sn);
        ohi[e] = (_Float16)(b * c + a * sn);
    }
}

// max / sum over the 4 lanes {r16 + 16 kg} that share a query: the two swaps leave, in every lane, its own row's value in one result and
// the partner row's in the other (rows of 16 lanes, then halves of 32)
__device__ __forceinline__ float nb_quad_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float nb_quad_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <bool V> struct NbBool { static constexpr bool value = V; };

template <int NW, int QG>
__global__ __launch_bounds__(64 * NW) void rq_nb_attention_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ len, const float* __restrict__ rope,
                                                                  _Float16* __restrict__ ctx, int L, int H, int nkmax, float scale_log2e, int packed) {
    extern __shared__ __attribute__((aligned(16))) char nb_smem[];
    _Float16* const Sk = reinterpret_cast<_Float16*>(nb_smem);
    const int vstride = nkmax + 8;
    _Float16* const Svt = Sk + (size_t)nkmax * NB_KSTRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    // padded batches: len[b] = valid tokens of sequence b, whose rows start at b * L.  Packed batches (no padding rows at all): len[] is the
    // offset table, sequence b = rows [len[b], len[b + 1])
    const int nraw = packed ? len[b + 1] - len[b] : len[b];
    const int n = nraw < L ? nraw : L;                      // valid tokens (keys) of this sequence
    const size_t row0 = packed ? (size_t)len[b] : (size_t)b * L;
    const int ld = 3 * H;
    const rq_half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // rows of padding: zero (the residual + LayerNorm that follows reads them)
    if (!packed)
        for (int i = n * 8 + tid; i < L * 8; i += 64 * NW) *(rq_half8*)(ctx + (row0 + (i >> 3)) * H + head * NB_HEAD_DIM + (i & 7) * 8) = zero8;
    if (n <= 0) return;
    const int nk = (n + 31) & ~31;                          // keys padded to whole 32-key steps (<= nkmax)
    // ---- stage K (rotated) and V^T once: 16-byte global loads, rows beyond n are zero
    for (int i = tid; i < nk * 4; i += 64 * NW) {
        const int r = i >> 2, j = i & 3;
        rq_half8 olo = zero8, ohi = zero8;
        if (r < n) {
            const _Float16* p = qkv + (row0 + r) * ld + H + head * NB_HEAD_DIM + 8 * j;
            nb_rotate8(*(const rq_half8*)p, *(const rq_half8*)(p + 32), rope + r * 64 + 8 * j, olo, ohi);
        }
        *(rq_half8*)(Sk + r * NB_KSTRIDE + 8 * j) = olo;
        *(rq_half8*)(Sk + r * NB_KSTRIDE + 32 + 8 * j) = ohi;
    }
    // V^T: a thread takes 4 consecutive keys x 8 dims -- four 16-byte loads, eight 8-byte LDS writes; the lanes of a wave hold consecutive
    // key groups of one dim block, so every write instruction covers a linear address range (the 2-byte scatter of round 2's form cost
    // 61 M bank-conflict cycles per launch at 512 tokens, a fifth of the kernel's LDS time: profiles/r03_attention512_pmc.json)
    {
        const int ng = nk >> 2;                             // key groups (nk is a multiple of 32)
        for (int i = tid; i < ng * 8; i += 64 * NW) {
            const int r4 = i % ng, j = i / ng;
            rq_half8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * r4 + u;
                v[u] = zero8;
                if (r < n) v[u] = *(const rq_half8*)(qkv + (row0 + r) * ld + 2 * H + head * NB_HEAD_DIM + 8 * j);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) *(rq_half4*)(Svt + (8 * j + e) * vstride + 4 * r4) = rq_half4{v[0][e], v[1][e], v[2][e], v[3][e]};
        }
    }
    __syncthreads();                                        // the only barrier: K and V^T are read-only from here on
    const int r16 = lane & 15, kg = lane >> 4;
    // ---- query tiles of 16 QG, one wave each
    for (int q0 = 16 * QG * wave; q0 < n; q0 += 16 * QG * NW) {
        // B operand of S^T = K Q^T: query q0 + 16 g + r16, dims 8 kg .. (k-step 0) and 32 + 8 kg .. (k-step 1), rotated here
        rq_half8 qf0[QG], qf1[QG];
        float m[QG], l[QG];                                 // running max / sum of this lane's query (identical on its 4 lanes)
        rq_float4 o[QG][4];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const int pos = q0 + 16 * g + r16;
            qf0[g] = zero8; qf1[g] = zero8;
            if (pos < n) {
                const _Float16* p = qkv + (row0 + pos) * ld + head * NB_HEAD_DIM + 8 * kg;
                nb_rotate8(*(const rq_half8*)p, *(const rq_half8*)(p + 32), rope + pos * 64 + 8 * kg, qf0[g], qf1[g]);
            }
            m[g] = -__builtin_huge_valf(); l[g] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[g][t] = rq_float4{0.f, 0.f, 0.f, 0.f};
        }
        // one step of 32 keys; MASK: the sequence ends inside this step (only the last one can: nk = round32(n))
        auto step = [&](auto mask_tag, const int k0) {
            constexpr bool MASK = decltype(mask_tag)::value;
            const _Float16* kr = Sk + (k0 + r16) * NB_KSTRIDE + 8 * kg;
            const rq_half8 a00 = *(const rq_half8*)kr, a01 = *(const rq_half8*)(kr + 32);
            const rq_half8 a10 = *(const rq_half8*)(kr + 16 * NB_KSTRIDE), a11 = *(const rq_half8*)(kr + 16 * NB_KSTRIDE + 32);
            // A fragment of dim tile t of O^T += V^T P^T: Vt[16 t + r16][keys 4 kg .. of tile k0, then of tile k0 + 16]
            rq_half8 vf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const rq_half4 v0 = *(const rq_half4*)(Svt + (16 * t + r16) * vstride + k0 + 4 * kg);
                const rq_half4 v1 = *(const rq_half4*)(Svt + (16 * t + r16) * vstride + k0 + 16 + 4 * kg);
                vf[t] = rq_half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            }
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                rq_float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a00, qf0[g], s0, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a01, qf1[g], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a10, qf0[g], s1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11, qf1[g], s1, 0, 0, 0);
                // prefix mask, online softmax over this lane's 8 keys + the 3 other lanes of the query.  The maximum is taken over the raw
                // dot products (the scale is positive) and exp2(s c - m) is one fma + v_exp_f32 per key
                float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                if constexpr (MASK) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (k0 + 4 * kg + i >= n) sv[i] = -__builtin_huge_valf();
                        if (k0 + 16 + 4 * kg + i >= n) sv[4 + i] = -__builtin_huge_valf();
                    }
                }
                float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
                mx = nb_quad_max(mx);
                const float mn = fmaxf(m[g], mx * scale_log2e);   // finite: key 0 is always valid (n >= 1); -inf * c = -inf
                const float alpha = __builtin_amdgcn_exp2f(m[g] - mn);
                float ps = 0.f;
                rq_half8 pf;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[i], scale_log2e, -mn));   // exp2(-inf) = 0 for masked keys
                    ps += p;
                    pf[i] = (_Float16)p;
                }
                ps = nb_quad_sum(ps);
                l[g] = l[g] * alpha + ps;
                m[g] = mn;
                if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {      // some query's maximum moved: rescale the running output
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[g][t][i] *= alpha;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) o[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[t], pf, o[g][t], 0, 0, 0);
            }
        };
        int k0 = 0;
        for (; k0 + 32 <= n; k0 += 32) step(NbBool<false>{}, k0);
        if (k0 < nk) step(NbBool<true>{}, k0);
        // ---- O[query][dim 16 t + 4 kg + i] = o[t][i] / l
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const int pos = q0 + 16 * g + r16;
            if (pos < n) {
                const float inv = 1.f / l[g];
                _Float16* dst = ctx + (row0 + pos) * H + head * NB_HEAD_DIM + 4 * kg;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    rq_half4 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = (_Float16)(o[g][t][i] * inv);
                    *(rq_half4*)(dst + 16 * t) = h;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// out = LayerNorm(x + res) * gamma + beta over rows of `width` <= 1536 elements (multiple of 8); one wave per row, fp32
// statistics (two-pass over registers).  res may be null; out may alias x or res.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rq_nb_add_layernorm_kernel(const _Float16* x, const _Float16* res, const _Float16* __restrict__ gamma,
                                                                  const _Float16* __restrict__ beta, _Float16* out, int64_t rows, int width, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[24];
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int c = (p * 64 + lane) * 8;
        if (c < width) {
            const rq_half8 a = *(const rq_half8*)(x + row * width + c);
            rq_half8 r = {0, 0, 0, 0, 0, 0, 0, 0};
            if (res) r = *(const rq_half8*)(res + row * width + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[p * 8 + e] = (float)a[e] + (float)r[e]; sum += v[p * 8 + e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[p * 8 + e] = 0.f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)width;
    float var = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int c = (p * 64 + lane) * 8;
        if (c < width)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[p * 8 + e] - mean; var += d * d; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off, 64);
    const float rstd = rsqrtf(var / (float)width + eps);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int c = (p * 64 + lane) * 8;
        if (c < width) {
            const rq_half8 g = *(const rq_half8*)(gamma + c), bt = *(const rq_half8*)(beta + c);
            rq_half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((v[p * 8 + e] - mean) * rstd * (float)g[e] + (float)bt[e]);
            *(rq_half8*)(out + row * width + c) = o;
        }
    }
}

// out[t][j] = silu(gu[t][j]) * gu[t][inter + j];  gu: [rows][2 * inter], out: [rows][inter]; inter % 8 == 0
__global__ __launch_bounds__(256) void rq_nb_swiglu_kernel(const _Float16* __restrict__ gu, _Float16* __restrict__ out, int64_t rows, int inter) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= rows * inter) return;
    const int64_t t = i / inter;
    const int j = (int)(i - t * inter);
    const rq_half8 g = *(const rq_half8*)(gu + t * 2 * inter + j), u = *(const rq_half8*)(gu + t * 2 * inter + inter + j);
    rq_half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (float)g[e];
        o[e] = (_Float16)(x / (1.f + __expf(-x)) * (float)u[e]);
    }
    *(rq_half8*)(out + i) = o;
}

// out[b][d] = sum over the first len[b] tokens of h[b][t][d] / max(len[b], 1), fp32.  Grid (batch), 256 threads; width <= 2048.
// packed: len[] is the offset table of a batch without padding rows (sequence b = rows [len[b], len[b + 1])).
__global__ __launch_bounds__(256) void rq_nb_mean_pool_kernel(const _Float16* __restrict__ h, const int* __restrict__ len, float* __restrict__ out, int L, int width, int packed) {
    const int b = blockIdx.x;
    const int nraw = packed ? len[b + 1] - len[b] : len[b];
    const int n = nraw < L ? nraw : L;
    h += (packed ? (size_t)len[b] : (size_t)b * L) * width;
    for (int c = threadIdx.x * 8; c < width; c += 256 * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < n; ++t) {
            const rq_half8 v = *(const rq_half8*)(h + (size_t)t * width + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
        const float inv = 1.f / (float)(n > 0 ? n : 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) out[(size_t)b * width + c + e] = acc[e] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI (include/rq.h).  Device pointers, the caller's stream; shapes are checked on the host before any launch.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int rq_nb_rope_table_f32(float* d_rope, int seq, float rope_theta, void* stream) {
    if (!d_rope) return set_err(RQ_EINVAL, "null argument");
    if (seq < 1 || seq > 65536 || !(rope_theta > 1.f)) return set_err(RQ_EINVAL, "seq %d / rope_theta %g", seq, (double)rope_theta);
    hipLaunchKernelGGL(rq_nb_rope_table_kernel, dim3((unsigned)((seq * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rope, seq, rope_theta);
    HIPCHK(hipGetLastError());
    return RQ_OK;
}

static int nb_attention_common(const void* d_qkv, const int* d_len, const float* d_rope, void* d_ctx, int batch, int seq, int heads, void* stream, int packed) {
    if (!d_qkv || !d_len || !d_rope || !d_ctx) return set_err(RQ_EINVAL, "null argument");
    if (batch < 1 || batch > 65535 || heads < 1 || heads > 65535) return set_err(RQ_EINVAL, "batch %d / heads %d outside 1..65535", batch, heads);
    if (seq < 1 || seq > NB_MAX_SEQ) return set_err(RQ_EUNSUPPORTED, "sequence length %d outside 1..%d: use the framework's attention for longer inputs", seq, NB_MAX_SEQ);
    const int nkmax = (seq + 31) & ~31;
    // two or more 4-wave workgroups per CU while their LDS allows it, else one of 8 waves; two query groups per wave from 129 tokens on
    // (below that a sequence has too few 32-query tiles for its waves)
    const size_t lds = nb_attn_lds_bytes(nkmax);
    const int form = 2 * lds > (size_t)160 * 1024 ? 2 : (seq > 128 ? 1 : 0);
    const dim3 grid((unsigned)heads, (unsigned)batch);
    static unsigned long long attr_done = 0;   // one bit per device
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (!((attr_done >> (dev & 63)) & 1ull)) {
        HIPCHK(hipFuncSetAttribute((const void*)rq_nb_attention_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_attn_lds_bytes(NB_MAX_SEQ)));
        HIPCHK(hipFuncSetAttribute((const void*)rq_nb_attention_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_attn_lds_bytes(NB_MAX_SEQ)));
        HIPCHK(hipFuncSetAttribute((const void*)rq_nb_attention_kernel<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_attn_lds_bytes(NB_MAX_SEQ)));
        attr_done |= 1ull << (dev & 63);
    }
    const float scale_log2e = 0.125f * 1.4426950408889634f;      // 1 / sqrt(64), in the log2 domain of v_exp_f32
#define NB_ATTN_LAUNCH(NW_, QG_) hipLaunchKernelGGL((rq_nb_attention_kernel<NW_, QG_>), grid, dim3(64 * NW_), lds, (hipStream_t)stream, (const _Float16*)d_qkv, d_len, \
                                                    d_rope, (_Float16*)d_ctx, seq, heads * NB_HEAD_DIM, nkmax, scale_log2e, packed)
    if (form == 2) NB_ATTN_LAUNCH(8, 2);
    else if (form == 1) NB_ATTN_LAUNCH(4, 2);
    else NB_ATTN_LAUNCH(4, 1);
#undef NB_ATTN_LAUNCH
    HIPCHK(hipGetLastError());
    return RQ_OK;
}

extern "C" int rq_nb_attention_f16(const void* d_qkv, const int* d_len, const float* d_rope, void* d_ctx, int batch, int seq, int heads, void* stream) {
    return nb_attention_common(d_qkv, d_len, d_rope, d_ctx, batch, seq, heads, stream, 0);
}

extern "C" int rq_nb_attention_packed_f16(const void* d_qkv, const int* d_offsets, const float* d_rope, void* d_ctx, int batch, int max_seq, int heads, void* stream) {
    return nb_attention_common(d_qkv, d_offsets, d_rope, d_ctx, batch, max_seq, heads, stream, 1);
}

extern "C" int rq_nb_add_layernorm_f16(const void* d_x, const void* d_res, const void* d_gamma, const void* d_beta, void* d_out, int64_t rows, int width,
                                       float eps, void* stream) {
    if (!d_x || !d_gamma || !d_beta || !d_out) return set_err(RQ_EINVAL, "null argument");
    if (rows < 1 || width < 8 || width > 1536 || width % 8) return set_err(RQ_EINVAL, "rows %lld / width %d (8..1536, multiple of 8)", (long long)rows, width);
    hipLaunchKernelGGL(rq_nb_add_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)d_x,
                       (const _Float16*)d_res, (const _Float16*)d_gamma, (const _Float16*)d_beta, (_Float16*)d_out, rows, width, eps);
    HIPCHK(hipGetLastError());
    return RQ_OK;
}

extern "C" int rq_nb_swiglu_f16(const void* d_gate_up, void* d_out, int64_t rows, int inter, void* stream) {
    if (!d_gate_up || !d_out) return set_err(RQ_EINVAL, "null argument");
    if (rows < 1 || inter < 8 || inter % 8) return set_err(RQ_EINVAL, "rows %lld / intermediate size %d (multiple of 8)", (long long)rows, inter);
    const int64_t vec = rows * inter / 8;
    hipLaunchKernelGGL(rq_nb_swiglu_kernel, dim3((unsigned)((vec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)d_gate_up, (_Float16*)d_out,
                       rows, inter);
    HIPCHK(hipGetLastError());
    return RQ_OK;
}

extern "C" int rq_nb_mean_pool_f16(const void* d_h, const int* d_len, float* d_out, int batch, int seq, int width, void* stream) {
    if (!d_h || !d_len || !d_out) return set_err(RQ_EINVAL, "null argument");
    if (batch < 1 || seq < 1 || width < 8 || width > 2048 || width % 8) return set_err(RQ_EINVAL, "batch %d / seq %d / width %d", batch, seq, width);
    hipLaunchKernelGGL(rq_nb_mean_pool_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const _Float16*)d_h, d_len, d_out, seq, width, 0);
    HIPCHK(hipGetLastError());
    return RQ_OK;
}

extern "C" int rq_nb_mean_pool_packed_f16(const void* d_h, const int* d_offsets, float* d_out, int batch, int max_seq, int width, void* stream) {
    if (!d_h || !d_offsets || !d_out) return set_err(RQ_EINVAL, "null argument");
    if (batch < 1 || max_seq < 1 || width < 8 || width > 2048 || width % 8) return set_err(RQ_EINVAL, "batch %d / max_seq %d / width %d", batch, max_seq, width);
    hipLaunchKernelGGL(rq_nb_mean_pool_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const _Float16*)d_h, d_offsets, d_out, max_seq, width, 1);
    HIPCHK(hipGetLastError());
    return RQ_OK;
}
