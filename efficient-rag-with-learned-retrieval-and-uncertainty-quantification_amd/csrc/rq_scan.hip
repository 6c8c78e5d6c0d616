// rq_scan.hip -- pass 1 of the search: stream the fp16 corpus once from HBM, score it against a
// block of 64 (or 128) queries on the matrix cores, and keep per (query, bin = quad of 64 rows) the largest
// approximate score, the second largest and the position of the largest.
//
// This is the arithmetic ChromaDB's cosine index performs behind
// reference rag_uq/streaming_index.py:355-359 (collection.query), done exhaustively.
//
// Shape of the work (why it looks like this on MI355X):
//   * HBM-bound: N*1536 B of corpus per query block, 64 FLOP/B -> the matrix cores run ~16% busy.
//   * One workgroup = QW waves (4 or 8); wave w keeps queries 16w..16w+15 in registers as the B operand
//     of v_mfma_f32_16x16x32_f16 (24 fragments x 4 VGPRs), so LDS holds only corpus bytes.
//   * Corpus rows reach LDS by LDS-DMA (global_load_lds_dwordx4): every wave-instruction moves
//     1 KiB contiguous, no VGPRs, and stays in flight across barriers (counted vmcnt, raw s_barrier).
//     A stage = 16 rows x 384 elements (12 KiB, KS = 2) or 16 whole rows (24 KiB, KS = 1); S stages
//     form a ring, S-1 are in flight.  Every variant measured lands within 2 % of the same rate (6.3-6.5 TB/s).
//   * The stage rows alias in LDS banks, so the 16-byte chunks are XOR-swizzled with the row number on
//     the DMA *source* address and on the ds_read_b128 address (LDS image stays lane-linear as
//     LDS-DMA requires).
//   * Accumulator layout of the 16x16 MFMA puts the query on the lane and 4 corpus rows in the 4
//     result registers: max / second max / arg-max over the 16 rows a lane sees of a quad are lane-local (no
//     data-dependent control flow in the streaming loop); the 4 lanes that share a query are merged with two
//     xor-shuffles per quad.
//   * Outputs: ONE 8-byte record per (query, quad) -- largest score, upper bounds of the second and third largest,
//     rows of the largest two (rq_device.h); N/64 * 64 * 8 B = 0.5 % of the corpus bytes -- and
//     wgmax[query][workgroup].  Writes are what the kernel is sensitive to: with a record per 16 rows (65 MB per
//     launch, 16-byte pieces) the stores cost 62 of 289 us (measured by switching them off); a workgroup therefore
//     owns a CONTIGUOUS range of quads and parks its records in LDS until the range is done (see `flush`).
#include <hip/hip_ext.h>

#include "rq_device.h"
#include "rq_kernels.h"
#include "rq_tail_body.h"


extern __shared__ __attribute__((aligned(16))) char rq_smem[];

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// Records per query a scan workgroup parks in LDS: covers a whole range at the usual grid (1M rows: 30.5 quads per
// workgroup with 64-query workgroups, 61 with the 128-query ones, which run one per CU and have the LDS for it).
__host__ __device__ static constexpr int rq_stage_quads(int QW) { return QW == 8 ? 64 : 32; }

template <int N>
__device__ __forceinline__ void rq_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// S: ring depth; NT: non-temporal corpus loads; PF: A fragments read from LDS ahead of
// their MFMAs (1, 4, 6 or 12); OCC: waves per SIMD the register allocation must allow; KS: stages per
// tile (2: a stage is 16 half rows = 12 KiB; 1: a stage is 16 whole rows = 24 KiB contiguous in HBM);
// QW: waves per workgroup = 16-query groups scored per corpus pass (4: 64 queries, 8: 128 queries).
// EPI: 1 = selection with the row position in the low mantissa bits of the score (rq_device.h rq_insert3: 6 VALU per
//      score, no data-dependent code, no per-score row test -- the pad rows' row scale is NaN); 0 = compare / select form.
// I8: the corpus operand is the int8 image of the shard (rq_select.hip rq_quant_rows_kernel: 768 B per row, per-row scale)
//     and the queries are int8 too (rq_prep_body); v_mfma_i32_16x16x64_i8 sums exactly in int32.  A stage is 16 WHOLE rows
//     (12 KiB, the byte geometry of the fp16 half-row stage, so DMA, swizzle and LDS reads are the same code), every stage
//     ends a tile, and the per-query scale is applied once per quad.  Half the HBM bytes per row; the certificate's bound is
//     the quantisation error measured at add / preparation time (csrc/rq_api.hip scan8_eps).
//     I8 = 2: the queries are TWO int8 images (value and residual, q = s (254 q_hi + q_lo)); every corpus fragment feeds two
//     MFMAs and the score is (254 sum_hi + sum_lo) * scales: the query's share of the error bound drops from ~0.008 to ~3e-5
//     for twice the (idle) matrix-core work and no extra bytes.
//     I8 = 3: 128 queries per pass -- a wave keeps TWO groups of 16 queries (one int8 image each: 2 x 48 VGPRs, where the fp16
//     form would need 2 x 96) and every corpus fragment feeds one MFMA per group; records of 128 queries are parked in LDS
//     (68.5 KB per workgroup: two per CU, so this form is not fused with a tail).
template <int S, bool NT, int PF, int KS, int QW, int EPI = 0, int I8 = 0>
__device__ __forceinline__ void rq_scan_body(const RqScanArgs& a, const int b, const int G) {
    static_assert(S >= 2 && S <= 8, "ring depth");
    static_assert(PF == 1 || PF == 4 || PF == 6 || PF == 12, "fragment prefetch group");
    static_assert(KS == 1 || KS == 2, "stages per tile");
    static_assert(I8 == 0 || (KS == 2 && EPI == 1), "int8 scan: built for the half-row stage geometry and the med3 selection");
    constexpr int ROWB = I8 ? RQ_DPAD : RQ_DPAD * 2;   // bytes per corpus row
    constexpr int KL = I8 ? 1 : KS;                // stages per tile
    constexpr int CH = 96 / KS;                    // 16-byte chunks per stage row
    constexpr int STAGE_BYTES = 16 * CH * 16;      // 24576 / KS
    static_assert((24 / KS) % QW == 0, "DMA instructions of a stage must split evenly over the waves");
    constexpr int DPW = 24 / KS / QW;              // DMA wave-instructions per wave per stage
    constexpr int MF = 24 / KS;                    // MFMAs per stage per wave
    constexpr int NSTQ = 4 * KL;                   // stages per quad
    constexpr int VM_KEEP = DPW * (S - 2);         // DMA ops of stages st+1 .. st+S-2 may stay in flight
    constexpr unsigned AUX = NT ? 2u : 0u;
    constexpr int QG = I8 == 3 ? 2 : 1;            // 16-query groups per wave

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = lane >> 4;          // k-group of the MFMA operand / row group of the result
    const int r16 = lane & 15;         // corpus row inside the tile (A operand), query inside the wave (D)

    // (query fragments are built after the DMA prologue has been issued, see below)
    // ---- per-lane DMA source offsets: LDS chunk p = 64*j + lane of a stage holds
    //      row r = p / CH, source chunk c = (p % CH) ^ r   (r < 16, XOR stays inside a 16-chunk group)
    unsigned voff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int p = 64 * (wave * DPW + i) + lane;
        const int r = p / CH, cp = p % CH;
        voff[i] = (unsigned)(r * ROWB + ((cp ^ r) << 4));
    }
    // ---- per-lane LDS read offsets: logical chunk 4*s + kg of row r16 sits at chunk ((4s+kg) ^ r16)
    //      = 4*(s ^ (r16>>2)) + (kg ^ (r16&3));  split s = (s & ~3) | (s & 3): the high part is an immediate, the low
    //      part m = s & 3 enters as ((m ^ (r16>>2)) << 6) = (m << 6) ^ ((r16>>2) << 6).  The row base r16 * CH * 16 is a
    //      multiple of 256, so bits 6..7 of rbase0 hold only that term and rbase(m) = rbase0 ^ (m << 6): one VGPR and
    //      a v_xor per read instead of four VGPRs (the register budget is 168, see rq_scan_tail_kernel).
    static_assert((CH * 16) % 256 == 0, "row pitch must keep bits 6..7 free");
    const unsigned rbase0 = (unsigned)(r16 * (CH * 16) + ((kg ^ (r16 & 3)) << 4) + ((r16 >> 2) << 6));

    // this workgroup's quads: the contiguous range [q_lo, q_lo + nloc)
    const int q_lo = (int)((int64_t)b * a.nquads / G);
    const int nloc = (int)((int64_t)(b + 1) * a.nquads / G) - q_lo;
    const int nst = nloc * NSTQ;
    const char* xb = (const char*)a.x;
    char* norm_lds = rq_smem + S * STAGE_BYTES;                 // [2 parities][64 row scales], shared by the waves
    constexpr int SQ = rq_stage_quads(QW);                      // records per query parked in LDS before they are written out
    uint2* const stg = (uint2*)(norm_lds + 512);                // [16 * QW * QG queries][SQ] finished records

    auto issue = [&](int st, int slot) {
        const int lq = st / NSTQ, t = (st / KL) & 3, kh = st % KL;
        const int64_t quad = (int64_t)q_lo + lq;
        const char* g = xb + (quad * RQ_QUAD_ROWS + t * RQ_TILE_ROWS) * (int64_t)ROWB + kh * (ROWB / KL);
        char* l = rq_smem + slot * STAGE_BYTES + (wave * DPW) * 1024;
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + voff[i]), (lds_ptr_t)(l + i * 1024), 16, 0, AUX);
        if ((st % NSTQ) == 0 && wave == 0) {   // row scales of the quad (256 B): wave 0's counted wait + the stage barrier
            const float* ns = a.row_scale + quad * RQ_QUAD_ROWS + lane;   // make them visible to all waves
            __builtin_amdgcn_global_load_lds((glb_ptr_t)ns, (lds_ptr_t)(norm_lds + ((lq & 1) << 8)), 4, 0, 0);
        }
    };

    int islot = 0;   // slot the next issued stage goes to
    {
        const int pre = nst < S - 1 ? nst : S - 1;
        for (int st = 0; st < pre; ++st) { issue(st, islot); islot = (islot + 1 == S) ? 0 : islot + 1; }
    }
    int cslot = 0;   // slot of the stage being consumed
    const float NEG_INF = -__builtin_huge_valf();
    float wmax = NEG_INF;   // largest approximate score this lane has produced (feeds the tail's threshold)
    float wmax1 = NEG_INF;  // I8 = 3: the same for the second query group

    // ---- query fragments: B[k = 8*kg + j][col = r16] of k-step s == qh[16*wave + r16][32*s + 8*kg + j]
    //      (unit-norm fp16 queries written by rq_prep_queries_kernel); loaded while the first stages are in flight
    //      int8: B[k = 16*kg + j][col = r16] of k-step s == q8[16*wave + r16][64*s + 16*kg + j], 12 fragments
    constexpr int NQF = I8 ? 12 : 24;
    rq_half8 qf[NQF];   // (int8: the same 16 bytes per fragment, reinterpreted at the MFMA)
    rq_half8 ql[I8 >= 2 ? 12 : 1];   // I8 = 2: fragments of the residual image; I8 = 3: of the wave's second query group
    {
        const rq_half8* qsrc = (const rq_half8*)((const char*)a.qh + (size_t)(16 * QG * wave + r16) * ROWB + 16 * kg);
#pragma unroll
        for (int s = 0; s < NQF; ++s) qf[s] = qsrc[4 * s];
#pragma unroll
        for (int s = 0; s < NQF; ++s) asm volatile("" : "+v"(qf[s]));   // ordinary loads retired before the main loop
        if constexpr (I8 >= 2) {
            const rq_half8* lsrc = I8 == 2 ? (const rq_half8*)((const char*)a.qlo + (size_t)(16 * wave + r16) * ROWB + 16 * kg)
                                           : (const rq_half8*)((const char*)a.qh + (size_t)(16 * QG * wave + 16 + r16) * ROWB + 16 * kg);
#pragma unroll
            for (int s = 0; s < 12; ++s) ql[s] = lsrc[4 * s];
#pragma unroll
            for (int s = 0; s < 12; ++s) asm volatile("" : "+v"(ql[s]));
        }
    }
    float qsc = 1.f;    // int8: s_q / |q| of this lane's query (rq_prep_body), applied once per quad
    float qsc1 = 1.f;   // I8 = 3: scale of the second group's query
    if (I8) qsc = a.qscale[16 * QG * wave + r16];
    if (I8 == 2) qsc *= (1.f / 254.f);
    if (I8 == 3) qsc1 = a.qscale[16 * QG * wave + 16 + r16];

    // Finished records wait in LDS and leave in ONE burst per SQ quads (normally once, at the end of the
    // workgroup's range).  Stores inside the streaming loop are what this kernel is sensitive to: every store
    // instruction that touches 16 different lines holds up the CU's vector-memory address pipe for ~500 cycles
    // (TCP_TCP_TA_ADDR_STALL / _DATA_STALL counters), and the DMA loads queue behind it.  Measured per launch (same
    // box, same run; no store at all = 232 us): a record per 16 rows +62 us, a 16-byte record per quad +45 us, an
    // 8-byte record per quad stored every 4 quads +32 us (non-temporal / write-through / plain alike, any ring depth,
    // any position inside the stage), every quad +60 us.
    auto flush = [&](int quad0, int count) {
        // each wave writes the rows of its own 16 queries: 64 / SQ queries x SQ records (runs of 8 SQ bytes) per instruction
        constexpr int QPI = 64 / SQ;
#pragma unroll 1
        for (int i = 0; i < 16 * QG / QPI; ++i) {
            const int qi = 16 * QG * wave + QPI * i + lane / SQ, j = lane & (SQ - 1);
            if (j < count && qi < a.nq_valid) a.bins[(int64_t)qi * a.bins_stride + quad0 + j] = stg[qi * SQ + j];
        }
    };
    for (int lq = 0; lq < nloc; ++lq) {
        const int quad = q_lo + lq;
        float m1 = NEG_INF, m2 = NEG_INF, m3 = NEG_INF;   // the three largest approximate scores of the lane's 16 rows
        float n1 = NEG_INF, n2 = NEG_INF, n3 = NEG_INF;   // I8 = 3: the same for the second query group
        uint32_t ap = 0;                                  // rows (0..63) of the largest [7:0] and second largest [15:8]
        const char* nrow = norm_lds + ((lq & 1) << 8) + kg * 16;

#pragma unroll
        for (int t = 0; t < 4; ++t) {
            rq_float4 acc = {0.f, 0.f, 0.f, 0.f};
            rq_int4 iacc = {0, 0, 0, 0}, lacc = {0, 0, 0, 0};
#pragma unroll
            for (int kh = 0; kh < KL; ++kh) {
                const int st = lq * NSTQ + t * KL + kh;
                if (st + S - 2 <= nst - 1) rq_wait_vmcnt<VM_KEEP>(); else rq_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (st + S - 1 < nst) { issue(st + S - 1, islot); islot = (islot + 1 == S) ? 0 : islot + 1; }
                const char* sb = rq_smem + cslot * STAGE_BYTES;
                cslot = (cslot + 1 == S) ? 0 : cslot + 1;
#pragma unroll
                for (int g = 0; g < MF; g += PF) {
                    rq_half8 av[PF];
#pragma unroll
                    for (int s = 0; s < PF; ++s) av[s] = *(const rq_half8*)(sb + (rbase0 ^ (unsigned)(((g + s) & 3) << 6)) + (((g + s) & ~3) << 6));
#pragma unroll
                    for (int s = 0; s < PF; ++s) {
                        if constexpr (I8 != 0) {
                            iacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(rq_int4, av[s]), __builtin_bit_cast(rq_int4, qf[g + s]), iacc, 0, 0, 0);
                            if constexpr (I8 >= 2) lacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(rq_int4, av[s]), __builtin_bit_cast(rq_int4, ql[g + s]), lacc, 0, 0, 0);
                        } else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[s], qf[kh * MF + g + s], acc, 0, 0, 0);
                    }
                }
            }
            if constexpr (I8 != 0) {   // |sum| <= 768 * 127 * 127 < 2^24: the conversions are exact
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = I8 == 2 ? fmaf((float)iacc[i], 254.f, (float)lacc[i]) : (float)iacc[i];
            }
            // tile epilogue: D[row = 4*kg + i][query = r16]
            const rq_float4 nv = *(const rq_float4*)(nrow + t * 64);
            const int64_t row0 = (int64_t)quad * RQ_QUAD_ROWS + t * RQ_TILE_ROWS + 4 * kg;
            if (EPI == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // (int8: the score is an exact int32 sum times a finite row scale, or NaN on pad rows -- no clamp needed, rq_device.h)
                    if constexpr (I8 != 0) rq_insert3(m1, m2, m3, rq_pos_score_finite(acc[i] * nv[i], (uint32_t)(t * 16 + i)));
                    else rq_insert3(m1, m2, m3, rq_pos_score(acc[i] * nv[i], (uint32_t)(t * 16 + i)));
                }
                if constexpr (I8 == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rq_insert3(n1, n2, n3, rq_pos_score_finite((float)lacc[i] * nv[i], (uint32_t)(t * 16 + i)));
                }
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sc = acc[i] * nv[i];
                sc = (row0 + i < a.n_rows) ? sc : NEG_INF;
                // comparisons are false for NaN: NaN scores are dropped.  Ties count as separate rows: a tie with m1
                // becomes m2 (with its own position), a tie with m2 becomes m3.
                const bool gt1 = sc > m1, gt2 = sc > m2;
                const uint32_t pos = (uint32_t)(t * 16 + i);
                m3 = gt2 ? m2 : fmaxf(m3, sc);
                ap = gt1 ? ((ap << 8) | pos) : (gt2 ? ((ap & 0xffu) | (pos << 8)) : ap);
                m2 = gt1 ? m1 : fmaxf(m2, sc);
                m1 = gt1 ? sc : m1;
            }
        }
        if (EPI == 1) {
            // per query group: the query's scale (int8; positive, so the order is unchanged and the positions ride through it), the
            // lane's row group into the positions, then the four lanes that share the query insert each other's triples
            auto finish = [&](float x1, float x2, float x3, const float qs, float& wm, const int ql) {
                if constexpr (I8 != 0) { x1 = rq_scale_pos(x1, qs); x2 = rq_scale_pos(x2, qs); x3 = rq_scale_pos(x3, qs); }
                const uint32_t kgb = (uint32_t)kg << 2;
                x1 = __uint_as_float(__float_as_uint(x1) | kgb); x2 = __uint_as_float(__float_as_uint(x2) | kgb); x3 = __uint_as_float(__float_as_uint(x3) | kgb);
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float o1 = __shfl_xor(x1, off, 64), o2 = __shfl_xor(x2, off, 64), o3 = __shfl_xor(x3, off, 64);
                    rq_insert3(x1, x2, x3, o1);
                    rq_insert3(x1, x2, x3, o2);
                    rq_insert3(x1, x2, x3, o3);
                }
                asm("v_max_f32 %0, %1, %2" : "=v"(wm) : "v"(wm), "v"(x1));
                if (kg == 0) stg[ql * SQ + (lq & (SQ - 1))] = rq_record_from_triple(x1, x2, x3);
            };
            finish(m1, m2, m3, qsc, wmax, 16 * QG * wave + r16);
            if constexpr (I8 == 3) finish(n1, n2, n3, qsc1, wmax1, 16 * QG * wave + 16 + r16);
        } else {
        // merge the four lane groups that share this query (lanes r16, r16+16, r16+32, r16+48): all end up equal
        ap = (ap & 0xffffu) + (uint32_t)(4 * kg) * 0x0101u;
        // merge the sorted triples of the two lists (this lane's and the other lane's): both lanes compute the same
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float o1 = __shfl_xor(m1, off, 64), o2 = __shfl_xor(m2, off, 64), o3 = __shfl_xor(m3, off, 64);
            const uint32_t op = (uint32_t)__shfl_xor((int)ap, off, 64);
            // W = the list whose head wins (ties: smaller row), L = the other one
            const bool ow = o1 > m1 || (o1 == m1 && (op & 0xffu) < (ap & 0xffu));
            const float w1 = ow ? o1 : m1, w2 = ow ? o2 : m2, w3 = ow ? o3 : m3;
            const float l1 = ow ? m1 : o1, l2 = ow ? m2 : o2;
            const uint32_t wp = ow ? op : ap, lp = ow ? ap : op;
            const bool tl = l1 > w2 || (l1 == w2 && (lp & 0xffu) < (wp >> 8));   // L's head is the second largest
            m1 = w1;
            m2 = tl ? l1 : w2;
            m3 = tl ? fmaxf(w2, l2) : fmaxf(w3, l1);
            ap = (wp & 0xffu) | (tl ? (lp & 0xffu) << 8 : (wp & 0xff00u));
        }
        wmax = fmaxf(wmax, m1);
        if (kg == 0) {   // the four lane groups hold the same record
            const uint32_t c2 = rq_code16(m2), c3 = rq_code16(m3), d = c2 - c3;   // c3 <= c2
            stg[(16 * wave + r16) * SQ + (lq & (SQ - 1))] =
                make_uint2(rq_up26(m1) | (ap & 63u), (c2 << 16) | ((d < 1023u ? d : 1023u) << 6) | ((ap >> 8) & 63u));
        }
        }
        if ((lq & (SQ - 1)) == SQ - 1 || lq == nloc - 1)
            flush(q_lo + (lq & ~(SQ - 1)), (lq & (SQ - 1)) + 1);
    }
    // per-workgroup maximum of every query: wgmax[query][workgroup]
    wmax = fmaxf(wmax, __shfl_xor(wmax, 16, 64));
    wmax = fmaxf(wmax, __shfl_xor(wmax, 32, 64));
    if (kg == 0 && 16 * QG * wave + r16 < a.nq_valid) a.wgmax[(int64_t)(16 * QG * wave + r16) * a.wgmax_stride + b] = wmax;
    if constexpr (I8 == 3) {
        wmax1 = fmaxf(wmax1, __shfl_xor(wmax1, 16, 64));
        wmax1 = fmaxf(wmax1, __shfl_xor(wmax1, 32, 64));
        if (kg == 0 && 16 * QG * wave + 16 + r16 < a.nq_valid) a.wgmax[(int64_t)(16 * QG * wave + 16 + r16) * a.wgmax_stride + b] = wmax1;
    }
}

// ring + [2 parities][64 row scales] + record staging [16 * QW queries][rq_stage_quads(QW)]
static constexpr size_t rq_scan_lds_bytes(int S, int KS, int QW, int QG = 1) {
    return (size_t)S * (24576 / KS) + 512 + (size_t)16 * QW * QG * rq_stage_quads(QW) * 8;
}

// EPI of the kernels: 0 / 1 = selection form of the fp16 scan, 2 = the int8 scan (selection form 1), 3 = int8 with split queries,
// 4 = int8 with two query groups per wave (128 queries per pass)
template <int S, bool NT, int PF, int OCC, int KS, int QW, int EPI>
__global__ __launch_bounds__(64 * QW, OCC) void rq_scan_kernel(RqScanArgs a) {
    rq_scan_body<S, NT, PF, KS, QW, (EPI >= 2 ? 1 : EPI), (EPI >= 2 ? EPI - 1 : 0)>(a, (int)blockIdx.x, (int)gridDim.x);
}

// Fused launch: workgroups [0, scan_grid) scan the corpus for THIS batch, the others run the tail (threshold, fp64
// re-score, final top-k) of the PREVIOUS batch of the same stream, whose scan finished with the previous launch.
// One stream, no events: the tail's ~20 us hide under the scan, and the scan launches of consecutive batches never
// overlap each other.  Scan variant: ring of 3 half-row stages, prefetch 1 (36 KB ring + 16.5 KB of record staging =
// 53 760 B of LDS, <= 168 VGPRs), so a CU holds 2 scan workgroups + 1 tail workgroup (3 x 53 760 B <= 160 KB; the
// tail's 16 KB are carved from the ring).
template <bool NT, int NV, int EPI>
__global__ __launch_bounds__(256, 3) void rq_scan_tail_kernel(RqScanArgs sa, RqTailArgs ta, RqPrepArgs pa, int scan_grid, int tail_chunks) {
    unsigned long long t0 = 0;
    if (ta.dbg) t0 = wall_clock64();
    // block ids: [0, scan_grid) scan THIS batch | the tail workgroups of the PREVIOUS batch | pa.nslots workgroups that prepare
    // the queries of the NEXT batch (rq_search_hint_next_device), so that the next call needs no preparation launch
    const int bid = (int)blockIdx.x;
    const int nprep = pa.nslots;
    const int ntail = (int)gridDim.x - scan_grid - nprep;
    if (bid < scan_grid) {
        rq_scan_body<3, NT, 1, 2, 4, (EPI >= 2 ? 1 : EPI), (EPI >= 2 ? EPI - 1 : 0)>(sa, bid, scan_grid);
    } else if (bid < scan_grid + ntail) {
        const int t = bid - scan_grid;
        rq_tail_body<NV>(ta, t % tail_chunks, t / tail_chunks, tail_chunks, *reinterpret_cast<RqTailLds*>(rq_smem));
    } else {
        rq_prep_body(pa, bid - scan_grid - ntail, reinterpret_cast<double*>(rq_smem));
    }
    if (ta.dbg && threadIdx.x == 0) {
        ta.dbg[4 * blockIdx.x] = t0; ta.dbg[4 * blockIdx.x + 1] = wall_clock64();
        // HW_REG_HW_ID (4): wave/simd/cu/sh/se ids;  HW_REG_XCC_ID (20): the XCD
        ta.dbg[4 * blockIdx.x + 2] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        ta.dbg[4 * blockIdx.x + 3] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

template <int S, bool NT, int PF, int OCC, int KS, int QW, int EPI>
static hipError_t rq_scan_launch_t(const RqScanArgs& a, int grid, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = rq_scan_lds_bytes(S, KS, QW, EPI == 4 ? 2 : 1);
    static unsigned long long attr_done = 0;   // one bit per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!((attr_done >> (dev & 63)) & 1ull)) {
        e = hipFuncSetAttribute((const void*)rq_scan_kernel<S, NT, PF, OCC, KS, QW, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done |= 1ull << (dev & 63);
    }
    // e0 / e1: events attached to the dispatch itself (hipExtLaunchKernel): they read the kernel's own start and end
    // time stamps and cost no extra barrier packets (hipEventRecord around a launch costs ~5 us on each side)
    if (e0 && e1) hipExtLaunchKernelGGL((rq_scan_kernel<S, NT, PF, OCC, KS, QW, EPI>), dim3(grid), dim3(64 * QW), (uint32_t)lds, stream, e0, e1, 0, a);
    else hipLaunchKernelGGL((rq_scan_kernel<S, NT, PF, OCC, KS, QW, EPI>), dim3(grid), dim3(64 * QW), lds, stream, a);
    return hipGetLastError();
}

template <int S, int PF, int OCC, int KS, int QW, int EPI = 0>
static hipError_t rq_scan_launch_r(const RqScanArgs& a, bool nt, int grid, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    return nt ? rq_scan_launch_t<S, true, PF, OCC, KS, QW, EPI>(a, grid, stream, e0, e1) : rq_scan_launch_t<S, false, PF, OCC, KS, QW, EPI>(a, grid, stream, e0, e1);
}

// (ring S, prefetch PF, stages-per-tile KS, waves QW) combinations that are built; anything else is an error.
// qw = 4: 64 queries per pass; qw = 8: 128 queries per pass (one workgroup per CU, whole-row stages).
// epi = 1 (selection with positions inside the scores) exists for the default variant (ring 3, prefetch 1, half-row stages, 4 waves)
hipError_t rq_scan_launch(const RqScanArgs& a, int S, int pf, int ks, int qw, bool nt, int grid, int epi, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    if (grid <= 0) return hipErrorInvalidValue;
    if (a.i8 == 3) return (S == 3 && pf == 1 && ks == 2 && qw == 4) ? rq_scan_launch_r<3, 1, 2, 2, 4, 4>(a, nt, grid, stream, e0, e1) : hipErrorInvalidValue;
    if (a.i8 == 2) return (S == 3 && pf == 1 && ks == 2 && qw == 4) ? rq_scan_launch_r<3, 1, 3, 2, 4, 3>(a, nt, grid, stream, e0, e1) : hipErrorInvalidValue;
    if (a.i8) return (S == 3 && pf == 1 && ks == 2 && qw == 4) ? rq_scan_launch_r<3, 1, 3, 2, 4, 2>(a, nt, grid, stream, e0, e1) : hipErrorInvalidValue;
    if (epi && S == 3 && pf == 1 && ks == 2 && qw == 4) return rq_scan_launch_r<3, 1, 3, 2, 4, 1>(a, nt, grid, stream, e0, e1);
#define RQ_CASE(SS, PP, OO, KK, QQ) if (S == SS && pf == PP && ks == KK && qw == QQ) return rq_scan_launch_r<SS, PP, OO, KK, QQ>(a, nt, grid, stream, e0, e1);
    RQ_CASE(3, 1, 3, 2, 4) RQ_CASE(4, 1, 3, 2, 4)
    RQ_CASE(4, 4, 2, 2, 4) RQ_CASE(6, 4, 2, 2, 4)
    RQ_CASE(5, 6, 2, 2, 4) RQ_CASE(6, 12, 2, 2, 4)
    RQ_CASE(2, 4, 2, 1, 4) RQ_CASE(3, 4, 2, 1, 4) RQ_CASE(3, 12, 2, 1, 4) RQ_CASE(2, 1, 2, 1, 4) RQ_CASE(4, 4, 2, 1, 4)
    RQ_CASE(3, 4, 2, 1, 8)
#undef RQ_CASE
    return hipErrorInvalidValue;
}


// ---- fused scan(batch i) + tail(batch i-1) -------------------------------------------------------------------
template <bool NT, int NV, int EPI>
static hipError_t rq_scan_tail_launch_t(const RqScanArgs& sa, const RqTailArgs& ta, int tail_B, const RqPrepArgs& pa, int scan_grid, hipStream_t stream,
                                        hipEvent_t e0, hipEvent_t e1) {
    constexpr size_t lds = rq_scan_lds_bytes(3, 2, 4);
    static_assert(sizeof(RqTailLds) <= lds, "tail LDS must fit in the scan's LDS");
    static_assert(3 * lds <= 160 * 1024, "2 scan workgroups + 1 tail workgroup per CU");
    const int64_t chunks = (ta.nbins + 512 * NV - 1) / (512 * NV);
    if (chunks < 1 || chunks * tail_B > (1 << 24)) return hipErrorInvalidValue;
    static unsigned long long attr_done = 0;   // one bit per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!((attr_done >> (dev & 63)) & 1ull)) {
        e = hipFuncSetAttribute((const void*)rq_scan_tail_kernel<NT, NV, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done |= 1ull << (dev & 63);
    }
    const unsigned grid = (unsigned)(scan_grid + chunks * tail_B + pa.nslots);
    if (e0 && e1) hipExtLaunchKernelGGL((rq_scan_tail_kernel<NT, NV, EPI>), dim3(grid), dim3(256), (uint32_t)lds, stream, e0, e1, 0, sa, ta, pa, scan_grid, (int)chunks);
    else hipLaunchKernelGGL((rq_scan_tail_kernel<NT, NV, EPI>), dim3(grid), dim3(256), lds, stream, sa, ta, pa, scan_grid, (int)chunks);
    return hipGetLastError();
}

template <int EPI>
static hipError_t rq_scan_tail_launch_e(const RqScanArgs& sa, const RqTailArgs& ta, int tail_B, const RqPrepArgs& pa, bool nt, int scan_grid, hipStream_t stream,
                                        hipEvent_t e0, hipEvent_t e1) {
    if (scan_grid <= 0 || tail_B < 0) return hipErrorInvalidValue;   // tail_B = 0: development (the fused kernel without tail workgroups)
    if (ta.m < 1 || ta.m > RQ_FAST_MAX_M || ta.k < 1 || ta.k > RQ_FAST_MAX_K) return hipErrorInvalidValue;
    // Riding tails: as few workgroups as keep every CU's third slot busy once (~256): each tail workgroup costs the
    // scan a little while it is resident (measured at 1M rows: 256 workgroups of 4096 bins 248.7 us per launch, 512 of
    // 2048 bins 252.4 us).  The stand-alone launch (rq_tail_launch) prefers more, smaller ones: lower latency.
    const auto wgs = [&](int nv) { return ((ta.nbins + 512 * nv - 1) / (512 * nv)) * tail_B; };
    const int nv = (ta.fused_nv == 1 || ta.fused_nv == 4 || ta.fused_nv == 8 || ta.fused_nv == 16) ? ta.fused_nv : (wgs(1) <= 384 ? 1 : (wgs(4) <= 384 ? 4 : 8));
    if (nt && nv == 16) return rq_scan_tail_launch_t<true, 16, EPI>(sa, ta, tail_B, pa, scan_grid, stream, e0, e1);   // development (fused_nv = 16): 128 riding workgroups at 1M rows
    if (nt) return nv == 1 ? rq_scan_tail_launch_t<true, 1, EPI>(sa, ta, tail_B, pa, scan_grid, stream, e0, e1)
                 : nv == 4 ? rq_scan_tail_launch_t<true, 4, EPI>(sa, ta, tail_B, pa, scan_grid, stream, e0, e1)
                           : rq_scan_tail_launch_t<true, 8, EPI>(sa, ta, tail_B, pa, scan_grid, stream, e0, e1);
    return nv == 1 ? rq_scan_tail_launch_t<false, 1, EPI>(sa, ta, tail_B, pa, scan_grid, stream, e0, e1)
         : nv == 4 ? rq_scan_tail_launch_t<false, 4, EPI>(sa, ta, tail_B, pa, scan_grid, stream, e0, e1)
                   : rq_scan_tail_launch_t<false, 8, EPI>(sa, ta, tail_B, pa, scan_grid, stream, e0, e1);
}

// epi: selection form of the scan workgroups (0 = compare / select, 1 = positions inside the scores)
hipError_t rq_scan_tail_launch(const RqScanArgs& sa, const RqTailArgs& ta, int tail_B, const RqPrepArgs& pa, bool nt, int scan_grid, int epi, hipStream_t stream,
                               hipEvent_t e0, hipEvent_t e1) {
    if (pa.nslots < 0 || pa.nslots > 64) return hipErrorInvalidValue;
    if (sa.i8 == 3) return hipErrorInvalidValue;   // the 128-query form is not fused with a tail (LDS)
    if (sa.i8 == 2) return rq_scan_tail_launch_e<3>(sa, ta, tail_B, pa, nt, scan_grid, stream, e0, e1);
    if (sa.i8) return rq_scan_tail_launch_e<2>(sa, ta, tail_B, pa, nt, scan_grid, stream, e0, e1);
    return epi ? rq_scan_tail_launch_e<1>(sa, ta, tail_B, pa, nt, scan_grid, stream, e0, e1)
               : rq_scan_tail_launch_e<0>(sa, ta, tail_B, pa, nt, scan_grid, stream, e0, e1);
}
