// rq_scan_wide.hip -- the corpus scan for passes of MORE than 64 queries (128 per pass): same arithmetic, records and
// contracts as rq_scan.hip (which keeps the 64-query headline kernel), different schedule.
//
// Why a second schedule.  With 128 queries per pass every corpus byte feeds twice the matrix work, so a CU has to issue
// 2 x the MFMAs and LDS fragment reads per HBM byte while still streaming at the HBM rate.  rq_scan.hip's loop is
// "wait for the stage, barrier, then for every fragment: ds_read -> wait -> MFMA": the LDS latency of the first
// fragments after every barrier and of every read group is exposed, all waves of the (single) workgroup of a CU hit it at
// the same moment, and round 1's 8-wave variant measured 332-355 us per 128-query pass at 1M rows against 240 us for
// the 64-query pass (1.45x per corpus byte instead of 2x).  Here:
//   * a stage is one TILE (16 whole rows, 24 KiB, contiguous in HBM); the ring has 4 slots = the 4 tiles of a quad, so
//     every LDS address in the unrolled quad loop is static;
//   * the LDS fragment reads run D fragments ahead of the MFMAs in one continuous stream that does NOT stop at stage
//     boundaries: a stage is acquired (counted vmcnt for the wave's own LDS-DMA pieces, one raw s_barrier) when the READ
//     pointer enters it, i.e. D fragments before the MFMA pointer leaves the previous one, so the barrier and the first
//     reads of a stage hide under the MFMAs of the stage before;
//   * the slot refilled at that barrier is the one of the stage two back (fully consumed by every wave), which keeps two
//     stages (48 KiB per CU) in flight like the 64-query kernel does with its two workgroups per CU;
//   * QG = 2: a wave keeps 2 x 16 queries in registers (192 VGPRs, one wave per SIMD) and every fragment it reads feeds
//     two MFMAs -- half the LDS reads per query of the 8-wave form.
//   * I8 (round 3): the same schedule over the int8 image of the shard (rq_scan.hip I8) for 256 queries per
//     pass: a stage is still 24 KiB but holds TWO tiles (32 rows of 768 B), a quad is two stages, the ring of 4 slots two
//     quads (slot = 2 * (quad parity) + stage); v_mfma_i32_16x16x64_i8, 12 k-steps per tile, so a stage is 24 fragments in
//     both forms and the read-ahead stream is the same code.  The query fragments of 2 groups take 96 VGPRs instead of 192:
//     the reads run 12 fragments ahead where the fp16 256-query form has registers for 2.
// Replaces the arithmetic of reference rag_uq/streaming_index.py:355-359 (collection.query), like rq_scan.hip.
#include <hip/hip_ext.h>

#include "rq_device.h"
#include "rq_kernels.h"

extern __shared__ __attribute__((aligned(16))) char rq_smem_w[];

typedef __attribute__((address_space(3))) void* lds_ptr_w;
typedef const __attribute__((address_space(1))) void* glb_ptr_w;

template <int N>
__device__ __forceinline__ void rqw_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// records per query parked in LDS before they are written out (32 KiB of staging for 128 queries, 32 KiB for 256)
__host__ __device__ static constexpr int rqw_sq(int QW, int QG) { return QW * QG >= 16 ? 16 : 32; }

// NT: non-temporal corpus loads; D: fragments the LDS reads run ahead of the MFMAs (2..12);
// QW: waves per workgroup; QG: 16-query groups per wave.  Queries per pass = 16 * QW * QG.
// EPI: 1 = the selection keeps the row position in the 6 low mantissa bits of the score (v_med3 inserts, 6 VALU per score),
//      0 = rq_scan.hip's compare/select form (14 VALU per score).
// DBG (timing experiments only, results invalid): 1 = no selection epilogue, 2 = no LDS fragment reads, 3 = no MFMAs
template <bool NT, int D, int QW, int QG, int EPI, int DBG = 0, int PRIO = 0, int I8 = 0>
__device__ __forceinline__ void rq_scanw_body(const RqScanArgs& a, const int b, const int G) {
    static_assert(D >= 2 && D <= 12 && 24 % D == 0, "prefetch distance");
    static_assert(QW == 4 || QW == 8, "waves per workgroup");
    static_assert(QG == 1 || QG == 2 || QG == 4, "query groups per wave");
    static_assert(I8 == 0 || (EPI == 1 && DBG == 0 && PRIO != 2), "int8 form: med3 selection only");
    constexpr int ROWB = I8 ? RQ_DPAD : RQ_DPAD * 2;   // bytes per corpus row
    constexpr int KS = I8 ? 12 : 24;               // k-steps = LDS fragments per tile
    constexpr int TPS = I8 ? 2 : 1;                // tiles per stage
    constexpr int SPQ = 4 / TPS;                   // stages per quad
    constexpr int FQ = 4 * KS;                     // fragments per quad
    constexpr int CH = ROWB / 16;                  // 16-byte chunks per row
    constexpr int TILE_BYTES = 16 * ROWB;
    constexpr int STAGE_BYTES = 24576;             // 16 rows x 1536 B, or 32 rows x 768 B
    static_assert(FQ % D == 0, "the fragment ring must close over a quad");
    constexpr int DPW = 24 / QW;                   // LDS-DMA wave-instructions (1 KiB each) per wave per stage
    constexpr int VM_KEEP = DPW;                   // at an acquire, the wave's pieces of the NEXT stage may stay in flight
    constexpr unsigned AUX = NT ? 2u : 0u;
    constexpr int SQ = rqw_sq(QW, QG);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = lane >> 4;          // k-group of the MFMA operand / row group of the result
    const int r16 = lane & 15;         // corpus row inside the tile (A operand), query inside the group (D)

    // per-lane DMA source offsets: LDS chunk p = 64*j + lane of a stage holds row r = p / CH, source chunk (p % CH) ^ (r & 15)
    unsigned voff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int p = 64 * (wave * DPW + i) + lane;
        const int r = p / CH, cp = p % CH;
        voff[i] = (unsigned)(r * ROWB + ((cp ^ (r & 15)) << 4));
    }
    // per-lane LDS read offset of k-step s: (rbase0 ^ ((s & 3) << 6)) + ((s & ~3) << 6)   (see rq_scan.hip)
    static_assert(ROWB % 256 == 0, "row pitch must keep bits 6..7 free");
    const unsigned rbase0 = (unsigned)(r16 * ROWB + ((kg ^ (r16 & 3)) << 4) + ((r16 >> 2) << 6));

    const int q_lo = (int)((int64_t)b * a.nquads / G);
    const int nloc = (int)((int64_t)(b + 1) * a.nquads / G) - q_lo;
    const int nst = nloc * SPQ;        // stages of this workgroup's quads
    const char* xb = (const char*)a.x;
    // Row scales of a quad ride with the quad's first stage, which is issued TWO stages ahead.  fp16 (4 stages per quad): that is
    // during the quad before -> 2 buffers.  int8 (2 stages per quad): that is near the end of the quad TWO before, whose last tile
    // still reads its own scales -> 4 buffers (round 3: with 2, the slowest wave of a workgroup scored the last tile of quads 0 and 1
    // with the scales of quads 2 and 3 now and then -- found because its bin maxima differed from the 128-query kernel's).
    constexpr int NPAR = I8 ? 4 : 2;
    char* norm_lds = rq_smem_w + 4 * STAGE_BYTES;                // [NPAR][64 row scales]
    uint2* const stg = (uint2*)(norm_lds + 1024);                // [16 * QW * QG queries][SQ] finished records

    // stage gs = stage (gs % SPQ) of local quad (gs / SPQ) -> ring slot (gs & 3).  Stages are issued strictly in order, so the
    // source addresses are two running (wave-uniform) pointers instead of 64-bit multiplications per stage.
    const char* gnext = xb + (int64_t)q_lo * (RQ_QUAD_ROWS * ROWB);                  // corpus bytes of the next stage
    const float* nsnext = a.row_scale + (int64_t)q_lo * RQ_QUAD_ROWS + lane;       // row scales of the next quad
    auto issue = [&](int gs) {
        const int t = gs & 3;
        char* l = rq_smem_w + t * STAGE_BYTES + (wave * DPW) * 1024;
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_w)(gnext + voff[i]), (lds_ptr_w)(l + i * 1024), 16, 0, AUX);
        gnext += STAGE_BYTES;
        if ((gs & (SPQ - 1)) == 0) {   // row scales of the quad (256 B); visible to all waves after wave 0's wait + a barrier
            if (wave == 0) __builtin_amdgcn_global_load_lds((glb_ptr_w)nsnext, (lds_ptr_w)(norm_lds + (((gs / SPQ) & (NPAR - 1)) << 8)), 4, 0, 0);
            nsnext += RQ_QUAD_ROWS;
        }
    };
    // the read pointer enters stage gs: its bytes must have landed for every wave; the slot of stage gs - 2 is free
    auto acquire = [&](int gs) {
        if (gs + 1 < nst) rqw_wait_vmcnt<VM_KEEP>(); else rqw_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (gs + 2 < nst) issue(gs + 2);
    };

    issue(0);
    if (nst > 1) issue(1);

    const float NEG_INF = -__builtin_huge_valf();
    float wmax[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) wmax[g] = NEG_INF;

    // query fragments: fp16  B[k = 8*kg + j][col = r16] of k-step s == qh[16*(QG*wave + g) + r16][32*s + 8*kg + j]
    //                  int8  B[k = 16*kg + j][col = r16] of k-step s == q8[16*(QG*wave + g) + r16][64*s + 16*kg + j]   (16 bytes per fragment either way)
    rq_half8 qf[QG][KS];
    float qsc[QG];      // int8: s_q / |q| of this lane's query (rq_prep_body), applied once per quad
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const rq_half8* qsrc = (const rq_half8*)((const char*)a.qh + (size_t)(16 * (QG * wave + g) + r16) * ROWB + 16 * kg);
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[g][s] = qsrc[4 * s];
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[g][s]));   // ordinary loads retired before the main loop
        qsc[g] = I8 ? a.qscale[16 * (QG * wave + g) + r16] : 1.f;
    }

    // PRIO == 3: the fragment reads are asm (ds_read_b128 with immediate offsets from 8 per-lane base addresses) and every
    // MFMA group waits with a COUNTED s_waitcnt lgkmcnt(D): exactly D younger fragment reads may still be in flight.  The
    // compiler's own schedule drains the whole LDS queue (lgkmcnt(0)) every few fragments, which exposes the latency of the
    // read it has just issued.  LDS operations return in order, so other LDS instructions (the compiler's: row scales,
    // record staging, shuffles) between the asm reads can only make a counted wait stricter, never too lax.
    unsigned abase[2][4];   // [ring half: slots 0-1 / 2-3][s & 3]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        abase[0][m] = (unsigned)(size_t)(lds_ptr_w)rq_smem_w + (rbase0 ^ (unsigned)(m << 6));
        abase[1][m] = abase[0][m] + 2 * STAGE_BYTES;
    }
    // fp16: tile t of a quad sits in slot t (static).  int8: tile t sits in slot 2 * parity + t / 2 at offset (t & 1) * TILE_BYTES,
    // parity = that quad's (lq & 1): `bc` = per-lane bases of the quad being multiplied, `bn` = of the quad after it.
    unsigned bc[4], bn[4];
    auto set_bases = [&](int lq) {
        const unsigned cur = (unsigned)((lq & 1) * 2 * STAGE_BYTES), nxt = (unsigned)(((lq + 1) & 1) * 2 * STAGE_BYTES);
#pragma unroll
        for (int m = 0; m < 4; ++m) { bc[m] = abase[0][m] + cur; bn[m] = abase[0][m] + nxt; }
    };
    auto frag = [&](int t, int s, bool next_quad) -> rq_half8 {
        if (DBG == 2) { rq_half8 z; asm volatile("" : "=v"(z)); return z; }
        if (I8) {
            rq_half8 v;
            const unsigned base = next_quad ? bn[s & 3] : bc[s & 3];
            if (PRIO == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"((t >> 1) * STAGE_BYTES + (t & 1) * TILE_BYTES + ((s & ~3) << 6)));
            else v = *(const rq_half8*)((const char*)(__attribute__((address_space(3))) const char*)(size_t)base + (t >> 1) * STAGE_BYTES + (t & 1) * TILE_BYTES + ((s & ~3) << 6));
            return v;
        }
        if (PRIO == 3) {
            rq_half8 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(abase[t >> 1][s & 3]), "n"((t & 1) * STAGE_BYTES + ((s & ~3) << 6)));
            return v;
        }
        return *(const rq_half8*)(rq_smem_w + t * STAGE_BYTES + (rbase0 ^ (unsigned)((s & 3) << 6)) + ((s & ~3) << 6));
    };
    int nq_valid_r = 0;          // (set below, once the kernel arguments are pinned in registers)
    int64_t bins_stride_r = 0;
    uint2* bins_r = nullptr;
    auto flush = [&](int quad0, int count) {
        // each wave writes the rows of its own 16 QG queries: 64 / SQ queries x SQ records (8 SQ-byte runs) per instruction
        constexpr int QPI = 64 / SQ;
#pragma unroll 1
        for (int i = 0; i < 16 * QG / QPI; ++i) {
            const int ql = 16 * QG * wave + QPI * i + lane / SQ, j = lane & (SQ - 1);
            if (j < count && ql < nq_valid_r) bins_r[(int64_t)ql * bins_stride_r + quad0 + j] = stg[ql * SQ + j];
        }
    };

    // static priority for the second-dispatched half of an 8-wave workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (PRIO == 1 && QW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    rq_half8 av[D];   // fragment ring: fragment f of the quad lives in av[f % D]
    if (I8) set_bases(0);
    // The counted lgkmcnt waits below are sound only while LDS operations are the ONLY users of that counter: LDS returns in
    // order, scalar (kernel-argument) loads do not, and one s_load still in flight when the first fragment reads are issued can
    // satisfy "at most D outstanding" in place of the oldest ds_read (round 3: the compiler had sunk the load of a.nq_valid to just
    // before the loop of the int8 form; queries of some waves then multiplied a stale register in their first tile).  Every
    // argument the loop and the epilogue use is therefore pinned in registers here, and the queue is drained once.
    int nq_valid = a.nq_valid, wgmax_stride = a.wgmax_stride;
    int64_t bins_stride = a.bins_stride;
    uint2* bins_p = a.bins;
    float* wgmax_p = a.wgmax;
    asm volatile("" : "+s"(nq_valid), "+s"(wgmax_stride), "+s"(bins_stride), "+s"(bins_p), "+s"(wgmax_p));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    nq_valid_r = nq_valid; bins_stride_r = bins_stride; bins_r = bins_p;
    acquire(0);
#pragma unroll
    for (int f = 0; f < D; ++f) av[f] = frag(f / KS, f % KS, false);

    for (int lq = 0; lq < nloc; ++lq) {
        const int quad = q_lo + lq;
        const bool more = lq + 1 < nloc;
        if (I8) set_bases(lq);
        float m1[QG], m2[QG], m3[QG];   // the three largest approximate scores of the lane's 16 rows, per query group
        uint32_t ap[QG];                // rows (0..63) of the largest [7:0] and second largest [15:8]
#pragma unroll
        for (int g = 0; g < QG; ++g) { m1[g] = NEG_INF; m2[g] = NEG_INF; m3[g] = NEG_INF; ap[g] = 0; }
        const char* nrow = norm_lds + ((lq & (NPAR - 1)) << 8) + kg * 16;
        // accumulators alternate between two register sets by tile parity: with PRIO == 2 the waves 4..7 of an 8-wave
        // workgroup run the selection of tile t after the first 12 MFMAs of tile t + 1 (tiles 0..2; tile 3's at the end of the
        // quad), so that of the two waves that share a SIMD one is in its VALU epilogue while the other feeds the matrix
        // core instead of both doing the same thing at the same time (MI355X_MICROARCH.md, two waves per SIMD, item 9)
        rq_float4 acc2[2][QG];
        rq_int4 iacc2[2][QG];
        const bool late = PRIO == 2 && QW == 8 && wave >= 4;
        auto select_tile = [&](int t) {
            const rq_float4 nv = *(const rq_float4*)(nrow + t * 64);
#pragma unroll
            for (int g = 0; g < QG; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rq_insert3(m1[g], m2[g], m3[g], rq_pos_score(acc2[t & 1][g][i] * nv[i], (uint32_t)(t * 16 + i)));
                }
        };
#pragma clang loop unroll(full)
        for (int t = 0; t < 4; ++t)
#pragma clang loop unroll(full)
        for (int s = 0; s < KS; ++s) {
            const int f = t * KS + s;
            const rq_half8 cur = av[f % D];
            // the read pointer: fragment f + D, D fragments ahead, crossing stage (and quad) boundaries
            {
                const int r = f + D;
                if (r < FQ) {
                    if (r % 24 == 0) acquire(lq * SPQ + r / 24);
                    av[f % D] = frag(r / KS, r % KS, false);
                } else {
                    // the next quad's first fragments; after the workgroup's last quad the reads still run (unconditional
                    // code, no copies) on bytes nobody uses: nothing is in flight into that slot any more
                    if (r == FQ && more) acquire(lq * SPQ + SPQ);
                    av[f % D] = frag((r - FQ) / KS, (r - FQ) % KS, true);
                }
            }
            rq_half8 curw = cur;
            if (PRIO == 3) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(curw) : "n"(D));   // fragment f has landed; f+1 .. f+D may be in flight
            rq_float4 (&acc)[QG] = acc2[t & 1];
            rq_int4 (&iacc)[QG] = iacc2[t & 1];
            if (s == 0) {
#pragma unroll
                for (int g = 0; g < QG; ++g) { acc[g] = rq_float4{0.f, 0.f, 0.f, 0.f}; iacc[g] = rq_int4{0, 0, 0, 0}; }
            }
            if (DBG == 3) {
#pragma unroll
                for (int g = 0; g < QG; ++g) asm volatile("" : "+v"(acc[g]) : "v"(curw), "v"(qf[g][s]));
            } else if (I8) {
#pragma unroll
                for (int g = 0; g < QG; ++g)
                    iacc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(rq_int4, curw), __builtin_bit_cast(rq_int4, qf[g][s]), iacc[g], 0, 0, 0);
            } else {
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(curw, qf[g][s], acc[g], 0, 0, 0);
            }
            if (PRIO == 2 && EPI == 1 && DBG == 0) {
                if (s == 11 && t >= 1) { if (late) select_tile(t - 1); }
                if (s == KS - 1) { if (!late || t == 3) select_tile(t); }
            }
            if (s == KS - 1 && DBG == 1) {
#pragma unroll
                for (int g = 0; g < QG; ++g) m1[g] = fmaxf(m1[g], fmaxf(fmaxf(acc[g][0], acc[g][1]), fmaxf(acc[g][2], acc[g][3])));
            }
            if (s == KS - 1 && DBG != 1 && EPI == 1 && PRIO != 2) {
                // Tile epilogue, D[row = 4*kg + i][query = r16].  score = acc * row scale, clamped to the finite range (an
                // infinite score stays the largest, NaN becomes the smallest); its 6 low mantissa bits are then REPLACED by
                // the row's position in the quad (bits 4..5 tile, 0..1 register; the lane's row group 4*kg is OR-ed in
                // after the quad), so the three v_med3 inserts carry the positions along: 6 VALU per score and no
                // data-dependent code.  The perturbation (< 64 ulp, 7.6e-6 relative) is part of the scan's error bound
                // eps; every field of the record written below stays an UPPER bound of the unperturbed score.
                // Rows beyond the shard's end carry a NaN row scale (rq_api.hip) and therefore sort last.
                // int8: |sum| <= 768 * 127 * 127 < 2^24, the conversion to fp32 is exact; the query's scale follows once per quad.
                const rq_float4 nv = *(const rq_float4*)(nrow + t * 64);
#pragma unroll
                for (int g = 0; g < QG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (I8) rq_insert3(m1[g], m2[g], m3[g], rq_pos_score_finite((float)iacc[g][i] * nv[i], (uint32_t)(t * 16 + i)));
                        else rq_insert3(m1[g], m2[g], m3[g], rq_pos_score(acc[g][i] * nv[i], (uint32_t)(t * 16 + i)));
                    }
            }
            if (s == KS - 1 && DBG != 1 && EPI == 0) {
                // tile epilogue: D[row = 4*kg + i][query = r16]
                const rq_float4 nv = *(const rq_float4*)(nrow + t * 64);
                const int64_t row0 = (int64_t)quad * RQ_QUAD_ROWS + t * RQ_TILE_ROWS + 4 * kg;
#pragma unroll
                for (int g = 0; g < QG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float sc = acc[g][i] * nv[i];
                        sc = (row0 + i < a.n_rows) ? sc : NEG_INF;
                        // comparisons are false for NaN: NaN scores are dropped.  Ties count as separate rows.
                        const bool gt1 = sc > m1[g], gt2 = sc > m2[g];
                        const uint32_t pos = (uint32_t)(t * 16 + i);
                        m3[g] = gt2 ? m2[g] : fmaxf(m3[g], sc);
                        ap[g] = gt1 ? ((ap[g] << 8) | pos) : (gt2 ? ((ap[g] & 0xffu) | (pos << 8)) : ap[g]);
                        m2[g] = gt1 ? m1[g] : fmaxf(m2[g], sc);
                        m1[g] = gt1 ? sc : m1[g];
                    }
            }
        }
        if (EPI == 1) {
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                // (int8: the query's positive scale first -- the order is unchanged and the positions ride through it;) positions become
                // complete (row group of the lane), then the four lanes that share the query merge their
                // sorted triples: the other lane's three values are inserted one by one; all four lanes end up equal
                const uint32_t kgb = (uint32_t)kg << 2;
                float y1 = m1[g], y2 = m2[g], y3 = m3[g];
                if (I8) { y1 = rq_scale_pos(y1, qsc[g]); y2 = rq_scale_pos(y2, qsc[g]); y3 = rq_scale_pos(y3, qsc[g]); }
                float x1 = __uint_as_float(__float_as_uint(y1) | kgb), x2 = __uint_as_float(__float_as_uint(y2) | kgb),
                      x3 = __uint_as_float(__float_as_uint(y3) | kgb);
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float o1 = __shfl_xor(x1, off, 64), o2 = __shfl_xor(x2, off, 64), o3 = __shfl_xor(x3, off, 64);
                    rq_insert3(x1, x2, x3, o1);
                    rq_insert3(x1, x2, x3, o2);
                    rq_insert3(x1, x2, x3, o3);
                }
                asm("v_max_f32 %0, %1, %2" : "=v"(wmax[g]) : "v"(wmax[g]), "v"(x1));
                if (kg == 0) stg[(16 * (QG * wave + g) + r16) * SQ + (lq & (SQ - 1))] = rq_record_from_triple(x1, x2, x3);
            }
        } else
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            // merge the four lane groups that share this query (lanes r16, r16+16, r16+32, r16+48): all end up equal
            float x1 = m1[g], x2 = m2[g], x3 = m3[g];
            uint32_t xp = (ap[g] & 0xffffu) + (uint32_t)(4 * kg) * 0x0101u;
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float o1 = __shfl_xor(x1, off, 64), o2 = __shfl_xor(x2, off, 64), o3 = __shfl_xor(x3, off, 64);
                const uint32_t op = (uint32_t)__shfl_xor((int)xp, off, 64);
                const bool ow = o1 > x1 || (o1 == x1 && (op & 0xffu) < (xp & 0xffu));
                const float w1 = ow ? o1 : x1, w2 = ow ? o2 : x2, w3 = ow ? o3 : x3;
                const float l1 = ow ? x1 : o1, l2 = ow ? x2 : o2;
                const uint32_t wp = ow ? op : xp, lp = ow ? xp : op;
                const bool tl = l1 > w2 || (l1 == w2 && (lp & 0xffu) < (wp >> 8));
                x1 = w1;
                x2 = tl ? l1 : w2;
                x3 = tl ? fmaxf(w2, l2) : fmaxf(w3, l1);
                xp = (wp & 0xffu) | (tl ? (lp & 0xffu) << 8 : (wp & 0xff00u));
            }
            wmax[g] = fmaxf(wmax[g], x1);
            if (kg == 0) {
                const uint32_t c2 = rq_code16(x2), c3 = rq_code16(x3), d = c2 - c3;
                stg[(16 * (QG * wave + g) + r16) * SQ + (lq & (SQ - 1))] =
                    make_uint2(rq_up26(x1) | (xp & 63u), (c2 << 16) | ((d < 1023u ? d : 1023u) << 6) | ((xp >> 8) & 63u));
            }
        }
        if ((lq & (SQ - 1)) == SQ - 1 || lq == nloc - 1)
            flush(q_lo + (lq & ~(SQ - 1)), (lq & (SQ - 1)) + 1);
    }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int ql = 16 * (QG * wave + g) + r16;
        if (kg == 0 && ql < nq_valid) wgmax_p[(int64_t)ql * wgmax_stride + b] = wmax[g];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 256 queries per pass over the int8 image on v_mfma_i32_32x32x32_i8: a wave owns 32 queries as ONE operand.  Same ring, DMA, acquire
// points and record staging as the I8 form of rq_scanw_body (a stage = 32 rows of 768 B = exactly one 32-row A block, 24 k-steps of 32
// bytes; a quad = 2 stages; 4 row-scale buffers), but per wave and quad: half the MFMA instructions (each does twice the work), ONE
// cross-lane merge round instead of four (the two lanes l, l + 32 that share a query), one record instead of two -- the VALU issue the
// 16x16 form is bound by.  Result layout of the 32x32 forms: register i of lane l is D[row = 8 (i / 4) + 4 (l / 32) + i % 4]
// [query = l % 32].
// ---------------------------------------------------------------------------------------------------------------------------
typedef int rq_int16 __attribute__((ext_vector_type(16)));

// QS (A/B): the DMA unit is a whole quad (48 KiB, ring of 2) instead of a 32-row block (24 KiB, ring of 4): the same LDS addresses and the
// same bytes in flight, one workgroup barrier per quad instead of two.
template <bool NT, int D, bool QS = false>
__device__ __forceinline__ void rq_scanw32_body(const RqScanArgs& a, const int b, const int G) {
    static_assert(D >= 2 && D <= 12 && 48 % D == 0, "prefetch distance");
    constexpr int QW = 8, ROWB = RQ_DPAD, KS = 24, FQ = 48, CH = ROWB / 16, STAGE_BYTES = 24576, SPQ = QS ? 1 : 2;
    constexpr int DMA_BYTES = QS ? 2 * STAGE_BYTES : STAGE_BYTES;      // bytes of one DMA unit ("stage" of the issue / acquire logic)
    constexpr int FPS = QS ? 48 : 24;                                  // fragments per DMA unit
    constexpr int DPW = DMA_BYTES / 1024 / QW, VM_KEEP = DPW, SQ = 16, NPAR = 4;
    constexpr unsigned AUX = NT ? 2u : 0u;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31;         // corpus row inside the 32-row block (A operand), query inside the wave (B operand, result column)
    const int h = lane >> 5;           // k half of the operands / row sub-block of the result

    unsigned voff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int p = 64 * (wave * DPW + i) + lane;
        const int r = p / CH, cp = p % CH;
        voff[i] = (unsigned)(r * ROWB + ((cp ^ (r & 15)) << 4));
    }
    const int q_lo = (int)((int64_t)b * a.nquads / G);
    const int nloc = (int)((int64_t)(b + 1) * a.nquads / G) - q_lo;
    const int nst = nloc * SPQ;
    const char* xb = (const char*)a.x;
    char* norm_lds = rq_smem_w + 4 * STAGE_BYTES;
    uint2* const stg = (uint2*)(norm_lds + 1024);
    const char* gnext = xb + (int64_t)q_lo * (RQ_QUAD_ROWS * ROWB);
    const float* nsnext = a.row_scale + (int64_t)q_lo * RQ_QUAD_ROWS + lane;
    auto issue = [&](int gs) {
        const int t = QS ? (gs & 1) : (gs & 3);
        char* l = rq_smem_w + t * DMA_BYTES + (wave * DPW) * 1024;
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_w)(gnext + voff[i]), (lds_ptr_w)(l + i * 1024), 16, 0, AUX);
        gnext += DMA_BYTES;
        if ((gs & (SPQ - 1)) == 0) {
            if (wave == 0) __builtin_amdgcn_global_load_lds((glb_ptr_w)nsnext, (lds_ptr_w)(norm_lds + (((gs / SPQ) & (NPAR - 1)) << 8)), 4, 0, 0);
            nsnext += RQ_QUAD_ROWS;
        }
    };
    auto acquire = [&](int gs) {
        if (!QS && gs + 1 < nst) rqw_wait_vmcnt<VM_KEEP>(); else rqw_wait_vmcnt<0>();    // (QS: the next unit is only issued below, nothing younger is in flight)
        // QS refills the slot of the unit that was read LAST (ring of 2): every wave's fragment reads of it must have returned before
        // any wave may issue the DMA that overwrites it
        if (QS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (QS) { if (gs + 1 < nst) issue(gs + 1); }
        else if (gs + 2 < nst) issue(gs + 2);
    };
    issue(0);
    if (!QS && nst > 1) issue(1);

    // B operand: q8[32 wave + r32][32 s + 16 h .. + 15] of k-step s
    rq_int4 qf[KS];
    {
        const rq_int4* qsrc = (const rq_int4*)((const char*)a.qh + (size_t)(32 * wave + r32) * ROWB + 16 * h);
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = qsrc[2 * s];
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[s]));
    }
    const float qsc = a.qscale[32 * wave + r32];
    // A operand of k-step s: logical chunk 2 s + h of row r32, stored at chunk (2 s + h) ^ (r32 & 15): the low chunk bit is (h ^ r15 & 1),
    // bits 1..3 are ((s & 7) ^ (r15 >> 1)) -> one per-lane base per (s & 7), the rest of s is an immediate ((s & ~7) * 32 bytes)
    const int r15 = r32 & 15;
    unsigned lbase[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
        lbase[m] = (unsigned)(size_t)(lds_ptr_w)rq_smem_w + (unsigned)(r32 * ROWB + (((h ^ (r15 & 1)) + 2 * (m ^ (r15 >> 1))) << 4));
    // tile t (= stage t of the quad) sits in ring slot 2 * (quad parity) + t
    auto frag = [&](int t, int s, unsigned par_off) -> rq_int4 {
        return *(const rq_int4*)((const char*)(__attribute__((address_space(3))) const char*)(size_t)(lbase[s & 7] + par_off) + t * STAGE_BYTES + ((s & ~7) << 5));
    };
    float wmax = -__builtin_huge_valf();
    const float NEG_INF = -__builtin_huge_valf();
    int nq_valid = a.nq_valid, wgmax_stride = a.wgmax_stride;
    int64_t bins_stride = a.bins_stride;
    uint2* bins_p = a.bins;
    float* wgmax_p = a.wgmax;
    asm volatile("" : "+s"(nq_valid), "+s"(wgmax_stride), "+s"(bins_stride), "+s"(bins_p), "+s"(wgmax_p));
    auto flush = [&](int quad0, int count) {
        constexpr int QPI = 64 / SQ;
#pragma unroll 1
        for (int i = 0; i < 32 / QPI; ++i) {
            const int ql = 32 * wave + QPI * i + lane / SQ, j = lane & (SQ - 1);
            if (j < count && ql < nq_valid) bins_p[(int64_t)ql * bins_stride + quad0 + j] = stg[ql * SQ + j];
        }
    };

    rq_int4 av[D];
    acquire(0);
#pragma unroll
    for (int f = 0; f < D; ++f) av[f] = frag(f / KS, f % KS, 0u);

    for (int lq = 0; lq < nloc; ++lq) {
        const bool more = lq + 1 < nloc;
        const unsigned pc = (unsigned)((lq & 1) * 2 * STAGE_BYTES), pn = (unsigned)(((lq + 1) & 1) * 2 * STAGE_BYTES);
        float m1 = NEG_INF, m2 = NEG_INF, m3 = NEG_INF;
        const char* nrow = norm_lds + ((lq & (NPAR - 1)) << 8) + h * 16;
        rq_int16 acc2[2];
#pragma clang loop unroll(full)
        for (int t = 0; t < 2; ++t)
#pragma clang loop unroll(full)
        for (int s = 0; s < KS; ++s) {
            const int f = t * KS + s;
            const rq_int4 cur = av[f % D];
            {
                const int r = f + D;
                if (r < FQ) {
                    if (r % FPS == 0) acquire(lq * SPQ + r / FPS);
                    av[f % D] = frag(r / KS, r % KS, pc);
                } else {
                    if (r == FQ && more) acquire(lq * SPQ + SPQ);
                    av[f % D] = frag((r - FQ) / KS, (r - FQ) % KS, pn);
                }
            }
            rq_int16 (&acc) = acc2[t];
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0;
            }
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur, qf[s], acc, 0, 0, 0);
            if (s == KS - 1) {
                // block epilogue: register i is row 8 (i / 4) + 4 h + i % 4 of the 32-row block, query r32
#pragma unroll
                for (int bq = 0; bq < 4; ++bq) {
                    const rq_float4 nv = *(const rq_float4*)(nrow + t * 128 + bq * 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        rq_insert3(m1, m2, m3, rq_pos_score_finite((float)acc[4 * bq + e] * nv[e], (uint32_t)(t * 32 + bq * 8 + e)));
                }
            }
        }
        {
            float y1 = rq_scale_pos(m1, qsc), y2 = rq_scale_pos(m2, qsc), y3 = rq_scale_pos(m3, qsc);
            const uint32_t hb = (uint32_t)h << 2;
            float x1 = __uint_as_float(__float_as_uint(y1) | hb), x2 = __uint_as_float(__float_as_uint(y2) | hb), x3 = __uint_as_float(__float_as_uint(y3) | hb);
            const float o1 = __shfl_xor(x1, 32, 64), o2 = __shfl_xor(x2, 32, 64), o3 = __shfl_xor(x3, 32, 64);
            rq_insert3(x1, x2, x3, o1);
            rq_insert3(x1, x2, x3, o2);
            rq_insert3(x1, x2, x3, o3);
            asm("v_max_f32 %0, %1, %2" : "=v"(wmax) : "v"(wmax), "v"(x1));
            if (h == 0) stg[(32 * wave + r32) * SQ + (lq & (SQ - 1))] = rq_record_from_triple(x1, x2, x3);
        }
        if ((lq & (SQ - 1)) == SQ - 1 || lq == nloc - 1)
            flush(q_lo + (lq & ~(SQ - 1)), (lq & (SQ - 1)) + 1);
    }
    {
        const int ql = 32 * wave + r32;
        if (h == 0 && ql < nq_valid) wgmax_p[(int64_t)ql * wgmax_stride + b] = wmax;
    }
}

template <bool NT, int D, bool QS>
__global__ __launch_bounds__(512, 2) void rq_scanw32_kernel(RqScanArgs a) {
    rq_scanw32_body<NT, D, QS>(a, (int)blockIdx.x, (int)gridDim.x);
}

static constexpr size_t rq_scanw_lds_bytes(int QW, int QG) { return (size_t)4 * 24576 + 1024 + (size_t)16 * QW * QG * rqw_sq(QW, QG) * 8; }

template <bool NT, int D, int OCC, int QW, int QG, int EPI, int DBG, int PRIO, int I8>
__global__ __launch_bounds__(64 * QW, OCC) void rq_scanw_kernel(RqScanArgs a) {
    rq_scanw_body<NT, D, QW, QG, EPI, DBG, PRIO, I8>(a, (int)blockIdx.x, (int)gridDim.x);
}

template <bool NT, int D, int OCC, int QW, int QG, int EPI = 0, int DBG = 0, int PRIO = 0, int I8 = 0>
static hipError_t rq_scanw_launch_t(const RqScanArgs& a, int grid, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    constexpr size_t lds = rq_scanw_lds_bytes(QW, QG);
    static_assert(lds <= 160 * 1024, "LDS of one workgroup");
    static unsigned long long attr_done = 0;   // one bit per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!((attr_done >> (dev & 63)) & 1ull)) {
        e = hipFuncSetAttribute((const void*)rq_scanw_kernel<NT, D, OCC, QW, QG, EPI, DBG, PRIO, I8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done |= 1ull << (dev & 63);
    }
    if (e0 && e1) hipExtLaunchKernelGGL((rq_scanw_kernel<NT, D, OCC, QW, QG, EPI, DBG, PRIO, I8>), dim3(grid), dim3(64 * QW), (uint32_t)lds, stream, e0, e1, 0, a);
    else hipLaunchKernelGGL((rq_scanw_kernel<NT, D, OCC, QW, QG, EPI, DBG, PRIO, I8>), dim3(grid), dim3(64 * QW), lds, stream, a);
    return hipGetLastError();
}

// Variants (option "wide128" / "wide256"; queries per pass = 16 * waves * groups):
//   0  128 queries: 8 waves x 1 group, reads 12 fragments ahead, compiler-scheduled LDS waits          <- default for 128
//   1  128 queries: as 0, reads 4 ahead
//   8  128 queries: as 0 with asm fragment reads + counted lgkmcnt waits (within 1 % of 0)
//  11  256 queries: 8 waves x 2 groups, reads 2 ahead, asm fragment reads + counted lgkmcnt waits: 256 VGPRs, nothing
//      spilled (385-398 us per pass); A/B only since round 3 (see the int8 note below)
//   2  256 queries: as 11 with compiler-scheduled reads (7 values spilled outside the streaming loop; 400-415 us)   <- default for 256
//   4  128 queries: as 1 with rq_scan.hip's compare/select epilogue (A/B of the selection forms)
//   5  128 queries: 4 waves x 2 groups, one wave per SIMD (A/B: a lone wave cannot overlap its own VALU with its MFMAs)
//   6  128 queries: as 0 with the selection of waves 4..7 staggered by half a tile;  7: as 0 with s_setprio 1 for waves 4..7
//      (A/B of the two-waves-per-SIMD levers of MI355X_MICROARCH.md: neither beats 0 once the accumulators alternate)
//   90..95  timing experiments, results invalid: no selection / no LDS fragment reads / no MFMAs, on variant 4's (90-92)
//      and variant 0's (93-95) shape
template <bool NT, int D, bool QS = false>
static hipError_t rq_scanw32_launch_t(const RqScanArgs& a, int grid, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    constexpr size_t lds = rq_scanw_lds_bytes(8, 2);
    static unsigned long long attr_done = 0;   // one bit per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!((attr_done >> (dev & 63)) & 1ull)) {
        e = hipFuncSetAttribute((const void*)rq_scanw32_kernel<NT, D, QS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done |= 1ull << (dev & 63);
    }
    if (e0 && e1) hipExtLaunchKernelGGL((rq_scanw32_kernel<NT, D, QS>), dim3(grid), dim3(512), (uint32_t)lds, stream, e0, e1, 0, a);
    else hipLaunchKernelGGL((rq_scanw32_kernel<NT, D, QS>), dim3(grid), dim3(512), lds, stream, a);
    return hipGetLastError();
}

hipError_t rq_scan_wide_launch(const RqScanArgs& a, int variant, int queries, bool nt, int grid, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    if (grid <= 0) return hipErrorInvalidValue;
    // int8 image, 256 queries on the 32x32x32 form: 30 reads 8 fragments ahead, 31 reads 12 ahead, 32 reads 4 ahead
    if (queries == 256 && variant == 33) return nt ? rq_scanw32_launch_t<true, 12, true>(a, grid, stream, e0, e1) : rq_scanw32_launch_t<false, 12, true>(a, grid, stream, e0, e1);   // 31 with whole-quad DMA units
    if (queries == 256 && variant >= 30 && variant <= 32) {
        if (variant == 30) return nt ? rq_scanw32_launch_t<true, 8>(a, grid, stream, e0, e1) : rq_scanw32_launch_t<false, 8>(a, grid, stream, e0, e1);
        if (variant == 31) return nt ? rq_scanw32_launch_t<true, 12>(a, grid, stream, e0, e1) : rq_scanw32_launch_t<false, 12>(a, grid, stream, e0, e1);
        return nt ? rq_scanw32_launch_t<true, 4>(a, grid, stream, e0, e1) : rq_scanw32_launch_t<false, 4>(a, grid, stream, e0, e1);
    }
#define RQW_CASE(V, DD, OO, QQ, GG, ...) \
    if (variant == V && queries == 16 * QQ * GG) \
        return nt ? rq_scanw_launch_t<true, DD, OO, QQ, GG, ##__VA_ARGS__>(a, grid, stream, e0, e1) : rq_scanw_launch_t<false, DD, OO, QQ, GG, ##__VA_ARGS__>(a, grid, stream, e0, e1);
    RQW_CASE(0, 12, 2, 8, 1, 1) RQW_CASE(1, 4, 2, 8, 1, 1) RQW_CASE(2, 2, 2, 8, 2, 1)
    RQW_CASE(4, 4, 2, 8, 1, 0) RQW_CASE(5, 4, 1, 4, 2, 1)
    RQW_CASE(6, 12, 2, 8, 1, 1, 0, 2) RQW_CASE(7, 12, 2, 8, 1, 1, 0, 1)
    RQW_CASE(8, 12, 2, 8, 1, 1, 0, 3) RQW_CASE(11, 2, 2, 8, 2, 1, 0, 3)
    // int8 image (a.i8 = 4): 22  256 queries, 8 waves x 2 groups, reads 12 ahead, compiler-scheduled LDS waits   <- default ("wide256_8")
    //     25  as 22, reads 6 ahead.  (Round 3 also built this form with asm fragment reads + counted lgkmcnt waits, as variants 8 / 11 do
    //     for fp16: same speed -- 227 us per pass at 1M rows -- but NOT exact: one row in ~10^5 bins was scored from a stale register,
    //     differently from run to run.  An asm ds_read is invisible to the compiler's own hazard tracking: it may copy or re-use the
    //     destination register before the data has landed.  Removed; and the fp16 256-query default went back from 11 to 2 for the same reason.)
    RQW_CASE(22, 12, 2, 8, 2, 1, 0, 0, 1) RQW_CASE(25, 6, 2, 8, 2, 1, 0, 0, 1)
    // (a 128-query int8 form here -- 8 waves x 1 group, one workgroup per CU -- measured 154-160 us per pass against 150-157 of rq_scan.hip's
    //  I8 = 3 form, 4 waves x 2 groups at two per CU: not kept)
    RQW_CASE(90, 4, 2, 8, 1, 0, 1) RQW_CASE(91, 4, 2, 8, 1, 0, 2) RQW_CASE(92, 4, 2, 8, 1, 0, 3)
    RQW_CASE(93, 12, 2, 8, 1, 1, 1) RQW_CASE(94, 12, 2, 8, 1, 1, 2) RQW_CASE(95, 12, 2, 8, 1, 1, 3)
#undef RQW_CASE
    return hipErrorInvalidValue;
}
