// rq_exact.hip -- the LAST rung of the repair ladder: the exact scan of a whole shard (every row re-scored in fp64, no
// approximate pass, no certificate), as a dense fp64 contraction on the matrix cores.
//
// Until round 3 this rung was rq_rescore_kernel with one workgroup per (bin, query): the shard was read once PER QUERY
// (98 GB for 64 queries at 1M rows: 18.5 ms, 140x the normal step).  Here a workgroup keeps 32 queries in LDS (fp32, 97 KB)
// and reads a 16-row tile once for them: wave (g, r) multiplies query group g (16 queries) with the tiles of parity r; rows
// and queries are converted to fp64 on the way into v_mfma_f64_16x16x4_f64 -- fp16 x fp32 products are exact in fp64, the
// sums differ from numpy's only in their order (~1e-16 relative), and the ONE rounding to fp32 of the score definition
//  hides that except on a rounding boundary -- the same statement rq_rescore_kernel and the tail's re-score
// make.  Replaces, exhaustively and exactly, the collection.query of reference rag_uq/streaming_index.py:355-359.
//
// k mapping.  The dot product does not care in which order k is consumed as long as A and B agree: step (j, e) of a tile
// multiplies k = 32 j + 8 kq + e for kq = lane / 16 = 0..3, so that a lane's A operands of 8 consecutive steps are the 16
// contiguous bytes x[row][32 j + 8 kq .. + 7] (one dwordx4 load) and its B operands 8 contiguous floats of its query
// (two ds_read_b128; query rows are padded to 772 floats so that the 16 lanes of a quarter hit 64 different banks).
#include <hip/hip_runtime.h>

#include "rq_device.h"
#include "rq_kernels.h"

typedef double rq_double4 __attribute__((ext_vector_type(4)));

#define RQ_EXACT_QB 32            // queries per workgroup
#define RQ_EXACT_QPITCH 772       // floats per query row in LDS

// grid (G, ceil(B / 32)); 256 threads, one workgroup per CU (LDS).  a.binkeys must be null (exact mode: cand is [B][nb * 64] and
// row r of query q lands at cand[q * nb * 64 + r]).
__global__ __launch_bounds__(256, 1) void rq_exact_scan_kernel(RqRescoreArgs a, int B) {
    __shared__ __attribute__((aligned(16))) float qs[RQ_EXACT_QB * RQ_EXACT_QPITCH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wave & 1, par = wave >> 1;      // query group of this wave, parity of the tiles it multiplies
    const int c16 = lane & 15;          // row inside the tile (A operand) / query inside the group (B operand, result column)
    const int kq = lane >> 4;           // k quarter of the operands / row group of the result
    const int q0 = (int)blockIdx.y * RQ_EXACT_QB;
    // q32 is zero padded to 768 floats per query and to whole blocks of 64 query slots (rq_prep_body): every load is in range
    for (int i = threadIdx.x; i < RQ_EXACT_QB * (RQ_DPAD / 4); i += 256) {
        const int qi = i / (RQ_DPAD / 4), c = i % (RQ_DPAD / 4);
        *(float4*)&qs[qi * RQ_EXACT_QPITCH + 4 * c] = *(const float4*)(a.q32 + (size_t)(q0 + qi) * RQ_DPAD + 4 * c);
    }
    __syncthreads();
    const int q = q0 + 16 * g + c16;                                 // this lane's query (result column)
    const bool live_q = q < B;
    const double qn = live_q ? a.qnorm64[q] : 0.0;
    const float* qrow = qs + (16 * g + c16) * RQ_EXACT_QPITCH + 8 * kq;
    const int64_t ntiles = (a.n_rows + 15) / 16;
    const int64_t rows_alloc = (int64_t)a.nb * RQ_BIN_ROWS;         // cand entries per query
    const char* xb = (const char*)a.x;
    uint64_t* out = a.cand + (int64_t)(live_q ? q : 0) * rows_alloc;

    rq_half8 cur[24], nxt[24];
    auto load_tile = [&](int64_t t, rq_half8 (&dst)[24]) {
        // rows beyond the shard's end inside the last quad are zero padded storage (cap % 64 == 0): readable, keyed 0 below
        const char* r = xb + (t * 16 + c16) * (int64_t)(RQ_DPAD * 2) + 16 * kq;
#pragma unroll
        for (int j = 0; j < 24; ++j) dst[j] = *(const rq_half8*)(r + 64 * j);
    };
    const int64_t tstep = 2 * (int64_t)gridDim.x;
    int64_t t = 2 * (int64_t)blockIdx.x + par;
    if (t < ntiles) load_tile(t, cur);
    for (; t < ntiles; t += tstep) {
        const int64_t tn = t + tstep;
        if (tn < ntiles) load_tile(tn, nxt);                         // the next tile's 24 KiB travel while this one is multiplied
        rq_double4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            const float4 b0 = *(const float4*)(qrow + 32 * j), b1 = *(const float4*)(qrow + 32 * j + 4);
            const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(float)cur[j][e], (double)bq[e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // conversions stay with their 8 MFMAs: hoisting all 384 of them spilled 94 registers
        }
        // D of the fp64 16x16x4 form: register i of lane (c16, kq) is D[row = 4 i + kq][query = c16] (NOT 4 kq + i as for the fp32 /
        // int32 16x16 forms): for a fixed i the four lanes that share a query hold four consecutive rows -> 32 contiguous bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = t * 16 + 4 * i + kq;
            uint64_t key = 0;
            if (row < a.n_rows) {
                double s = acc[i];
                if (a.metric == 0) s = acc[i] / (qn * a.rownorm64[row] + 1e-30);
                key = rq_make_key(rq_sanitize((float)s), (uint32_t)row);
            }
            if (live_q) out[row] = key;
        }
        if (tn < ntiles) {
#pragma unroll
            for (int j = 0; j < 24; ++j) cur[j] = nxt[j];
        }
    }
    // the slots between the last tile and the end of the last bin (rows_alloc is a multiple of 64, tiles are 16 rows): empty keys
    if (blockIdx.x == 0 && par == 0 && live_q) {
        for (int64_t row = ntiles * 16 + kq; row < rows_alloc; row += 4) out[row] = 0;
    }
}

hipError_t rq_exact_scan_launch(const RqRescoreArgs& a, int B, int cu_count, hipStream_t stream) {
    if (a.binkeys != nullptr || a.nb <= 0 || B <= 0 || (int64_t)a.nb * RQ_BIN_ROWS < a.n_rows) return hipErrorInvalidValue;
    const int64_t npairs = ((a.n_rows + 15) / 16 + 1) / 2;
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(npairs, 1), std::max(cu_count, 1));
    hipLaunchKernelGGL(rq_exact_scan_kernel, dim3(gx, (B + RQ_EXACT_QB - 1) / RQ_EXACT_QB), dim3(256), 0, stream, a, B);
    return hipGetLastError();
}
