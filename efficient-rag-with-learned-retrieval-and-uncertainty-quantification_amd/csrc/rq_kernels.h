// rq_kernels.h -- host-visible launchers of the HIP kernels (internal; the public C ABI is include/rq.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct RqScanArgs {
    const void* x;            // fp16 corpus shard [rows_padded][768], rows_padded % 64 == 0, pad rows zero
    const float* row_scale;   // [rows_padded] 2^-12 / ||row|| (cosine) or 2^-12 (inner product); pad entries NaN
    const _Float16* qh;       // [QB][768] fp16(q / ||q|| * 2^12) of this block (QB = 64, 128 or 256), unused slots zero
    uint2* bins;              // [QB][bins_stride] per (query, quad) record, see rq_device.h
    int64_t bins_stride;      // records per query row, >= nquads
    int64_t n_rows;           // valid rows of the shard
    int nquads;               // ceil(n_rows / 64)
    int nq_valid;             // queries of this block that are real (<= QB)
    float* wgmax;             // [QB][wgmax_stride] largest approximate score per (query, scan workgroup)
    int wgmax_stride;         // >= grid
    // int8 scan (i8 = 1): x = the int8 image of the shard [rows_padded][768], row_scale = s_row / ||row|| (pad entries NaN),
    // qh = int8 queries [QB][768], qscale[QB] = s_query / ||query||; approximate score = int32 sum * row_scale * qscale
    // i8 = 2: the queries come as TWO int8 images (q = s (254 q_hi + q_lo), rq_prep_body): qh = q_hi, qlo = q_lo, approximate
    // score = (254 * sum_hi + sum_lo) * row_scale * qscale / 254 -- the query's quantisation error all but disappears from the bound
    int i8;
    const float* qscale;
    const void* qlo;
};
#define RQ_WGMAX_STRIDE 1024   // scan grids never exceed this many workgroups

// e0 / e1 (all scan launchers): optional events attached to the dispatch itself (hipExtLaunchKernel): the kernel's own start / end
// time stamps, no extra barrier packets on the stream
hipError_t rq_scan_launch(const RqScanArgs& a, int S, int pf, int ks, int qw, bool nt, int grid, int epi, hipStream_t stream,
                          hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// Passes of 128 / 256 queries (rq_scan_wide.hip): one 512-thread workgroup per CU, LDS reads running ahead of the MFMAs
// across stage boundaries, v_med3 selection.  Returns hipErrorInvalidValue for a variant that is not built or does not
// score `queries` queries per pass.
hipError_t rq_scan_wide_launch(const RqScanArgs& a, int variant, int queries, bool nt, int grid, hipStream_t stream,
                               hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// Row statistics at add time: row_norm64[i] = sqrt(sum x^2) in fp64; stats[3] (device, running maxima as double bits):
// largest norm, largest relative and absolute mass of fp16-subnormal elements in a row (see rq_select.hip).
hipError_t rq_rownorm_launch(const void* x, int64_t row_begin, int64_t row_end, double* norm64, unsigned long long* stats, hipStream_t stream);

// fp32 rows (device) -> fp16 rows, optionally L2-normalised first (see include/rq.h rq_index_add_f32).
hipError_t rq_convert_f32_launch(const float* src, int dim, int64_t n, int normalize, void* dst_rows_f16, hipStream_t stream);
hipError_t rq_pad_f16_launch(const void* src, int dim, int64_t n, void* dst_rows_f16, hipStream_t stream);

// Query preparation: qnorm64[q] = ||q|| (fp64); qh = fp16(q / ||q|| * 2^12) padded to 768, slots >= B zero.
// One 256-thread workgroup per query slot (rq_device.h rq_prep_body); run by rq_prep_queries_kernel or, for the NEXT batch
// of a throughput loop, by extra workgroups of the fused scan + tail launch (include/rq.h rq_search_hint_next_device).
struct RqPrepArgs {
    const float* q; int dim; int B;      // raw fp32 queries [B][dim]
    _Float16* qh; float* q32pad; double* qnorm64;
    int nslots;                          // workgroups = query slots written (padded batch size), 0 = none
    // int8 image for the int8 scan (all three NULL = not wanted): q8[slot][768] = round(q / s), s = max|q_i| / 127;
    // qscale8[slot] = s / ||q|| (1 for a zero / unused / non-finite query); qeps8[slot] = ||q - s q8|| / ||q||, the
    // query's share of the certificate's error bound (+inf for a non-finite query)
    signed char* q8; float* qscale8; float* qeps8;
    // second int8 image of the residual: r = q - s q8, q8lo = round(r / (s / 254)); qeps8s[slot] = ||q - s (q8 + q8lo / 254)|| / ||q||
    signed char* q8lo; float* qeps8s;
};
hipError_t rq_prep_queries_launch(const RqPrepArgs& a, hipStream_t stream);   // a.nslots workgroups

// int8 image of rows [row_begin, row_end) for the int8 scan: x8[row][768] = round(x / s_row), s_row = max|x_i| / 127;
// scale_cos[row] = s_row / ||row|| (0 for a zero row), scale_ip[row] = s_row; stat[0] (device, double bits, running
// maximum) = largest ||x - s x8|| / ||x|| of a row (+inf for a row with non-finite elements)
// binerr[row / 64] (device, fp32 bits, running maximum, zero before the first call) = the same figure for the worst row of every bin
hipError_t rq_quant_rows_launch(const void* x, const double* norm64, int64_t row_begin, int64_t row_end, signed char* x8,
                                float* scale_cos, float* scale_ip, unsigned long long* stat, float* binerr, hipStream_t stream);

// Pass 2: per query, the m best bins of bins[q][0..nbins) as sorted keys (score desc, bin asc); 0-padded.
hipError_t rq_select_bins_launch(const uint2* bins, int64_t bins_stride, int64_t nbins, int B, int m,
                                 uint64_t* binkeys, hipStream_t stream);

// Pass 3: exact fp64 re-score of every row of the first nb bins of each query -> candidate keys (score, local row).
struct RqRescoreArgs {
    const void* x;
    const float* q32;          // [B][768] raw fp32 queries, zero padded
    const double* qnorm64;     // [B]
    const double* rownorm64;   // [rows_padded]
    const uint64_t* binkeys;   // [B][binkeys_stride]
    int binkeys_stride;
    int nb;                    // bins re-scored per query
    int metric;                // 0 cosine, 1 inner product
    int64_t n_rows;
    uint64_t* cand;            // [B][nb*64]
};
hipError_t rq_rescore_launch(const RqRescoreArgs& a, int B, hipStream_t stream);
// Exact mode only (a.binkeys == nullptr, every bin of the shard): the same keys from a dense fp64 contraction on the matrix cores
// (v_mfma_f64_16x16x4_f64, rq_exact.hip): a 16-row tile is read once for 64 queries instead of once per query.
hipError_t rq_exact_scan_launch(const RqRescoreArgs& a, int B, int cu_count, hipStream_t stream);

// Pass 4: top-k of the candidates, certificate, outputs.
struct RqFinalArgs {
    const uint64_t* cand;      // [B][ncand]
    int ncand;
    const uint64_t* binkeys;   // [B][binkeys_stride]; entry nb is the best bin that was NOT re-scored
    int binkeys_stride;
    int nb;
    int64_t nbins;             // bins of the shard
    const double* qnorm64;
    int metric;
    float eps;                 // bound on |approximate - exact| in unit-query units
    float max_row_norm;        // for the inner-product bound
    int k;
    int64_t row_offset;        // global id of the shard's row 0
    int64_t n_rows;
    float* out_scores;         // [B][k]
    int64_t* out_rows;         // [B][k], -1 padded
    uint64_t* out_keys;        // [B][k] optional (global-row keys for cross-shard merging), may be null
    int* out_status;           // [B] 0 = certified exact, 1 = not certified
};
hipError_t rq_final_launch(const RqFinalArgs& a, int B, hipStream_t stream);

// ---- fast tail (the common case: m <= 128 bins wanted, k <= 128) ----
#define RQ_FAST_MAX_M 320
#define RQ_FAST_MAX_K 320       // beyond that the 512 partition maxima give too loose a threshold (k = 512: every query overflows)
#define RQ_CAND_CAP 4096       // candidate keys per query (compact list written by the tail kernel)

// Fused tail: threshold + bin collection + exact re-score + final top-k + certificate in one launch.
struct RqTailArgs {
    const float* q; int dim;                       // raw fp32 queries [B][dim]
    const void* x; const double* rownorm64; int64_t n_rows;
    const uint2* bins; int64_t bins_stride; int64_t nbins;
    const float* wgmax; int wgmax_stride; int nwg;
    int nwg_split, nwg2;                           // queries >= nwg_split were scanned by a grid of nwg2 workgroups (a call's narrow passes follow its wide ones)
    int m, metric, k;
    float eps, max_row_norm;
    int64_t row_offset;
    uint64_t* cand;                                // [B][RQ_CAND_CAP] compact candidate keys
    int* rowcount; int* done; int* ovf;            // [B] each, zero before the launch, reset by the kernel
    float* out_scores; int64_t* out_rows; uint64_t* out_keys; int* out_status;
    const float* qeps;                             // per query error share e_q (int8 scan): bound = e_q (1 + eps) + eps; null = eps alone
    const float* binerr;                           // int8 scan: worst row error of every bin (rq_quant_rows_kernel), or null.  A bin is tested against
    float eps_rows_max;                            //   T + (1 + e_q) (eps_rows_max - binerr[bin]): its own rows' bound instead of the shard's worst row
    int local_topk;                                // 1: a workgroup with more than k row jobs publishes only its own k best keys (rq_tail_body.h)
    float thr_mult;                                // threshold T = P - bound - (thr_mult - 1) * max(bound, thr_slack): 2.25 always
    float thr_slack;                               //   certifies; less = fewer candidates, the rare query is repaired (rq_tail_body.h)
    unsigned long long* dbg;                       // development: per-workgroup (start, end) wall-clock stamps of the fused launch, or null
    int stop_after;                                // development: 0 = full kernel, 1..4 = return after phase A..D
    int fused_nv;                                  // development: 0 = the launcher's rule, 1 / 4 / 8 = 512 / 2048 / 4096 bins per riding tail workgroup
};
// workgroups [0, scan_grid) run the scan `sa`, the next ones the tail `ta` of an EARLIER batch, the last pa.nslots the query
// preparation `pa` of a LATER batch
hipError_t rq_scan_tail_launch(const RqScanArgs& sa, const RqTailArgs& ta, int tail_B, const RqPrepArgs& pa, bool nt, int scan_grid, int epi, hipStream_t stream,
                               hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t rq_tail_launch(const RqTailArgs& a, int B, hipStream_t stream);
// chunk size rule shared by both launchers: 512-bin chunks while that keeps the grid around a thousand workgroups
// (enough to spread the hits, few enough to be one dispatch round), else 2048-bin chunks
static inline bool rq_tail_small_chunks(int64_t nbins, int B) { return ((nbins + 511) / 512) * B <= 1536; }

// Merge G sorted key lists per query (cross-shard): in [B][G*k] -> top-k scores/rows/keys.
hipError_t rq_merge_keys_launch(const uint64_t* keys, int n_per_query, int B, int k, float* out_scores, int64_t* out_rows,
                                uint64_t* out_keys, hipStream_t stream);
hipError_t rq_read_probe_launch(const void* x, int64_t bytes, bool nt, int grid, uint32_t* sink, hipStream_t stream);
