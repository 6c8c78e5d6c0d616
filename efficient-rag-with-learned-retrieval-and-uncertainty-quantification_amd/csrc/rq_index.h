// rq_index.h -- internal definitions shared by rq_api.hip (single-device index, the C ABI) and rq_multi.hip (the
// multi-device parent): error channel, the index object, the per-stream workspaces, the device guard.
// Not part of the public boundary (that is include/rq.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/rq.h"
#include "rq_device.h"
#include "rq_kernels.h"

#define RQ_INTERNAL __attribute__((visibility("hidden")))

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
RQ_INTERNAL int set_err(int code, const char* fmt, ...);   // stores the message of rq_last_error() (thread local), returns code
RQ_INTERNAL const char* rq_err_text();
#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_err(RQ_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// index object
// ---------------------------------------------------------------------------------------------
struct Workspace {
    int bcap = 0;                 // query slots (multiple of 64)
    int64_t bins_stride = 0;      // bin records per query
    int64_t binkeys_cap = 0;      // entries per query
    size_t cand_elems = 0;        // candidate keys allocated in total (queries of a call x keys per query)
    _Float16* qh = nullptr;
    float* q32 = nullptr;
    double* qn = nullptr;
    signed char* q8 = nullptr;    // int8 scan: the queries' int8 image, their scales and quantisation errors (rq_kernels.h RqPrepArgs)
    float* qscale8 = nullptr;
    float* qeps8 = nullptr;
    signed char* q8lo = nullptr;  // second int8 image (the residual) and the error left after both
    float* qeps8s = nullptr;
    uint2* bins = nullptr;        // [bcap][bins_stride] scan output: one record per (query, quad), see rq_device.h
    uint64_t* binkeys = nullptr;
    uint64_t* cand = nullptr;
    float* wgmax = nullptr;       // [bcap][RQ_WGMAX_STRIDE]
    int* rowcount = nullptr;      // [bcap] fast tail: candidate rows appended so far
    float* thr = nullptr;         // [bcap]
    int* done = nullptr;          // [bcap] fast tail: workgroups of the query that have finished
    int* ovf = nullptr;           // [bcap] fast tail: a workgroup found more bins / rows than it could hold
    bool counters_zero = false;   // rowcount/done/ovf known to be all zero (the tail kernel leaves them so)
    // staging for rq_search_fixup_device
    int fix_bcap = 0, fix_k = 0;
    float* fix_q = nullptr;
    float* fix_scores = nullptr;
    int64_t* fix_rows = nullptr;
    uint64_t* fix_keys = nullptr;
    int* fix_status = nullptr;
};

// Per caller stream: two workspaces (alternating calls), an internal tail stream and the events that
// order scan -> tail and tail -> reuse of the same workspace two calls later ("pipeline" option).
struct StreamCtx {
    Workspace w[2];
    hipStream_t tail = nullptr;
    hipEvent_t ev_scan[2] = {nullptr, nullptr};
    hipEvent_t ev_tail[2] = {nullptr, nullptr};
    bool tail_pending[2] = {false, false};
    uint64_t calls = 0;
    // "pipeline" = 2: the tail of the last search waits here and rides along with the next scan launch of the stream
    bool fused_pending = false;
    RqTailArgs fused_tail;
    int fused_B = 0;
    // "pipeline" = 2 keeps the prepared queries of three consecutive calls apart (slot = call number mod 3): call i scans
    // slot i, the tail of call i - 1 that rides with it reads slot i - 1, and the extra workgroups of the same launch prepare
    // the hinted queries of call i + 1 into slot i + 1 (rq_search_hint_next_device)
    _Float16* ring_qh[3] = {nullptr, nullptr, nullptr};
    float* ring_q32[3] = {nullptr, nullptr, nullptr};
    double* ring_qn[3] = {nullptr, nullptr, nullptr};
    signed char* ring_q8[3] = {nullptr, nullptr, nullptr};
    float* ring_qscale8[3] = {nullptr, nullptr, nullptr};
    float* ring_qeps8[3] = {nullptr, nullptr, nullptr};
    signed char* ring_q8lo[3] = {nullptr, nullptr, nullptr};
    float* ring_qeps8s[3] = {nullptr, nullptr, nullptr};
    const float* hint_q = nullptr;      // queries announced for the next fused call, not yet prepared
    int hint_B = 0;
    const float* prepped_q = nullptr;   // queries a launch has already prepared ...
    int prepped_B = 0, prepped_slot = -1;   // ... and the slot they are in
};

// Default scan variant: half-row stages (kstage 2), ring of 3, one LDS fragment ahead (prefetch 1, <= 168 VGPRs),
// 2 workgroups per CU.  All variants stream at the same rate; this one leaves room on every CU (registers:
// 2 x 168 + 168 <= 512 VGPRs; LDS: 3 x 53 760 B <= 160 KB) for a tail workgroup to be resident beside the scan.
struct rq_index {
    int dim = 0, device = 0, cu_count = 256;
    int64_t n = 0, cap = 0, row_offset = 0;
    char* x = nullptr;
    double* rownorm64 = nullptr;
    float* inv_norm = nullptr;
    float* ones = nullptr;
    int64_t ones_valid = 0;
    // int8 image of the shard for the int8 scan (option "scan8"), built lazily for rows [0, x8_valid)
    signed char* x8 = nullptr;
    float* scale8_cos = nullptr;   // [cap] s_row / ||row||, pad rows NaN
    float* scale8_ip = nullptr;    // [cap] s_row
    float* binerr8 = nullptr;      // [cap / 64] worst row's relative quantisation error of every bin (fp32, rounded up)
    unsigned long long* d_stat8 = nullptr;   // device: bits of the largest relative quantisation error of a row
    int64_t x8_valid = 0;
    double max_e8 = 0.0;           // host copy of that maximum over rows [0, x8_valid)
    double* d_maxnorm = nullptr;   // device: bits of the running maxima {row norm, relative, absolute fp16-subnormal mass of a row}
    double max_row_norm = 0.0, max_sub_rel = 0.0, max_sub_abs = 0.0;
    unsigned long long* dbg_stamps = nullptr;   // development (rq_debug_stamps)
    // rq_search_fixup_device saw too many repairs behind the int8 scan; kept apart for k <= 32 ([0]) and larger k ([1])
    // per class: 0 = one int8 image per query, 1 = two images, 2 = suspended (fp16 scan); escalated by rq_search_fixup_device
    int scan8_level[2] = {0, 1};
    int64_t scan8_checked[2] = {0, 0}, scan8_repaired[2] = {0, 0};
    int64_t scan8_used = 0;        // searches that scanned the int8 image
    bool calibrating = false;      // scan8_calibrate is running its sample searches
    int64_t calib_rows = 0;        // rows of the shard when the ladder's start levels were last measured (0 = not yet)
    float calib_ms[2][3] = {{0, 0, 0}, {0, 0, 0}};   // per class, per rung (one image / two images / fp16): ms of the 64-query sample search
    int calib_unc[2][3] = {{0, 0, 0}, {0, 0, 0}};    // ... and its uncertified queries
    bool wg_auto = true;           // scan workgroups per CU by the library's rule (option "wg_per_cu" pins it)
    bool last_use8 = false;        // the caller's last search scanned the int8 image (what rq_search_fixup_device's repairs are counted against)
    // Calls of more than 64 queries in a class whose own rung is "two images" (k > 32 by default): the 256- / 128-query passes exist for ONE image only,
    // and one image there beats the fp16 wide passes (1M rows, 500 queries, k = 100: 740 against 944 us).  Allowed when the image-build measurement found the
    // one-image rung eligible for the class (or the image is forced, scan8 = 2); given up -- for wide calls only -- when more than 1 in 16 of their queries needed repair.
    bool wide1_ok[2] = {false, false}, wide1_off[2] = {false, false}, last_wide1 = false;
    int64_t wide1_checked[2] = {0, 0}, wide1_repaired[2] = {0, 0};
    int64_t repaired_total = 0;    // queries that came back uncertified and were repaired (any rung of the repair ladder)
    int64_t hints_used = 0;        // rq_search_hint_next_device: searches that skipped their preparation launch
    uint64_t scan_seq = 0;         // scan launches seen while profile = 1 (every profile_stride-th one is timed)
    // options
    int ring = 3, prefetch = 1, kstage = 2, wide_batch = 1, wg_per_cu = 2, nt = -1, slack_bins = -1, profile = 0, profile_stride = 1, scan_nostore = 0, fast_tail = 1, pipeline = 0, tail_stop = 0, poison_cand = 0, wide128 = 0, wide256 = 2, epi = 1, use_hint = 1, profile_legacy = 0, scan8 = 1, tail_local = 1, scan8_split = -1, wide8 = 1, wide256_8 = 31, bin_bound = 1, exact_mfma = 1, fused_nv = 0;
    double thr_mult8 = 1.25;       // int8 scan: threshold = P - thr_mult8 * bound (rq_tail_body.h)
    double eps = -1.0;
    std::map<hipStream_t, StreamCtx> ctx;
    hipStream_t own_stream = nullptr;
    // host-call staging
    float* h_dq = nullptr; float* h_dscores = nullptr; int64_t* h_drows = nullptr; int* h_dstatus = nullptr;
    int h_bcap = 0, h_kcap = 0;
    // small blocking searches (the reference's one-query-per-call pattern): results leave in ONE copy into pinned memory
    char* hs_dev = nullptr; char* hs_pin = nullptr; float* hs_pin_q = nullptr; size_t hs_bytes = 0, hs_qfloats = 0;
    void* add_stage = nullptr; size_t add_stage_bytes = 0;   // device staging of host-row appends
    bool hs_small = false;         // which staging path the search in flight uses (search_begin / search_end)
    // Multi-device parent (rq_index_create with n_devices > 1): no device memory of its own, one single-device child per
    // entry of device_ids.  Global rows are dealt to the children in stripes of `stripe` rows, round robin (rq_multi.hip):
    // global row g -> stripe s = g / stripe, child s % G, local row (s / G) * stripe + g % stripe.
    std::vector<rq_index*> shards;
    int64_t stripe = 65536;
    bool poisoned = false;         // an append failed part-way inside a child: the global <-> local mapping no longer holds
    std::vector<size_t> m_live;    // search staging of the parent, kept between calls: children that hold rows,
    std::vector<float> m_scores;   // their k best scores [live child][B][k]
    std::vector<int64_t> m_rows;   // and local rows
    int m_B = 0;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t ev_used = 0;
    int64_t ev_bytes = 0;          // algorithmic corpus bytes of the launches those events time (fp16 rows or their int8 image)
    rq_timing t = {};
};

// Derived bound on |approximate scan score - exact score| for unit queries and cosine scaling:
//   fp16 rounding of the unit query (2^-11 relative, 2^-25 absolute in the subnormal range),
//   fp32 accumulation inside the MFMA chain (<= 4 * 768 * 2^-24 of sum|q_i x_i| <= 1, conservative),
//   two fp32 roundings for the row scale,
//   2^-17 relative for the row position that rq_scan_wide.hip writes into the 6 low mantissa bits of a score
//   (4.88e-4 + 8e-7 + 1.83e-4 + 1.2e-7 + 7.6e-6 = 6.8e-4).  See DESIGN.md "certificate".
static const float RQ_EPS_DEFAULT = 7.0e-4f;

// Bound on |scan score - exact score| handed to the tail kernels, which use it as is for cosine and multiplied by the
// largest row norm for the inner product: the derived bound (or option "eps") plus what the matrix cores drop by flushing
// the fp16-subnormal elements of a stored row (rq_select.hip rq_rownorm_kernel).
RQ_INTERNAL float scan_eps(const rq_index* idx, int metric);
// Beyond this bound the approximate pass cannot narrow anything down (cosine scores live in [-1, 1]): scan exactly.
static const float RQ_EPS_USELESS = 0.05f;

static const int RQ_NB_MAX = 3071;

// Every entry point works on the index's device and puts the caller's current device back on return (a caller that
// holds tensors on another GPU, e.g. torch with several devices, must not find its device switched under it).
struct DeviceGuard {
    int prev = -1, dev = -1;
    bool ok = true;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define RQ_ON_DEVICE(idx)                                                                                  \
    DeviceGuard dg_((idx)->device);                                                                        \
    if (!dg_.ok) return set_err(RQ_EHIP, "cannot select device %d", (idx)->device)


// ---- entry points shared between the two translation units ------------------------------------------------
RQ_INTERNAL int rq_add_host_common(rq_index* idx, const void* rows, int64_t n_rows, bool is_f32, int normalize);
RQ_INTERNAL int rq_search_begin(rq_index* idx, const float* queries, int B, int k, int metric);      // stage + enqueue, no wait
RQ_INTERNAL int rq_search_end(rq_index* idx, int B, int k, int metric, float* out_scores, int64_t* out_rows);   // wait, repair, hand over
RQ_INTERNAL rq_index* rq_multi_create(int dim, int n_devices, const int* device_ids);
RQ_INTERNAL int rq_multi_reserve(rq_index* idx, int64_t n_rows);
RQ_INTERNAL int rq_multi_add(rq_index* idx, const void* rows, int64_t n_rows, bool is_f32, int normalize);
RQ_INTERNAL int rq_multi_get_rows(const rq_index* idx, int64_t row_begin, int64_t n_rows, uint16_t* out);
RQ_INTERNAL int rq_multi_search(rq_index* idx, const float* queries, int B, int k, int metric, float* out_scores, int64_t* out_rows);
