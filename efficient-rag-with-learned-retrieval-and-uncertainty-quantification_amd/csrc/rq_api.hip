// rq_api.hip -- the C ABI of include/rq.h: shard storage in HBM, search orchestration, persistence.
//
// One rq_index = one row shard resident on one MI355X.  Layout in HBM:
//   x          [cap][768] fp16, cap % 64 == 0, rows >= n are zero      (the only large array)
//   rownorm64  [cap]      fp64 L2 norm of the stored row               (exact re-score)
//   inv_norm   [cap]      fp32 2^-12 / norm, 0 for zero rows, NaN for pad rows (scan, cosine; the queries carry 2^12)
//   ones       [cap]      fp32 2^-12 for rows < n, NaN beyond                  (scan, inner product; lazy)
// plus one workspace per stream (query fragments, per-bin scan records, bin keys, candidate keys).
// Internal definitions: rq_index.h; the multi-device parent (n_devices > 1): rq_multi.hip.
#include "rq_index.h"

hipError_t rq_rowscale_launch(const double* norm64, int64_t row_begin, int64_t row_end, float* inv_norm, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* rq_err_text() { return g_err; }

float scan_eps(const rq_index* idx, int metric) {
    const double base = idx->eps < 0 ? (double)RQ_EPS_DEFAULT : idx->eps;
    if (metric == RQ_METRIC_COSINE) return (float)(base + idx->max_sub_rel * (1.0 + 1e-6));
    return (float)(base + (idx->max_row_norm > 0.0 ? idx->max_sub_abs / idx->max_row_norm * (1.0 + 1e-6) : 0.0));
}

static int nb_default(const rq_index* idx, int k) {
    const int slack = idx->slack_bins >= 0 ? idx->slack_bins : std::max(8, k / 8);
    return k + slack;
}

extern "C" int rq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" const char* rq_last_error(void) { return rq_err_text(); }
extern "C" const char* rq_version(void) { return "rq-hip 0.1 (gfx950)"; }

static int flush_all(rq_index* idx);   // launches every tail still waiting for a scan ("pipeline" = 2)
static const size_t RQ_MAX_STREAM_CTX = 8;

static void drop_x8(rq_index* idx) {
    void* p[] = {idx->x8, idx->scale8_cos, idx->scale8_ip, idx->binerr8};
    for (void* q : p) if (q) (void)hipFree(q);
    idx->x8 = nullptr; idx->scale8_cos = nullptr; idx->scale8_ip = nullptr; idx->binerr8 = nullptr;
    idx->x8_valid = 0; idx->max_e8 = 0.0;
}

static int grow(rq_index* idx, int64_t want_rows) {
    if (want_rows <= idx->cap) return RQ_OK;
    int64_t cap = std::max<int64_t>(idx->cap * 2, 4096);
    while (cap < want_rows) cap *= 2;
    if (want_rows > (int64_t)1 << 27 && cap > want_rows + want_rows / 8) cap = want_rows + want_rows / 8;   // big shards: 12% headroom
    cap = (cap + 63) / 64 * 64;
    if (int r = flush_all(idx)) return r;
    HIPCHK(hipDeviceSynchronize());   // searches in flight on any stream still read the old buffers
    char* nx = nullptr; double* nn = nullptr; float* ni = nullptr;
    hipError_t e = hipMalloc((void**)&nx, (size_t)cap * RQ_DPAD * 2);
    if (e != hipSuccess) return set_err(RQ_ENOMEM, "hipMalloc of %lld corpus rows failed: %s", (long long)cap, hipGetErrorString(e));
    if ((e = hipMalloc((void**)&nn, (size_t)cap * sizeof(double))) != hipSuccess || (e = hipMalloc((void**)&ni, (size_t)cap * sizeof(float))) != hipSuccess) {
        (void)hipFree(nx);
        if (nn) (void)hipFree(nn);
        return set_err(RQ_ENOMEM, "hipMalloc of the row statistics of %lld rows failed: %s", (long long)cap, hipGetErrorString(e));
    }
    const int64_t keep = idx->n;
    if (keep > 0) {
        HIPCHK(hipMemcpyAsync(nx, idx->x, (size_t)keep * RQ_DPAD * 2, hipMemcpyDeviceToDevice, idx->own_stream));
        HIPCHK(hipMemcpyAsync(nn, idx->rownorm64, (size_t)keep * sizeof(double), hipMemcpyDeviceToDevice, idx->own_stream));
        HIPCHK(hipMemcpyAsync(ni, idx->inv_norm, (size_t)keep * sizeof(float), hipMemcpyDeviceToDevice, idx->own_stream));
    }
    HIPCHK(hipMemsetAsync(nx + (size_t)keep * RQ_DPAD * 2, 0, (size_t)(cap - keep) * RQ_DPAD * 2, idx->own_stream));
    HIPCHK(hipMemsetAsync(nn + keep, 0, (size_t)(cap - keep) * sizeof(double), idx->own_stream));
    // row scales of the pad rows are NaN (0xffffffff): their scan scores sort last without a per-score row test
    // (rq_scan_wide.hip); rq_scan.hip masks rows >= n on its own
    HIPCHK(hipMemsetAsync(ni + keep, 0xff, (size_t)(cap - keep) * sizeof(float), idx->own_stream));
    HIPCHK(hipStreamSynchronize(idx->own_stream));
    if (idx->x) (void)hipFree(idx->x);
    if (idx->rownorm64) (void)hipFree(idx->rownorm64);
    if (idx->inv_norm) (void)hipFree(idx->inv_norm);
    if (idx->ones) { (void)hipFree(idx->ones); idx->ones = nullptr; idx->ones_valid = 0; }
    drop_x8(idx);   // the int8 image is rebuilt for the new capacity by the next search that wants it
    idx->x = nx; idx->rownorm64 = nn; idx->inv_norm = ni; idx->cap = cap;
    return RQ_OK;
}

extern "C" rq_index* rq_index_create(int dim, int n_devices, const int* device_ids) {
    if (dim < 1 || dim > RQ_MAX_DIM) { set_err(RQ_EINVAL, "dim %d outside 1..%d", dim, RQ_MAX_DIM); return nullptr; }
    if (n_devices < 1 || n_devices > 64 || !device_ids) { set_err(RQ_EINVAL, "n_devices %d outside 1..64 or no device list", n_devices); return nullptr; }
    if (n_devices > 1) return rq_multi_create(dim, n_devices, device_ids);
    const int ndev = rq_device_count();
    if (ndev <= 0) { set_err(RQ_ENODEVICE, "no HIP device visible: the gfx950 backend has no CPU fallback"); return nullptr; }
    if (device_ids[0] < 0 || device_ids[0] >= ndev) { set_err(RQ_EINVAL, "device %d outside 0..%d", device_ids[0], ndev - 1); return nullptr; }
    rq_index* idx = new rq_index();
    idx->dim = dim;
    idx->device = device_ids[0];
    DeviceGuard dg_(idx->device);
    hipDeviceProp_t prop;
    if (!dg_.ok || hipGetDeviceProperties(&prop, idx->device) != hipSuccess) {
        set_err(RQ_EHIP, "cannot open device %d", idx->device);
        delete idx;
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(RQ_ENODEVICE, "device %d is %s; this library holds gfx950 code only", idx->device, prop.gcnArchName);
        delete idx;
        return nullptr;
    }
    idx->cu_count = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&idx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&idx->d_maxnorm, 3 * sizeof(double)) != hipSuccess ||
        hipMemset(idx->d_maxnorm, 0, 3 * sizeof(double)) != hipSuccess) {
        set_err(RQ_EHIP, "device setup failed on device %d", idx->device);
        delete idx;
        return nullptr;
    }
    return idx;
}

static void free_ws(Workspace& w) {
    void* p[] = {w.qh, w.q32, w.qn, w.q8, w.qscale8, w.qeps8, w.q8lo, w.qeps8s, w.bins, w.binkeys, w.cand, w.wgmax, w.rowcount, w.thr, w.done, w.ovf, w.fix_q, w.fix_scores, w.fix_rows, w.fix_keys, w.fix_status};
    for (void* q : p) if (q) (void)hipFree(q);
    w = Workspace();
}

static void free_ctx(StreamCtx& c) {
    free_ws(c.w[0]);
    free_ws(c.w[1]);
    for (int p = 0; p < 3; ++p) {
        if (c.ring_qh[p]) (void)hipFree(c.ring_qh[p]);
        if (c.ring_q32[p]) (void)hipFree(c.ring_q32[p]);
        if (c.ring_qn[p]) (void)hipFree(c.ring_qn[p]);
        if (c.ring_q8[p]) (void)hipFree(c.ring_q8[p]);
        if (c.ring_qscale8[p]) (void)hipFree(c.ring_qscale8[p]);
        if (c.ring_qeps8[p]) (void)hipFree(c.ring_qeps8[p]);
        if (c.ring_q8lo[p]) (void)hipFree(c.ring_q8lo[p]);
        if (c.ring_qeps8s[p]) (void)hipFree(c.ring_qeps8s[p]);
    }
    if (c.tail) (void)hipStreamDestroy(c.tail);
    for (int p = 0; p < 2; ++p) {
        if (c.ev_scan[p]) (void)hipEventDestroy(c.ev_scan[p]);
        if (c.ev_tail[p]) (void)hipEventDestroy(c.ev_tail[p]);
    }
    c = StreamCtx();
}

extern "C" void rq_index_destroy(rq_index* idx) {
    if (!idx) return;
    if (!idx->shards.empty()) {
        for (rq_index* c : idx->shards) rq_index_destroy(c);
        delete idx;
        return;
    }
    DeviceGuard dg_(idx->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : idx->ctx) free_ctx(kv.second);
    for (auto& ev : idx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (idx->hs_pin) (void)hipHostFree(idx->hs_pin);
    if (idx->hs_pin_q) (void)hipHostFree(idx->hs_pin_q);
    void* p[] = {idx->add_stage, idx->x, idx->rownorm64, idx->inv_norm, idx->ones, idx->x8, idx->scale8_cos, idx->scale8_ip, idx->binerr8, idx->d_stat8, idx->d_maxnorm, idx->h_dq, idx->h_dscores, idx->h_drows, idx->h_dstatus, idx->hs_dev, idx->dbg_stamps};
    for (void* q : p) if (q) (void)hipFree(q);
    if (idx->own_stream) (void)hipStreamDestroy(idx->own_stream);
    delete idx;
}

extern "C" int rq_index_dim(const rq_index* idx) { return idx ? idx->dim : RQ_EINVAL; }
extern "C" int64_t rq_index_size(const rq_index* idx) { return idx ? idx->n : RQ_EINVAL; }
extern "C" int rq_index_set_row_offset(rq_index* idx, int64_t off) {
    if (!idx || off < 0) return set_err(RQ_EINVAL, "bad row offset");
    // keys carry the global row as a 32-bit field (0xffffffff is reserved): the whole shard must stay below it
    if ((uint64_t)off + (uint64_t)idx->n >= 0xffffffffull) return set_err(RQ_EUNSUPPORTED, "row_offset %lld + %lld rows reaches 2^32-1: row ids are 32-bit inside keys", (long long)off, (long long)idx->n);
    idx->row_offset = off;
    return RQ_OK;
}
extern "C" int rq_index_reserve(rq_index* idx, int64_t n_rows) {
    if (!idx || n_rows < 0) return set_err(RQ_EINVAL, "bad reserve");
    if (!idx->shards.empty()) return rq_multi_reserve(idx, n_rows);
    RQ_ON_DEVICE(idx);
    return grow(idx, n_rows);
}

// ---- append ----------------------------------------------------------------------------------
static int finish_add(rq_index* idx, int64_t n_new) {
    hipStream_t s = idx->own_stream;
    const int64_t b = idx->n, e = idx->n + n_new;
    HIPCHK(rq_rownorm_launch(idx->x, b, e, idx->rownorm64, (unsigned long long*)idx->d_maxnorm, s));
    HIPCHK(rq_rowscale_launch(idx->rownorm64, b, e, idx->inv_norm, s));
    double st[3] = {0.0, 0.0, 0.0};
    HIPCHK(hipMemcpyAsync(st, idx->d_maxnorm, sizeof(st), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    idx->max_row_norm = st[0]; idx->max_sub_rel = st[1]; idx->max_sub_abs = st[2];
    idx->n = e;
    return RQ_OK;
}

static int add_device_common(rq_index* idx, const void* d_rows, int64_t n_rows, bool is_f32, int normalize) {
    if (!idx || (!d_rows && n_rows > 0) || n_rows < 0) return set_err(RQ_EINVAL, "bad add arguments");
    if (!idx->shards.empty()) return set_err(RQ_EUNSUPPORTED, "device-pointer appends on a multi-device index: use rq_index_add_f16 / _f32 (host rows)");
    if (n_rows == 0) return RQ_OK;
    if ((uint64_t)idx->row_offset + (uint64_t)idx->n + (uint64_t)n_rows >= 0xffffffffull)   // checked BEFORE anything is appended
        return set_err(RQ_EUNSUPPORTED, "row ids beyond 2^32-1 are not supported (row_offset %lld + %lld rows)", (long long)idx->row_offset, (long long)(idx->n + n_rows));
    RQ_ON_DEVICE(idx);
    // The rows may have been produced on any stream of the caller (e.g. torch's): wait for all of it.
    // Appending is not a hot path; searches in flight on other streams are drained too, which also
    // makes it safe to reallocate the shard below.
    if (int r = flush_all(idx)) return r;
    HIPCHK(hipDeviceSynchronize());
    if (int r = grow(idx, idx->n + n_rows)) return r;
    char* dst = idx->x + (size_t)idx->n * RQ_DPAD * 2;
    if (is_f32) HIPCHK(rq_convert_f32_launch((const float*)d_rows, idx->dim, n_rows, normalize, dst, idx->own_stream));
    else if (idx->dim == RQ_DPAD) HIPCHK(hipMemcpyAsync(dst, d_rows, (size_t)n_rows * RQ_DPAD * 2, hipMemcpyDeviceToDevice, idx->own_stream));
    else HIPCHK(rq_pad_f16_launch(d_rows, idx->dim, n_rows, dst, idx->own_stream));
    return finish_add(idx, n_rows);
}
extern "C" int rq_index_add_f16_device(rq_index* idx, const void* d_rows, int64_t n_rows) { return add_device_common(idx, d_rows, n_rows, false, 0); }
extern "C" int rq_index_add_f32_device(rq_index* idx, const float* d_rows, int64_t n_rows, int normalize) {
    return add_device_common(idx, d_rows, n_rows, true, normalize);
}

int rq_add_host_common(rq_index* idx, const void* rows, int64_t n_rows, bool is_f32, int normalize) {
    if (!idx || (!rows && n_rows > 0) || n_rows < 0) return set_err(RQ_EINVAL, "bad add arguments");
    if (n_rows == 0) return RQ_OK;
    if (!idx->shards.empty()) return rq_multi_add(idx, rows, n_rows, is_f32, normalize);
    RQ_ON_DEVICE(idx);
    if (int r = grow(idx, idx->n + n_rows)) return r;
    // stream the host rows through a bounded device staging buffer, kept between calls (an indexing run appends
    // 100 documents at a time: no hipMalloc / hipFree per batch); a buffer beyond 16 MiB is given back after the call
    const size_t esz = is_f32 ? 4 : 2;
    const int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / (int64_t)(idx->dim * esz));
    const size_t need = (size_t)std::min(chunk, n_rows) * idx->dim * esz;
    if (need > idx->add_stage_bytes) {
        if (idx->add_stage) (void)hipFree(idx->add_stage);
        idx->add_stage = nullptr; idx->add_stage_bytes = 0;
        HIPCHK(hipMalloc(&idx->add_stage, need));
        idx->add_stage_bytes = need;
    }
    int rc = RQ_OK;
    for (int64_t off = 0; off < n_rows && rc == RQ_OK; off += chunk) {
        const int64_t m = std::min(chunk, n_rows - off);
        hipError_t e = hipMemcpy(idx->add_stage, (const char*)rows + (size_t)off * idx->dim * esz, (size_t)m * idx->dim * esz, hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = set_err(RQ_EHIP, "H2D copy failed: %s", hipGetErrorString(e)); break; }
        rc = add_device_common(idx, idx->add_stage, m, is_f32, normalize);
    }
    if (idx->add_stage_bytes > ((size_t)16 << 20)) { (void)hipFree(idx->add_stage); idx->add_stage = nullptr; idx->add_stage_bytes = 0; }
    return rc;
}
extern "C" int rq_index_add_f16(rq_index* idx, const uint16_t* rows, int64_t n_rows) { return rq_add_host_common(idx, rows, n_rows, false, 0); }
extern "C" int rq_index_add_f32(rq_index* idx, const float* rows, int64_t n_rows, int normalize) {
    return rq_add_host_common(idx, rows, n_rows, true, normalize);
}

extern "C" int rq_index_get_rows_f16(const rq_index* idx, int64_t row_begin, int64_t n_rows, uint16_t* out) {
    if (!idx || !out || row_begin < 0 || n_rows < 0 || row_begin + n_rows > idx->n) return set_err(RQ_EINVAL, "row range outside the index");
    if (n_rows == 0) return RQ_OK;
    if (!idx->shards.empty()) return rq_multi_get_rows(idx, row_begin, n_rows, out);
    RQ_ON_DEVICE(idx);
    HIPCHK(hipMemcpy2D(out, (size_t)idx->dim * 2, idx->x + (size_t)row_begin * RQ_DPAD * 2, (size_t)RQ_DPAD * 2, (size_t)idx->dim * 2,
                       (size_t)n_rows, hipMemcpyDeviceToHost));
    return RQ_OK;
}

// ---- options ---------------------------------------------------------------------------------
// int8 scan, per class of k (<= 32 / larger): where the adaptive ladder one image -> two images -> fp16 scan starts
static void scan8_reset_levels(rq_index* idx) {
    for (int c = 0; c < 2; ++c) {
        idx->scan8_level[c] = idx->scan8_split < 0 ? c : (idx->scan8_split ? 1 : 0);
        idx->scan8_checked[c] = idx->scan8_repaired[c] = 0;
        idx->wide1_ok[c] = idx->wide1_off[c] = false;
        idx->wide1_checked[c] = idx->wide1_repaired[c] = 0;
    }
    idx->calib_rows = 0;   // "scan8" = 1: the next search that brings the image up to date calibrates again
}

extern "C" int rq_set_option(rq_index* idx, const char* name, double v) {
    if (!idx || !name) return set_err(RQ_EINVAL, "bad option call");
    if (!idx->shards.empty()) {
        if (std::string(name) == "stripe_rows") {   // multi-device parent: rows per stripe (rq_multi.hip), before the first append only
            if (idx->n != 0) return set_err(RQ_EINVAL, "stripe_rows can only be set on an empty multi-device index");
            if (v < 64 || v > (double)(1 << 30) || (int64_t)v % 64 != 0) return set_err(RQ_EINVAL, "stripe_rows must be a multiple of 64 in 64..2^30");
            idx->stripe = (int64_t)v;
            return RQ_OK;
        }
        for (rq_index* c : idx->shards)
            if (int r = rq_set_option(c, name, v)) return r;
        return RQ_OK;
    }
    const std::string s(name);
    if (s == "ring") { if (v < 2 || v > 6) return set_err(RQ_EINVAL, "ring must be 2..6"); idx->ring = (int)v; }
    else if (s == "wide_batch") { if (v < 0 || v > 3) return set_err(RQ_EINVAL, "wide_batch must be 0..3"); idx->wide_batch = (int)v; }
    else if (s == "wide128") { if (v < 0 || v > 99) return set_err(RQ_EINVAL, "wide128: a 128-query variant of csrc/rq_scan_wide.hip"); idx->wide128 = (int)v; }
    else if (s == "wide256") { if (v < 0 || v > 92) return set_err(RQ_EINVAL, "wide256: a 256-query variant of csrc/rq_scan_wide.hip"); idx->wide256 = (int)v; }
    else if (s == "kstage") { if (v != 1 && v != 2) return set_err(RQ_EINVAL, "kstage must be 1 or 2"); idx->kstage = (int)v; }
    else if (s == "prefetch") { if (v != 1 && v != 4 && v != 6 && v != 12) return set_err(RQ_EINVAL, "prefetch must be 1, 4, 6 or 12"); idx->prefetch = (int)v; }
    else if (s == "wg_per_cu") { if (v < 0 || v > 8) return set_err(RQ_EINVAL, "wg_per_cu must be 1..8 (0: back to the library's rule)"); idx->wg_auto = v == 0; idx->wg_per_cu = v == 0 ? 2 : (int)v; }
    else if (s == "nt") idx->nt = (int)v;
    else if (s == "cu_count") { if (v < 1 || v > 1024) return set_err(RQ_EINVAL, "cu_count must be 1..1024"); idx->cu_count = (int)v; }   // test hook: shrinks the scan grid
    else if (s == "slack_bins") idx->slack_bins = (int)v;
    else if (s == "eps") idx->eps = v;
    else if (s == "profile") idx->profile = (int)v;
    else if (s == "scan_nostore") idx->scan_nostore = (int)v;   // timing experiments only: 1 = the scan writes nothing (results invalid)
    else if (s == "profile_stride") { if (v < 1) return set_err(RQ_EINVAL, "profile_stride must be >= 1"); idx->profile_stride = (int)v; }
    else if (s == "fast_tail") idx->fast_tail = (int)v;
    else if (s == "pipeline") { if (v != 0 && v != 1 && v != 2) return set_err(RQ_EINVAL, "pipeline must be 0, 1 or 2"); if (int r = flush_all(idx)) return r; idx->pipeline = (int)v; }
    else if (s == "tail_stop") idx->tail_stop = (int)v;
    else if (s == "epi" || s == "fused_epi") idx->epi = (int)v != 0;   // selection form of the 64-query scan (default variant and fused launch): 1 = positions inside the scores, 0 = compare / select
    else if (s == "profile_legacy") idx->profile_legacy = (int)v != 0;   // time scans with hipEventRecord around the launch (round 1) instead of dispatch-attached events
    else if (s == "scan8") { if (v < 0 || v > 2) return set_err(RQ_EINVAL, "scan8 must be 0, 1 or 2"); idx->scan8 = (int)v; scan8_reset_levels(idx); }   // see run_pipeline
    else if (s == "wide256_8") { if (v != 0 && v != 22 && v != 25 && !(v >= 30 && v <= 33)) return set_err(RQ_EINVAL, "wide256_8: 0 (off) or a 256-query int8 variant of csrc/rq_scan_wide.hip (22, 25, 30..33)"); idx->wide256_8 = (int)v; }
    else if (s == "wide8") idx->wide8 = (int)v != 0;   // calls of more than 64 queries may use 128-query passes over the int8 image
    else if (s == "scan8_split") { if (v < -1 || v > 1) return set_err(RQ_EINVAL, "scan8_split must be -1, 0 or 1"); idx->scan8_split = (int)v; scan8_reset_levels(idx); }   // see run_pipeline
    else if (s == "thr_mult8") { if (!(v >= 1.05 && v <= 2.25)) return set_err(RQ_EINVAL, "thr_mult8 %g outside 1.05..2.25", v); idx->thr_mult8 = v; }
    else if (s == "exact_mfma") idx->exact_mfma = (int)v != 0;   // A/B: 0 = the exact scan of a whole shard re-scores bin by bin and query by query (rq_rescore_kernel, rounds 1-2)
    else if (s == "fused_nv") idx->fused_nv = (int)v;   // development: bins per riding tail workgroup (0 = the launcher's rule, 1 / 4 / 8 x 512)
    else if (s == "bin_bound") idx->bin_bound = (int)v != 0;     // A/B: 0 = every bin is tested with the shard's worst row error (round 2)
    else if (s == "tail_local") idx->tail_local = (int)v != 0;   // A/B: 0 = every re-scored row's key goes to the query's global list
    else if (s == "use_hint") idx->use_hint = (int)v != 0;   // 0: rq_search_hint_next_device is ignored (A/B of the folded query preparation)
    else if (s == "poison_cand") idx->poison_cand = (int)v;   // test hook: candidate lists are filled with 0xff..ff keys before every tail
    else return set_err(RQ_EINVAL, "unknown option '%s'", name);
    return RQ_OK;
}
extern "C" double rq_get_option(const rq_index* idx, const char* name) {
    if (!idx || !name) return NAN;
    if (!idx->shards.empty()) {   // the statistics are shard maxima, everything else is the same on every child
        const std::string o(name);
        if (o == "stripe_rows") return (double)idx->stripe;
        double v = rq_get_option(idx->shards[0], name);
        if (o.rfind("max_", 0) == 0 || o.rfind("eps_", 0) == 0 || o == "scan8_row_err" || o == "scan8_suspended")
            for (rq_index* c : idx->shards) v = std::max(v, rq_get_option(c, name));
        if (o == "scan8_used" || o == "hints_used") {   // counters: summed
            v = 0;
            for (rq_index* c : idx->shards) v += rq_get_option(c, name);
        }
        return v;
    }
    const std::string s(name);
    if (s == "ring") return idx->ring;
    if (s == "prefetch") return idx->prefetch;
    if (s == "kstage") return idx->kstage;
    if (s == "wide_batch") return idx->wide_batch;
    if (s == "wg_per_cu") return idx->wg_per_cu;
    if (s == "nt") return idx->nt;
    if (s == "slack_bins") return idx->slack_bins;
    if (s == "eps") return idx->eps < 0 ? RQ_EPS_DEFAULT : idx->eps;   // the base bound; "eps_cosine" / "eps_ip": with the shard's flush term
    if (s == "profile") return idx->profile;
    if (s == "fast_tail") return idx->fast_tail;
    if (s == "pipeline") return idx->pipeline;
    if (s == "profile_stride") return idx->profile_stride;
    if (s == "cu_count") return idx->cu_count;
    if (s == "max_row_norm") return idx->max_row_norm;
    if (s == "scan8") return idx->scan8;
    if (s == "thr_mult8") return idx->thr_mult8;
    if (s == "scan8_split") return idx->scan8_split;
    if (s == "wide8") return idx->wide8;
    if (s == "wide256_8") return idx->wide256_8;
    if (s == "scan8_row_err") return idx->x8_valid == idx->n && idx->x8 ? idx->max_e8 : -1.0;   // worst row's relative int8 error (-1: image not built)
    if (s == "scan8_suspended") return (idx->scan8_level[0] == 2 ? 1.0 : 0.0) + (idx->scan8_level[1] == 2 ? 2.0 : 0.0);   // bit 0: k <= 32, bit 1: larger k
    if (s == "scan8_wide_one_image") {   // per class (units: k <= 32, tens: larger k): 1 = its calls of more than 64 queries scan ONE int8 image per query now
        double r = 0;
        for (int c = 0; c < 2; ++c) {
            const int lvl = idx->scan8_level[c];
            const bool on = idx->scan8 && idx->wide8 && (lvl == 0 || (lvl == 1 && idx->scan8_split < 0 && !idx->wide1_off[c] && (idx->scan8 == 2 || idx->wide1_ok[c])));
            r += (c ? 10.0 : 1.0) * (on ? 1 : 0);
        }
        return r;
    }
    if (s == "scan8_level") return idx->scan8_level[0] + 10.0 * idx->scan8_level[1];   // per class: 0 one image, 1 two images, 2 fp16 scan   // too many repairs behind the int8 scan (rq_search_fixup_device)
    if (s == "scan8_calibrated_rows") return (double)idx->calib_rows;   // rows of the shard when the int8 ladder's start was last measured (0: never)
    // "scan8_calib_ms_<class><rung>" / "scan8_calib_unc_<class><rung>" (class 0: k <= 32, 1: larger; rung 0 one image, 1 two images,
    // 2 fp16 rows): milliseconds and uncertified queries of the 64-query sample search the calibration measured for that rung
    for (int unc = 0; unc < 2; ++unc) {
        const std::string pre = unc ? "scan8_calib_unc_" : "scan8_calib_ms_";
        if (s.rfind(pre, 0) == 0 && s.size() == pre.size() + 2) {
            const int c = s[pre.size()] - '0', l = s[pre.size() + 1] - '0';
            if (c >= 0 && c < 2 && l >= 0 && l < 3) return unc ? (double)idx->calib_unc[c][l] : (double)idx->calib_ms[c][l];
        }
    }
    if (s == "repaired_queries") return (double)idx->repaired_total;   // queries rq_search_fixup_device (or the blocking rq_search) had to repair so far
    if (s == "scan8_used") return (double)idx->scan8_used;   // searches that scanned the int8 image
    if (s == "hints_used") return (double)idx->hints_used;   // searches that found their queries prepared by the launch before them
    if (s == "max_sub_rel") return idx->max_sub_rel;   // largest share of a row's norm that sits in fp16-subnormal elements
    if (s == "max_sub_abs") return idx->max_sub_abs;
    if (s == "eps_cosine") return scan_eps(idx, RQ_METRIC_COSINE);
    if (s == "eps_ip") return scan_eps(idx, RQ_METRIC_IP);
    return NAN;
}

// ---- search ----------------------------------------------------------------------------------
// (Re)allocate a device buffer of want_elems elements; the old contents are dropped.  hipFree waits for the
// device, so kernels still using the old buffer have finished.
template <class T>
static int ensure(T*& p, size_t want_elems) {
    if (p) (void)hipFree(p);
    p = nullptr;
    hipError_t e = hipMalloc((void**)&p, want_elems * sizeof(T));
    if (e != hipSuccess) return set_err(RQ_ENOMEM, "workspace hipMalloc of %zu bytes failed: %s", want_elems * sizeof(T), hipGetErrorString(e));
    return RQ_OK;
}

static int ensure_ws(Workspace& w, int bpad, int64_t stride, int64_t m, size_t cand_elems) {
    const bool regrow_b = bpad > w.bcap;
    const int bcap = std::max(bpad, w.bcap);
    if (regrow_b) {
        if (int r = ensure(w.qh, (size_t)bcap * RQ_DPAD)) return r;
        if (int r = ensure(w.q32, (size_t)bcap * RQ_DPAD)) return r;
        if (int r = ensure(w.qn, (size_t)bcap)) return r;
        if (int r = ensure(w.q8, (size_t)bcap * RQ_DPAD)) return r;
        if (int r = ensure(w.qscale8, (size_t)bcap)) return r;
        if (int r = ensure(w.qeps8, (size_t)bcap)) return r;
        if (int r = ensure(w.q8lo, (size_t)bcap * RQ_DPAD)) return r;
        if (int r = ensure(w.qeps8s, (size_t)bcap)) return r;
        if (int r = ensure(w.wgmax, (size_t)bcap * RQ_WGMAX_STRIDE)) return r;
        if (int r = ensure(w.rowcount, (size_t)bcap)) return r;
        if (int r = ensure(w.thr, (size_t)bcap)) return r;
        if (int r = ensure(w.done, (size_t)bcap)) return r;
        if (int r = ensure(w.ovf, (size_t)bcap)) return r;
        w.counters_zero = false;
    }
    if (regrow_b || stride > w.bins_stride) {
        const int64_t st = std::max(stride, w.bins_stride);
        if (int r = ensure(w.bins, (size_t)bcap * st)) return r;
        w.bins_stride = st;
    }
    if (regrow_b || m > w.binkeys_cap) {
        const int64_t mm = std::max(m, w.binkeys_cap);
        if (int r = ensure(w.binkeys, (size_t)bcap * mm)) return r;
        w.binkeys_cap = mm;
    }
    if (cand_elems > w.cand_elems) {   // sized by the queries of the call, not by the padded slot count: an exact scan of
        if (int r = ensure(w.cand, cand_elems)) return r;   // one query holds a key for every row of the shard
        w.cand_elems = cand_elems;
    }
    w.bcap = bcap;
    return RQ_OK;
}

static int ensure_ones(rq_index* idx, hipStream_t s) {
    if (idx->ones && idx->ones_valid == idx->n) return RQ_OK;
    if (!idx->ones) HIPCHK(hipMalloc((void**)&idx->ones, (size_t)idx->cap * sizeof(float)));
    std::vector<float> h((size_t)idx->cap, std::nanf(""));   // pad rows: NaN, like inv_norm (see grow)
    std::fill(h.begin(), h.begin() + idx->n, RQ_QSCALE_INV);  // the queries carry 2^12 (rq_select.hip)
    HIPCHK(hipMemcpy(idx->ones, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    idx->ones_valid = idx->n;
    return RQ_OK;
}

#define RQ_SCAN8_MIN_ROWS 100000
// ... and k <= 128 (beyond that the candidate sets of the looser bound outweigh the bytes saved)
#define RQ_SCAN8_AUTO_MAX_K 128
#define RQ_SCAN8_SMALL_K 32     // up to here one int8 image per query, beyond two (run_pipeline)
#define RQ_SCAN8_MAX_ROW_ERR 0.03   // beyond that the candidate sets stop being small: such a shard keeps the fp16 scan

static int run_pipeline(rq_index* idx, const float* d_q, int B, int k, int metric, int nb, float* d_scores, int64_t* d_rows,
                        uint64_t* d_keys, int* d_status, hipStream_t s, bool may_defer = false, bool force_generic = false, bool allow8 = true);

// Where the int8 ladder STARTS on this shard ("scan8" = 1, the automatic rule), decided when the image is built instead of
// after slow batches (round 2 started every shard at one image / two images and let rq_search_fixup_device escalate: a
// clustered 1M-row corpus paid 4-8 batches of 0.5-0.7 ms, and a document-structured 125k-row shard kept an int8 scan that
// was twice as slow as the fp16 one).  64 STORED rows, evenly spread, are searched as queries -- on-topic queries are the
// hard case: their neighbourhoods are where the quantisation bound collects candidates -- through every rung (one image,
// two images, fp16 rows) for each class of k (k = 10 for k <= 32, k = 100 beyond), timed with HIP events (prep + scan + tail,
// plain sequential form, best of three).  A rung is eligible when at most 1 in 16 sample queries came back uncertified (the
// ladder's own rule); the LOWEST eligible rung wins unless a higher one is 8 % faster, the fp16 rows being always eligible.  Costs ~20 scans of the shard,
// once per image build (and again when the shard has doubled).  "scan8" = 2 (always) skips this and starts as round 2 did.
static int scan8_calibrate(rq_index* idx, hipStream_t s) {
    if (idx->scan8 != 1 || idx->calibrating || !idx->x8 || idx->n < 64 * 64) return RQ_OK;
    idx->calibrating = true;
    struct Done { rq_index* i; ~Done() { i->calibrating = false; } } done{idx};
    const int S = 64, KMAX = 100;
    std::vector<uint16_t> h16((size_t)S * RQ_DPAD);
    std::vector<float> h32((size_t)S * idx->dim);
    for (int i = 0; i < S; ++i) {
        const int64_t row = (int64_t)((double)i + 0.5) * idx->n / S;
        HIPCHK(hipMemcpy(h16.data() + (size_t)i * RQ_DPAD, idx->x + (size_t)std::min(row, idx->n - 1) * RQ_DPAD * 2, RQ_DPAD * 2, hipMemcpyDeviceToHost));
        for (int j = 0; j < idx->dim; ++j) {
            _Float16 v; __builtin_memcpy(&v, &h16[(size_t)i * RQ_DPAD + j], 2);
            h32[(size_t)i * idx->dim + j] = (float)v;
        }
    }
    float* d_q = nullptr; float* d_sc = nullptr; int64_t* d_rw = nullptr; int* d_st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = RQ_OK;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc((void**)&d_q, h32.size() * sizeof(float)));
        HIPCHK(hipMalloc((void**)&d_sc, (size_t)S * KMAX * sizeof(float)));
        HIPCHK(hipMalloc((void**)&d_rw, (size_t)S * KMAX * sizeof(int64_t)));
        HIPCHK(hipMalloc((void**)&d_st, (size_t)S * sizeof(int)));
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipMemcpy(d_q, h32.data(), h32.size() * sizeof(float), hipMemcpyHostToDevice));
        const int64_t used0 = idx->scan8_used;
        for (int c = 0; c < 2; ++c) {
            const int k = c == 0 ? 10 : KMAX;
            if ((int64_t)k * 2 * 64 > idx->n) { idx->scan8_level[c] = 2; continue; }
            float ms_of[3] = {0.f, 0.f, 0.f};
            int unc_of[3] = {0, 0, 0};
            for (int level = 2; level >= 0; --level) {          // fp16 first: always eligible
                idx->scan8_level[c] = level;
                float ms = 1e30f;
                int unc = 0;
                for (int rep = 0; rep < 4; ++rep) {             // (the first run warms the workspace of this shape)
                    HIPCHK(hipEventRecord(e0, s));
                    if (int r = run_pipeline(idx, d_q, S, k, RQ_METRIC_COSINE, nb_default(idx, k), d_sc, d_rw, nullptr, d_st, s, false, false, level < 2)) return r;
                    HIPCHK(hipEventRecord(e1, s));
                    HIPCHK(hipEventSynchronize(e1));
                    float t = 0.f;
                    HIPCHK(hipEventElapsedTime(&t, e0, e1));
                    if (rep > 0) ms = std::min(ms, t);
                }
                int st[64];
                HIPCHK(hipMemcpy(st, d_st, sizeof st, hipMemcpyDeviceToHost));
                for (int i = 0; i < S; ++i) unc += st[i] != 0;
                ms_of[level] = ms; unc_of[level] = unc;
            }
            // the lowest eligible rung, unless a higher one is clearly (8 %) faster: one image per query is also the only form with
            // wide int8 passes, and two rungs within the boxes' run-to-run noise must not flip the choice between processes
            int best = 2;
            for (int level = 1; level >= 0; --level)
                if (unc_of[level] * 16 <= S) best = level;
            for (int level = best + 1; level < 3; ++level)
                if ((level == 2 || unc_of[level] * 16 <= S) && ms_of[level] < 0.92f * ms_of[best]) best = level;
            idx->scan8_level[c] = best;
            idx->scan8_checked[c] = idx->scan8_repaired[c] = 0;
            idx->wide1_ok[c] = unc_of[0] * 16 <= S;      // one image is eligible on the sample (whichever rung 64-query calls were given)
            idx->wide1_off[c] = false;
            idx->wide1_checked[c] = idx->wide1_repaired[c] = 0;
            for (int l = 0; l < 3; ++l) { idx->calib_ms[c][l] = ms_of[l]; idx->calib_unc[c][l] = unc_of[l]; }
        }
        idx->scan8_used = used0;      // (the calibration's own scans are not the caller's searches)
        idx->calib_rows = idx->n;
        return RQ_OK;
    };
    const int level_before[2] = {idx->scan8_level[0], idx->scan8_level[1]};
    rc = body();
    if (rc != RQ_OK) { idx->scan8_level[0] = level_before[0]; idx->scan8_level[1] = level_before[1]; }   // (a failed measurement leaves no trial rung behind)
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    void* p[] = {d_q, d_sc, d_rw, d_st};
    for (void* q : p) if (q) (void)hipFree(q);
    return rc;
}

// int8 scan ("scan8"): bring the int8 image of the shard up to date (rows appended since the last search that used it) and
// read back the worst row's relative quantisation error.  One blocking 8-byte copy per append, nothing when up to date.
static int ensure_x8(rq_index* idx, hipStream_t s) {
    if (idx->x8 && idx->x8_valid == idx->n) {
        if (idx->scan8 == 1 && idx->calib_rows == 0 && !idx->calibrating && idx->max_e8 <= RQ_SCAN8_MAX_ROW_ERR) return scan8_calibrate(idx, s);
        return RQ_OK;
    }
    if (!idx->x8) {
        hipError_t e = hipMalloc((void**)&idx->x8, (size_t)idx->cap * RQ_DPAD);
        if (e == hipSuccess) e = hipMalloc((void**)&idx->scale8_cos, (size_t)idx->cap * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&idx->scale8_ip, (size_t)idx->cap * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&idx->binerr8, (size_t)(idx->cap / 64 + 64) * sizeof(float));   // (+64: the tail reads whole record strides)
        if (e == hipSuccess && !idx->d_stat8) e = hipMalloc((void**)&idx->d_stat8, sizeof(unsigned long long));
        if (e != hipSuccess) {   // no room for the image (+50 % of the shard): not an error, the fp16 rows remain the scan operand
            drop_x8(idx);
            (void)hipGetLastError();
            idx->scan8_level[0] = idx->scan8_level[1] = 2;
            return RQ_OK;
        }
        HIPCHK(hipMemsetAsync(idx->x8, 0, (size_t)idx->cap * RQ_DPAD, s));
        HIPCHK(hipMemsetAsync(idx->scale8_cos, 0xff, (size_t)idx->cap * sizeof(float), s));   // pad rows: NaN (see grow)
        HIPCHK(hipMemsetAsync(idx->scale8_ip, 0xff, (size_t)idx->cap * sizeof(float), s));
        HIPCHK(hipMemsetAsync(idx->d_stat8, 0, sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync(idx->binerr8, 0, (size_t)(idx->cap / 64 + 64) * sizeof(float), s));
        idx->x8_valid = 0; idx->max_e8 = 0.0;
    }
    HIPCHK(rq_quant_rows_launch(idx->x, idx->rownorm64, idx->x8_valid, idx->n, idx->x8, idx->scale8_cos, idx->scale8_ip, idx->d_stat8, idx->binerr8, s));
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, idx->d_stat8, sizeof bits, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    __builtin_memcpy(&idx->max_e8, &bits, sizeof bits);
    idx->x8_valid = idx->n;
    if (idx->scan8 == 1 && idx->max_e8 <= RQ_SCAN8_MAX_ROW_ERR && (idx->calib_rows == 0 || idx->n >= 2 * idx->calib_rows))
        return scan8_calibrate(idx, s);
    return RQ_OK;
}
// the shard's share of the int8 scan's bound (unit-query units; the query's own share is added per query by the tail):
// worst row + the fp32 steps between the exact int32 sum and the bin record (two scale products, two 6-bit truncations)
static inline float scan8_eps(const rq_index* idx) { return (float)(idx->max_e8 * 1.000001 + 2e-5); }
static int fill_empty(int B, int k, float* d_scores, int64_t* d_rows, uint64_t* d_keys, int* d_status, hipStream_t s) {
    HIPCHK(hipMemsetAsync(d_scores, 0, (size_t)B * k * sizeof(float), s));
    HIPCHK(hipMemsetAsync(d_rows, 0xff, (size_t)B * k * sizeof(int64_t), s));
    if (d_keys) HIPCHK(hipMemsetAsync(d_keys, 0, (size_t)B * k * sizeof(uint64_t), s));
    HIPCHK(hipMemsetAsync(d_status, 0, (size_t)B * sizeof(int), s));
    return RQ_OK;
}

// Test hook ("poison_cand"): before a tail runs, its queries' candidate lists are overwritten with the largest
// possible key.  A consumer that reads a candidate slot it was not handed (a stale line) then returns row 0 with a NaN
// score at rank 1, which no oracle comparison can miss -- instead of a plausible key of an earlier batch.
static int poison_cand(const rq_index* idx, const RqTailArgs& t, int B, hipStream_t s) {
    if (!idx->poison_cand || !t.cand || B <= 0) return RQ_OK;
    HIPCHK(hipMemsetAsync(t.cand, 0xff, (size_t)B * RQ_CAND_CAP * sizeof(uint64_t), s));
    return RQ_OK;
}

// Make `s` wait for every tail still running on the internal tail stream of `s` (pipeline = 1) and launch the
// tail that was waiting for the next scan (pipeline = 2).
static int flush_tails(rq_index* idx, hipStream_t s) {
    auto it = idx->ctx.find(s);
    if (it == idx->ctx.end()) return RQ_OK;
    StreamCtx& c = it->second;
    c.hint_q = nullptr;   // a flush ends the loop the hint belonged to (queries already prepared stay usable)
    if (c.fused_pending) {
        c.fused_pending = false;
        if (int r = poison_cand(idx, c.fused_tail, c.fused_B, s)) return r;
        HIPCHK(rq_tail_launch(c.fused_tail, c.fused_B, s));
    }
    for (int p = 0; p < 2; ++p)
        if (c.tail_pending[p]) {
            HIPCHK(hipStreamWaitEvent(s, c.ev_tail[p], 0));
            c.tail_pending[p] = false;
        }
    return RQ_OK;
}

// Every tail still waiting for a scan ("pipeline" = 2), of every stream the index remembers.  The callers' stream handles are
// NOT used for it (a caller may have destroyed a stream it no longer searches on; only rq_stream_release tells us): the device
// is drained first -- every scan those tails depend on has then finished -- and the tails run on the index's own stream.
static int flush_all(rq_index* idx) {
    bool any = false;
    for (auto& kv : idx->ctx) any = any || kv.second.fused_pending || kv.second.tail_pending[0] || kv.second.tail_pending[1];
    if (!any) return RQ_OK;
    RQ_ON_DEVICE(idx);
    HIPCHK(hipDeviceSynchronize());
    for (auto& kv : idx->ctx) {
        StreamCtx& c = kv.second;
        c.hint_q = nullptr;
        c.tail_pending[0] = c.tail_pending[1] = false;   // (their events have fired: the device is idle)
        if (!c.fused_pending) continue;
        c.fused_pending = false;
        if (int r = poison_cand(idx, c.fused_tail, c.fused_B, idx->own_stream)) return r;
        HIPCHK(rq_tail_launch(c.fused_tail, c.fused_B, idx->own_stream));
    }
    HIPCHK(hipStreamSynchronize(idx->own_stream));
    return RQ_OK;
}

static void free_ctx(StreamCtx& c);

// Drop the workspace of `only` (or of every stream when all = true) after the device has drained.
static int release_contexts(rq_index* idx, hipStream_t only, bool all) {
    if (int r = flush_all(idx)) return r;
    HIPCHK(hipDeviceSynchronize());
    for (auto it = idx->ctx.begin(); it != idx->ctx.end();) {
        if (all || it->first == only) { free_ctx(it->second); it = idx->ctx.erase(it); }
        else ++it;
    }
    return RQ_OK;
}

extern "C" int rq_stream_release(rq_index* idx, void* stream) {
    if (!idx) return set_err(RQ_EINVAL, "null index");
    if (!idx->shards.empty()) return RQ_OK;
    RQ_ON_DEVICE(idx);
    if (idx->ctx.find((hipStream_t)stream) == idx->ctx.end()) return RQ_OK;
    return release_contexts(idx, (hipStream_t)stream, false);
}

// One pass of the pipeline for B queries.  nb < 0: exact scan (every bin re-scored, no corpus scan).
// may_defer: the caller accepts results that are complete only after rq_search_flush_device ("pipeline" option).
static int run_pipeline(rq_index* idx, const float* d_q, int B, int k, int metric, int nb, float* d_scores, int64_t* d_rows,
                        uint64_t* d_keys, int* d_status, hipStream_t s, bool may_defer, bool force_generic, bool allow8) {
    if (idx->n == 0) return fill_empty(B, k, d_scores, d_rows, d_keys, d_status, s);
    const int binrows = RQ_BIN_ROWS;
    const int nquads = (int)((idx->n + 63) / 64);
    const int64_t nbins = nquads;   // bin = quad
    // int8 scan ("scan8": 0 = never; 1 = k <= RQ_SCAN8_AUTO_MAX_K on shards of RQ_SCAN8_MIN_ROWS rows and more; 2 = always): the scan
    // reads the int8 image of the shard when its worst row quantises well enough.  The size rule: on Gaussian rows the image
    // pays down to 125k rows (fused two-stream loop, us per batch int8 / fp16: 250k rows 35.0 / 59.8, 125k rows 25.0 / 29.0),
    // but a 125k-row document-structured shard takes 64 us against 34 in the same loop (profiles/r02_shard_shapes.txt).  Its bound does not
    // involve fp16 subnormals (the image is relative to each row's largest element), so it is decided BEFORE `exact` below.
    bool use8 = false;
    const int kclass = k <= RQ_SCAN8_SMALL_K ? 0 : 1;
    // Calls of more than 64 queries: passes of 128 queries over the image (two 16-query groups per wave, rq_scan.hip I8 = 3) while
    // the class runs with one image per query and "wide8" is on; otherwise the fp16 passes of rq_scan_wide.hip.
    // ... or while it runs with two images for its 64-query calls and one image is known to be good enough for the wide ones (wide1, rq_index.h)
    auto wide_level_ok = [&]() {
        const int lvl = idx->scan8_level[kclass];
        // (not under an explicit "scan8_split" = 1: the caller asked for two images everywhere, which no wide pass offers)
        return lvl == 0 || (lvl == 1 && idx->scan8_split < 0 && !idx->wide1_off[kclass] && (idx->scan8 == 2 || idx->wide1_ok[kclass] || idx->calib_rows == 0));
    };
    const bool wide_ok = B <= 64 || (idx->wide8 && idx->wide_batch != 0 && wide_level_ok());
    if (allow8 && idx->scan8 && idx->scan8_level[kclass] < 2 && nb >= 0 && 2 * (int64_t)nb < nbins && wide_ok && !force_generic && idx->fast_tail &&
        k <= RQ_FAST_MAX_K && (idx->scan8 == 2 || (idx->n >= RQ_SCAN8_MIN_ROWS && k <= RQ_SCAN8_AUTO_MAX_K))) {
        if (int r = ensure_x8(idx, s)) return r;   // (may calibrate: the class's level is read again below)
        use8 = idx->x8 && idx->x8_valid == idx->n && idx->max_e8 <= RQ_SCAN8_MAX_ROW_ERR && idx->scan8_level[kclass] < 2 &&
               (B <= 64 || (wide_level_ok() && (idx->scan8_level[kclass] == 0 || idx->scan8 == 2 || idx->wide1_ok[kclass])));
    }
    // Queries as ONE int8 image or as TWO (value + residual: the query's share of the bound vanishes, every corpus fragment
    // feeds two MFMAs).  Measured at 1M rows, fused loop: k = 10  132 us per batch with one image, 143-146 with two (the
    // scan stops being purely HBM-bound); k = 100  189 us with one, 153 with two (a third of the candidate rows).  "scan8_split"
    // -1 (default): one image for k <= 32, two beyond; 0 / 1: one / two for every k.  That is only where a class STARTS: when
    // more than 1 in 16 checked queries of a class needed repair, rq_search_fixup_device moves it one step along
    // one image -> two images -> fp16 scan (clustered corpus + random queries at k = 10: one image 19 of 64 queries repaired,
    // two images none, 245 us per batch against 275 with the fp16 scan).
    const bool split8 = use8 && idx->scan8_level[kclass] == 1 && B <= 64;      // (wide calls: one image)
    // tiny shards (fewer than two bins per wanted bin): the approximate pass cannot narrow anything down
    // ... and shards whose rows keep so much of their norm in fp16-subnormal elements that the fp16 scan's scores say nothing
    const bool exact = nb < 0 || 2 * (int64_t)nb >= nbins || (!use8 && idx->eps < 0 && scan_eps(idx, metric) > RQ_EPS_USELESS);
    if (exact) nb = (int)std::min<int64_t>(nbins, INT32_MAX / 64);
    if (!exact && nb > RQ_NB_MAX) return set_err(RQ_EINVAL, "nb %d too large", nb);
    // Queries per corpus pass: 64, or -- once a call has more than 64 -- 128 / 256 (every LDS fragment of the corpus feeds
    // two MFMAs per wave; option "wide_batch": 0 = passes of 64 only, 1 = 64/128/256, 2 = round 1's 8-wave 128-query pass,
    // 3 = 64/128 without the 256-query pass).  A call is cut into passes greedily: 256 while more than 128 queries remain,
    // then 128, then 64.
    int pass_q[1024], npass = 0, bpad = 0;
    {
        const int wb = idx->wide_batch;
        const int big = wb == 1 ? 256 : (wb == 2 || wb == 3 ? 128 : 64);
        for (int left = B; left > 0;) {
            int qb = 64;
            if (use8 && B > 64) {                               // int8 image: passes of 256 (rq_scan_wide.hip I8), 128 (rq_scan.hip I8 = 3) and 64 queries
                if (idx->wide256_8 && big >= 256 && left > 128) qb = 256;
                else if (left > 64 || !idx->wide256_8) qb = 128;
            }
            else if (big >= 256 && left > 128) qb = 256;
            else if (big >= 128 && left > 64) qb = 128;
            if (npass == 1024) return set_err(RQ_EINVAL, "too many passes");
            pass_q[npass++] = qb;
            bpad += qb;
            left -= qb;
        }
    }
    const int64_t stride = (nbins + 63) / 64 * 64;
    const int m = nb + 1;   // generic tail: bins re-scored + the first one that is not
    const bool fast = !exact && !force_generic && idx->fast_tail && k <= RQ_FAST_MAX_K;
    const int64_t ncand = fast ? (int64_t)RQ_CAND_CAP : (int64_t)nb * binrows;
    // Workspaces are kept per caller stream.  A caller that keeps creating streams (torch hands out a pool of 32) would
    // pile them up: beyond RQ_MAX_STREAM_CTX streams everything idle is dropped (rare; costs one device synchronisation).
    if (idx->ctx.find(s) == idx->ctx.end() && idx->ctx.size() >= RQ_MAX_STREAM_CTX)
        if (int r = release_contexts(idx, nullptr, true)) return r;
    StreamCtx& cx = idx->ctx[s];
    const bool piped = fast && may_defer && idx->pipeline == 1;
    // fused mode: one scan launch per call (<= 64 queries), which carries the tail of the previous call
    const bool fused = fast && may_defer && idx->pipeline == 2 && bpad == 64;
    int par = 0, slot = -1;
    if (fused) {
        slot = (int)(cx.calls % 3);
        par = (int)(cx.calls++ & 1);
        for (int p = 0; p < 3; ++p) {
            if (!cx.ring_qh[p]) HIPCHK(hipMalloc((void**)&cx.ring_qh[p], (size_t)64 * RQ_DPAD * sizeof(_Float16)));
            if (!cx.ring_q32[p]) HIPCHK(hipMalloc((void**)&cx.ring_q32[p], (size_t)64 * RQ_DPAD * sizeof(float)));
            if (!cx.ring_qn[p]) HIPCHK(hipMalloc((void**)&cx.ring_qn[p], (size_t)64 * sizeof(double)));
            if (!cx.ring_q8[p]) HIPCHK(hipMalloc((void**)&cx.ring_q8[p], (size_t)64 * RQ_DPAD));
            if (!cx.ring_qscale8[p]) HIPCHK(hipMalloc((void**)&cx.ring_qscale8[p], (size_t)64 * sizeof(float)));
            if (!cx.ring_qeps8[p]) HIPCHK(hipMalloc((void**)&cx.ring_qeps8[p], (size_t)64 * sizeof(float)));
            if (!cx.ring_q8lo[p]) HIPCHK(hipMalloc((void**)&cx.ring_q8lo[p], (size_t)64 * RQ_DPAD));
            if (!cx.ring_qeps8s[p]) HIPCHK(hipMalloc((void**)&cx.ring_qeps8s[p], (size_t)64 * sizeof(float)));
        }
    } else if (piped) {
        if (cx.fused_pending) { if (int r = flush_tails(idx, s)) return r; }
        if (!cx.tail) {
            // plain priority: a high-priority tail stream was measured to slow the scan it overlaps
            HIPCHK(hipStreamCreateWithFlags(&cx.tail, hipStreamNonBlocking));
            for (int p = 0; p < 2; ++p) {
                HIPCHK(hipEventCreateWithFlags(&cx.ev_scan[p], hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&cx.ev_tail[p], hipEventDisableTiming));
            }
        }
        par = (int)(cx.calls++ & 1);
        // this workspace was last used two calls ago: its tail must have finished before it is overwritten
        if (cx.tail_pending[par]) { HIPCHK(hipStreamWaitEvent(s, cx.ev_tail[par], 0)); cx.tail_pending[par] = false; }
    } else {
        if (int r = flush_tails(idx, s)) return r;   // order after anything still on the tail stream
    }
    if (!fused) cx.hint_q = nullptr;   // a hint is for the next FUSED call of the stream only
    Workspace& w = cx.w[par];
    if (int r = ensure_ws(w, bpad, exact ? 64 : stride, exact ? 1 : m, (size_t)B * (size_t)ncand)) return r;
    const float* scale = idx->inv_norm;
    if (metric == RQ_METRIC_IP) { if (int r = ensure_ones(idx, s)) return r; scale = idx->ones; }

    // counter protocol of the tail kernel: rowcount/done/ovf are zero on entry and the kernel leaves them zero
    if (fast && !w.counters_zero) {
        HIPCHK(hipMemsetAsync(w.rowcount, 0, (size_t)w.bcap * sizeof(int), s));
        HIPCHK(hipMemsetAsync(w.done, 0, (size_t)w.bcap * sizeof(int), s));
        HIPCHK(hipMemsetAsync(w.ovf, 0, (size_t)w.bcap * sizeof(int), s));
        w.counters_zero = true;
    }
    // unit-norm fp16 query fragments for the scan (+ padded fp32 queries / fp64 norms for the generic tail)
    _Float16* const qh = fused ? cx.ring_qh[slot] : w.qh;
    float* const q32 = fused ? cx.ring_q32[slot] : w.q32;
    double* const qn = fused ? cx.ring_qn[slot] : w.qn;
    signed char* const q8 = fused ? cx.ring_q8[slot] : w.q8;
    float* const qscale8 = fused ? cx.ring_qscale8[slot] : w.qscale8;
    float* const qeps8 = fused ? cx.ring_qeps8[slot] : w.qeps8;
    signed char* const q8lo = fused ? cx.ring_q8lo[slot] : w.q8lo;
    float* const qeps8s = fused ? cx.ring_qeps8s[slot] : w.qeps8s;
    if (use8) idx->scan8_used++;
    if (may_defer) { idx->last_use8 = use8; idx->last_wide1 = use8 && B > 64 && idx->scan8_level[kclass] == 1; }   // (may_defer: the caller's own search, not a repair pass of rq_search_fixup_device)
    // ... unless the previous launch of this stream has already prepared exactly these queries (rq_search_hint_next_device)
    const bool prepared = fused && cx.prepped_q == d_q && cx.prepped_B == B && cx.prepped_slot == slot;
    cx.prepped_q = nullptr;
    if (prepared) idx->hints_used++;
    if (!prepared) {
        RqPrepArgs me{};
        me.q = d_q; me.dim = idx->dim; me.B = B; me.nslots = bpad;
        me.qh = qh; me.q32pad = q32; me.qnorm64 = qn; me.q8 = q8; me.qscale8 = qscale8; me.qeps8 = qeps8; me.q8lo = q8lo; me.qeps8s = qeps8s;
        HIPCHK(rq_prep_queries_launch(me, s));
    }
    // the queries announced for the NEXT call are prepared by extra workgroups of this call's fused launch
    RqPrepArgs pa{};
    if (fused && cx.hint_q) {
        const int nslot = (slot + 1) % 3;
        pa.q = cx.hint_q; pa.dim = idx->dim; pa.B = cx.hint_B; pa.nslots = 64;
        pa.qh = cx.ring_qh[nslot]; pa.q32pad = cx.ring_q32[nslot]; pa.qnorm64 = cx.ring_qn[nslot];
        pa.q8 = cx.ring_q8[nslot]; pa.qscale8 = cx.ring_qscale8[nslot]; pa.qeps8 = cx.ring_qeps8[nslot];
        pa.q8lo = cx.ring_q8lo[nslot]; pa.qeps8s = cx.ring_qeps8s[nslot];
    }
    // Scan grids.  Every fp16 pass of more than 64 queries, and the int8 256-query pass, runs ONE 512-thread workgroup per CU ("wide");
    // the 64-query passes and the int8 128-query pass run wg_per_cu 256-thread workgroups per CU.  A call's passes are cut widest first,
    // so its wide passes precede its narrow ones: the tail is told where the second grid starts (nwg_split).  Until round 3 the first
    // pass's grid served the whole call, and the remainder pass of e.g. 384 int8 queries ran at half its occupancy (232 us instead of 150).
    auto pass_wide = [&](int qb) { return qb > 64 && (!use8 || qb == 256); };
    int wg_cu = idx->wg_per_cu;
    // Small int8 shards searched from SEVERAL caller streams (the per-rank shape of a multi-GPU run: 125k rows, two streams): two fused
    // launches are resident at once, so ONE scan workgroup per CU and launch already keeps two per CU streaming, and each lives twice as
    // long -- the prologue (48 KB of query fragments + the ring fill) is paid half as often.  Measured, two streams + exchange, us per
    // batch with 2 / 1 workgroups per CU: 125k rows 24.2 / 20.5, 250k rows 32.5 / 31.4; the fp16 rows are at the streaming rate either way
    // (28.7 / 29.7): int8 only, below 8 quads per workgroup, and only while another stream has a fused tail pending.
    if (fused && use8 && idx->wg_auto && wg_cu == 2 && (int64_t)nquads < (int64_t)16 * idx->cu_count) {
        bool other_stream_busy = false;
        for (auto& kv : idx->ctx) other_stream_busy = other_stream_busy || (kv.first != s && kv.second.fused_pending);
        if (other_stream_busy) wg_cu = 1;
    }
    const int grid_narrow = (int)std::min<int64_t>(std::min<int64_t>(nquads, RQ_WGMAX_STRIDE), (int64_t)idx->cu_count * wg_cu);
    const int grid_wide = (int)std::min<int64_t>(std::min<int64_t>(nquads, RQ_WGMAX_STRIDE), (int64_t)idx->cu_count);
    int nwg_split = bpad;                                          // first query of the first narrow pass
    for (int blk = npass - 1, q0 = bpad; blk >= 0 && !pass_wide(pass_q[blk]); --blk) nwg_split = (q0 -= pass_q[blk]);
    if (!exact) {
        // non-temporal loads only for shards that cannot stay in the 256 MiB Infinity Cache between two scans
        // (measured: 192 MB shard 36 us with default policy vs 39 us nt; 1.5 GB shard 250 us nt vs 285 us default)
        const bool nt = idx->nt < 0 ? (idx->n * (int64_t)(use8 ? RQ_DPAD : RQ_DPAD * 2) > ((int64_t)208 << 20)) : idx->nt != 0;
        for (int blk = 0, q0 = 0; blk < npass; q0 += pass_q[blk], ++blk) {
            const int qb = pass_q[blk];
            const int grid = q0 >= nwg_split ? grid_narrow : grid_wide;
            RqScanArgs a;
            a.i8 = 0; a.qscale = nullptr; a.qlo = nullptr;
            a.x = idx->x;
            a.row_scale = scale;
            a.qh = qh + (size_t)q0 * RQ_DPAD;
            a.bins = w.bins + (size_t)q0 * w.bins_stride;
            a.bins_stride = w.bins_stride;
            a.n_rows = idx->n;
            a.nquads = nquads;
            a.nq_valid = idx->scan_nostore == 1 ? 0 : std::min(qb, B - q0);
            a.wgmax = w.wgmax + (size_t)q0 * RQ_WGMAX_STRIDE;
            a.wgmax_stride = RQ_WGMAX_STRIDE;
            if (use8) {
                a.i8 = qb == 256 ? 4 : (qb == 128 ? 3 : (split8 ? 2 : 1)); a.qlo = q8lo; a.x = idx->x8; a.row_scale = metric == RQ_METRIC_IP ? idx->scale8_ip : idx->scale8_cos;
                a.qh = (const _Float16*)(q8 + (size_t)q0 * RQ_DPAD); a.qscale = qscale8 + q0;
            }
            const bool prof = idx->profile == 1 && idx->ev_used < 16384 && (idx->scan_seq++ % (uint64_t)idx->profile_stride) == 0;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (prof) {   // the event pair rides on the scan dispatch itself (kernel start / end time stamps, no barrier packets)
                if (idx->ev_used == idx->events.size()) {
                    hipEvent_t n0, n1;
                    HIPCHK(hipEventCreate(&n0));
                    HIPCHK(hipEventCreate(&n1));
                    idx->events.push_back({n0, n1});
                }
                e0 = idx->events[idx->ev_used].first; e1 = idx->events[idx->ev_used].second;
                if (idx->profile_legacy) { HIPCHK(hipEventRecord(e0, s)); e0 = e1 = nullptr; }   // A/B: hipEventRecord around the launch
            }
            if (fused && cx.fused_pending) {
                cx.fused_pending = false;
                if (int r = poison_cand(idx, cx.fused_tail, cx.fused_B, s)) return r;
                if (idx->tail_stop == 9) {   // development: fused kernel without its tail workgroups, tail launched after it
                    HIPCHK(rq_scan_tail_launch(a, cx.fused_tail, 0, pa, nt, grid, idx->epi, s, e0, e1));
                    RqTailArgs t9 = cx.fused_tail; t9.stop_after = 0;
                    HIPCHK(rq_tail_launch(t9, cx.fused_B, s));
                } else
                HIPCHK(rq_scan_tail_launch(a, cx.fused_tail, cx.fused_B, pa, nt, grid, idx->epi, s, e0, e1));
            } else if (fused && pa.nslots) {   // first call of a loop: no tail to carry yet, but queries to prepare
                RqTailArgs none{};
                none.nbins = nbins; none.m = none.k = 1; none.thr_mult = 2.25f; none.thr_slack = 0.f;
                HIPCHK(rq_scan_tail_launch(a, none, 0, pa, nt, grid, idx->epi, s, e0, e1));
            } else if (fused) HIPCHK(rq_scan_launch(a, 3, 1, 2, 4, nt, grid, idx->epi, s, e0, e1));
            else if (use8 && qb == 256) HIPCHK(rq_scan_wide_launch(a, idx->wide256_8, 256, nt, grid, s, e0, e1));   // 256 queries over the int8 image
            else if (use8) HIPCHK(rq_scan_launch(a, 3, 1, 2, 4, nt, grid, 1, s, e0, e1));   // 64 queries, or 128 (a.i8 = 3)
            else if (qb == 256) HIPCHK(rq_scan_wide_launch(a, idx->wide256, 256, nt, grid, s, e0, e1));
            else if (qb == 128 && idx->wide_batch == 2) HIPCHK(rq_scan_launch(a, 3, 4, 1, 8, nt, grid, 0, s, e0, e1));   // round 1's 8-wave pass
            else if (qb == 128) HIPCHK(rq_scan_wide_launch(a, idx->wide128, 128, nt, grid, s, e0, e1));
            else HIPCHK(rq_scan_launch(a, idx->ring, idx->prefetch, idx->kstage, 4, nt, grid, idx->epi, s, e0, e1));
            if (fused && pa.nslots) {
                cx.prepped_q = cx.hint_q; cx.prepped_B = cx.hint_B; cx.prepped_slot = (slot + 1) % 3;
                cx.hint_q = nullptr;
            }
            if (prof) { if (idx->profile_legacy) HIPCHK(hipEventRecord(idx->events[idx->ev_used].second, s)); idx->ev_used++; idx->ev_bytes += idx->n * (int64_t)(use8 ? RQ_DPAD : RQ_DPAD * 2); }
        }
        if (fast) {
            hipStream_t ts = s;
            if (piped) {
                HIPCHK(hipEventRecord(cx.ev_scan[par], s));
                HIPCHK(hipStreamWaitEvent(cx.tail, cx.ev_scan[par], 0));
                ts = cx.tail;
            }
            RqTailArgs ta;
            ta.q = d_q; ta.dim = idx->dim; ta.x = idx->x; ta.rownorm64 = idx->rownorm64; ta.n_rows = idx->n;
            ta.bins = w.bins; ta.bins_stride = w.bins_stride; ta.nbins = nbins;
            ta.wgmax = w.wgmax; ta.wgmax_stride = RQ_WGMAX_STRIDE; ta.nwg = grid_wide; ta.nwg_split = nwg_split; ta.nwg2 = grid_narrow;
            ta.m = (int)std::min<int64_t>(k, idx->n); ta.metric = metric; ta.k = k;
            ta.eps = use8 ? scan8_eps(idx) : scan_eps(idx, metric);
            ta.qeps = use8 ? (split8 ? qeps8s : qeps8) : nullptr;
            ta.binerr = use8 && idx->bin_bound ? idx->binerr8 : nullptr;
            ta.eps_rows_max = use8 ? (float)idx->max_e8 : 0.f;
            // int8 scan: T = P - bound - slack.  The slack covers how far the k-th EXACT score may sit below P (= a k-th largest
            // APPROXIMATE score, biased upward by the errors of the rows that won); it is (thr_mult8 - 1) x the larger of the
            // query's own bound and the one-image bound of a typical query -- also when the queries are split (their bound is
            // smaller, the rows' errors are not)
            ta.thr_mult = use8 ? (float)idx->thr_mult8 : 2.25f;
            ta.thr_slack = use8 ? (float)(idx->max_e8 + 0.009) : 0.f;
            ta.local_topk = idx->tail_local;
            ta.max_row_norm = (float)(idx->max_row_norm * (1.0 + 1e-6)); ta.row_offset = idx->row_offset;
            ta.cand = w.cand; ta.rowcount = w.rowcount; ta.done = w.done; ta.ovf = w.ovf;
            ta.out_scores = d_scores; ta.out_rows = d_rows; ta.out_keys = d_keys; ta.out_status = d_status;
            ta.dbg = idx->dbg_stamps;
            ta.stop_after = idx->tail_stop;
            ta.fused_nv = idx->fused_nv;
            if (idx->tail_stop) w.counters_zero = false;   // a truncated tail does not reset its counters
            if (fused) {
                // the tail runs with the NEXT scan launch (or at the flush): it reads the workspace's own copy of the
                // queries, so the caller's buffer is free as soon as this call's work has run
                ta.q = q32; ta.dim = RQ_DPAD;
                cx.fused_tail = ta; cx.fused_B = B; cx.fused_pending = true;
                return RQ_OK;
            }
            const bool tprof = idx->profile == 2 && idx->ev_used < 16384;   // profile = 2: time the tail instead of the scan
            if (tprof) {
                if (idx->ev_used == idx->events.size()) {
                    hipEvent_t e0, e1;
                    HIPCHK(hipEventCreate(&e0));
                    HIPCHK(hipEventCreate(&e1));
                    idx->events.push_back({e0, e1});
                }
                HIPCHK(hipEventRecord(idx->events[idx->ev_used].first, ts));
            }
            if (int r = poison_cand(idx, ta, B, ts)) return r;
            HIPCHK(rq_tail_launch(ta, B, ts));
            if (tprof) { HIPCHK(hipEventRecord(idx->events[idx->ev_used].second, ts)); idx->ev_used++; idx->ev_bytes += idx->n * (int64_t)(use8 ? RQ_DPAD : RQ_DPAD * 2); }
            if (piped) { HIPCHK(hipEventRecord(cx.ev_tail[par], cx.tail)); cx.tail_pending[par] = true; }
            return RQ_OK;
        }
        HIPCHK(rq_select_bins_launch(w.bins, w.bins_stride, nbins, B, m, w.binkeys, s));
    }
    RqRescoreArgs ra;
    ra.x = idx->x; ra.q32 = w.q32; ra.qnorm64 = w.qn; ra.rownorm64 = idx->rownorm64;
    ra.binkeys = exact ? nullptr : w.binkeys; ra.binkeys_stride = m; ra.nb = nb; ra.metric = metric;
    ra.n_rows = idx->n; ra.cand = w.cand;
    if (exact && idx->exact_mfma) HIPCHK(rq_exact_scan_launch(ra, B, idx->cu_count, s));   // the whole shard: fp64 contraction on the matrix cores
    else HIPCHK(rq_rescore_launch(ra, B, s));
    RqFinalArgs fa;
    fa.cand = w.cand; fa.ncand = (int)ncand; fa.binkeys = w.binkeys; fa.binkeys_stride = m; fa.nb = nb; fa.nbins = exact ? nb : nbins;
    fa.qnorm64 = w.qn; fa.metric = metric; fa.eps = scan_eps(idx, metric);
    fa.max_row_norm = (float)(idx->max_row_norm * (1.0 + 1e-6)); fa.k = k; fa.row_offset = idx->row_offset; fa.n_rows = idx->n;
    fa.out_scores = d_scores; fa.out_rows = d_rows; fa.out_keys = d_keys; fa.out_status = d_status;
    HIPCHK(rq_final_launch(fa, B, s));
    return RQ_OK;
}

static int check_search_args(const rq_index* idx, const void* q, int B, int k, int metric, const void* sc, const void* rows) {
    if (!idx || !q || !sc || !rows) return set_err(RQ_EINVAL, "null argument");
    if (B < 1 || B > 65535) return set_err(RQ_EINVAL, "B %d outside 1..65535", B);
    if (k < 1 || k > RQ_MAX_K) return set_err(RQ_EINVAL, "k %d outside 1..%d", k, RQ_MAX_K);
    if (metric != RQ_METRIC_COSINE && metric != RQ_METRIC_IP) return set_err(RQ_EINVAL, "unknown metric %d", metric);
    return RQ_OK;
}

extern "C" int rq_search_device(rq_index* idx, const float* d_queries, int B, int k, int metric, float* d_scores, int64_t* d_rows,
                                uint64_t* d_keys, int* d_status, void* stream) {
    if (int r = check_search_args(idx, d_queries, B, k, metric, d_scores, d_rows)) return r;
    if (!d_status) return set_err(RQ_EINVAL, "d_status is required");
    if (!idx->shards.empty()) return set_err(RQ_EUNSUPPORTED, "device-pointer searches on a multi-device index: use rq_search (host buffers), or one index per device");
    RQ_ON_DEVICE(idx);
    idx->t.searches++;
    idx->t.queries += B;
    return run_pipeline(idx, d_queries, B, k, metric, nb_default(idx, k), d_scores, d_rows, d_keys, d_status, (hipStream_t)stream, true);
}

// The queries of the NEXT rq_search_device call on `stream` ("pipeline" = 2 loops): the call made right after this one
// prepares them with 64 extra workgroups of its own launch, and the call after that -- if it is given exactly d_next_queries
// and B -- skips its preparation launch.  Advisory: anything else simply prepares its queries itself.
extern "C" int rq_search_hint_next_device(rq_index* idx, const float* d_next_queries, int B, void* stream) {
    if (!idx) return set_err(RQ_EINVAL, "null index");
    if (!idx->shards.empty()) return set_err(RQ_EUNSUPPORTED, "device-pointer searches on a multi-device index: use rq_search (host buffers), or one index per device");
    if (B < 0 || B > 65535) return set_err(RQ_EINVAL, "B %d outside 0..65535", B);
    // Advisory.  A hint for a stream no search has run on yet only makes a host-side record (no device memory, no device call:
    // hence no device guard here); the workspace is allocated by the search that follows.  Beyond RQ_MAX_STREAM_CTX remembered
    // streams the hint is dropped instead (the call then prepares its own queries).  Remembered stream handles are never used
    // again by the library on its own account (flush_all runs pending tails on the index's own stream), so a record for a stream
    // the caller destroys later is harmless.
    if (idx->ctx.find((hipStream_t)stream) == idx->ctx.end() && idx->ctx.size() >= RQ_MAX_STREAM_CTX) return RQ_OK;
    StreamCtx& c = idx->ctx[(hipStream_t)stream];
    const bool usable = d_next_queries && B >= 1 && B <= 64 && idx->use_hint && idx->pipeline == 2;
    c.hint_q = usable ? d_next_queries : nullptr;
    c.hint_B = usable ? B : 0;
    return RQ_OK;
}

extern "C" int rq_search_train_device(rq_index* idx, int n_batches, const float* const* d_queries, int B, int k, int metric,
                                      float* const* d_scores, int64_t* const* d_rows, uint64_t* const* d_keys, int* const* d_status,
                                      void* const* streams, int n_streams) {
    if (!idx || n_batches < 0 || !d_queries || !d_scores || !d_rows || !d_status || !streams || n_streams < 1 || n_streams > 8)
        return set_err(RQ_EINVAL, "bad train arguments");
    if (!idx->shards.empty()) return set_err(RQ_EUNSUPPORTED, "device-pointer searches on a multi-device index: use rq_search (host buffers), or one index per device");
    for (int i = 0; i < n_batches; ++i) {
        if (int r = check_search_args(idx, d_queries[i], B, k, metric, d_scores[i], d_rows[i])) return r;
        if (!d_status[i]) return set_err(RQ_EINVAL, "d_status is required");
    }
    RQ_ON_DEVICE(idx);
    for (int i = 0; i < n_batches; ++i) {
        hipStream_t s = (hipStream_t)streams[i % n_streams];
        if (const float* nxt = d_queries[i + n_streams])
            if (int r = rq_search_hint_next_device(idx, nxt, B, s)) return r;
        idx->t.searches++;
        idx->t.queries += B;
        if (int r = run_pipeline(idx, d_queries[i], B, k, metric, nb_default(idx, k), d_scores[i], d_rows[i], d_keys ? d_keys[i] : nullptr, d_status[i], s, true))
            return r;
    }
    return RQ_OK;
}

extern "C" int rq_search_flush_device(rq_index* idx, void* stream) {
    if (!idx) return set_err(RQ_EINVAL, "null index");
    if (!idx->shards.empty()) return RQ_OK;
    RQ_ON_DEVICE(idx);
    return flush_tails(idx, (hipStream_t)stream);
}

// The int8 scan bets that real errors stay well below its worst-case bound (threshold multiplier thr_mult8 < 2) and that few
// rows sit within that bound of the k-th score.  A shard / query mix on which either fails shows up as repairs: beyond 1 in 16
// CHECKED queries (windows of 256) the class of k moves one step along one image -> two images -> fp16 scan, until "scan8" /
// "scan8_split" is set again.  Every checked query counts, the clean ones too (rq_search_end's clean branch reports them: a
// server answering one query per call must not see only its failures), whatever the size of the call.
static void scan8_account(rq_index* idx, int k, int checked, int repaired) {
    if (!idx->last_use8 || !idx->x8 || !idx->scan8) return;
    const int kclass = k <= RQ_SCAN8_SMALL_K ? 0 : 1;
    if (idx->scan8_level[kclass] >= 2) return;
    if (idx->last_wide1) {       // a wide call on one image in a two-image class: its repairs decide about the wide calls only
        idx->wide1_checked[kclass] += checked; idx->wide1_repaired[kclass] += repaired;
        if (idx->wide1_checked[kclass] >= 256) {
            if (idx->wide1_repaired[kclass] * 16 > idx->wide1_checked[kclass]) idx->wide1_off[kclass] = true;
            idx->wide1_checked[kclass] = idx->wide1_repaired[kclass] = 0;
        }
        return;
    }
    idx->scan8_checked[kclass] += checked; idx->scan8_repaired[kclass] += repaired;
    if (idx->scan8_checked[kclass] >= 256) {
        if (idx->scan8_repaired[kclass] * 16 > idx->scan8_checked[kclass]) idx->scan8_level[kclass]++;
        idx->scan8_checked[kclass] = idx->scan8_repaired[kclass] = 0;
    }
}

extern "C" int rq_search_fixup_device(rq_index* idx, const float* d_queries, int B, int k, int metric, float* d_scores,
                                      int64_t* d_rows, uint64_t* d_keys, int* d_status, void* stream) {
    if (int r = check_search_args(idx, d_queries, B, k, metric, d_scores, d_rows)) return r;
    if (!d_status) return set_err(RQ_EINVAL, "d_status is required");
    if (!idx->shards.empty()) return set_err(RQ_EUNSUPPORTED, "device-pointer searches on a multi-device index: use rq_search (host buffers), or one index per device");
    RQ_ON_DEVICE(idx);
    hipStream_t s = (hipStream_t)stream;
    if (int r = flush_tails(idx, s)) return r;
    std::vector<int> st((size_t)B);
    HIPCHK(hipMemcpyAsync(st.data(), d_status, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<int> bad;
    for (int q = 0; q < B; ++q) if (st[q] != 0) bad.push_back(q);
    scan8_account(idx, k, B, (int)bad.size());
    if (bad.empty()) return 0;
    const int repaired = (int)bad.size();
    idx->repaired_total += repaired;
    Workspace& w = idx->ctx[s].w[0];
    const int fb = (int)bad.size();
    if (fb > w.fix_bcap || k > w.fix_k) {
        const int nb_ = std::max(fb, w.fix_bcap), nk = std::max(k, w.fix_k);
        if (int r = ensure(w.fix_q, (size_t)nb_ * RQ_MAX_DIM)) return r;
        if (int r = ensure(w.fix_scores, (size_t)nb_ * nk)) return r;
        if (int r = ensure(w.fix_rows, (size_t)nb_ * nk)) return r;
        if (int r = ensure(w.fix_keys, (size_t)nb_ * nk)) return r;
        if (int r = ensure(w.fix_status, (size_t)nb_)) return r;
        w.fix_bcap = nb_; w.fix_k = nk;
    }
    // ladder: (queries that came from the int8 scan: the fp16 scan, whose threshold certifies by construction,) 4x wider
    // candidate set, then the full fp64 scan
    const int nb1 = std::min(RQ_NB_MAX, 4 * nb_default(idx, k));
    const bool from8 = idx->last_use8 && idx->x8 && idx->scan8;
    for (int level = from8 ? -1 : 0; level < 2 && !bad.empty(); ++level) {
        const int nbq = (int)bad.size();
        for (int i = 0; i < nbq; ++i)
            HIPCHK(hipMemcpyAsync(w.fix_q + (size_t)i * idx->dim, d_queries + (size_t)bad[i] * idx->dim, (size_t)idx->dim * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
        if (level < 0) {
            if (int r = run_pipeline(idx, w.fix_q, nbq, k, metric, nb_default(idx, k), w.fix_scores, w.fix_rows, w.fix_keys, w.fix_status, s, false, false, false)) return r;
        } else if (level == 0) {
            idx->t.widened += nbq;
            // (the fast tail fails only when its candidate lists overflow: the wider pass uses the generic sorted tail)
            if (int r = run_pipeline(idx, w.fix_q, nbq, k, metric, nb1, w.fix_scores, w.fix_rows, w.fix_keys, w.fix_status, s, false, true)) return r;
        } else {
            idx->t.exact_scans += nbq;
            // bound the candidate memory: a few queries per exact pass
            const int64_t per_q = ((idx->n + 63) / 64) * 64 * (int64_t)sizeof(uint64_t);
            const int group = (int)std::max<int64_t>(1, std::min<int64_t>(nbq, ((int64_t)1 << 30) / std::max<int64_t>(per_q, 1)));
            for (int off = 0; off < nbq; off += group) {
                const int g = std::min(group, nbq - off);
                if (int r = run_pipeline(idx, w.fix_q + (size_t)off * idx->dim, g, k, metric, -1, w.fix_scores + (size_t)off * k,
                                         w.fix_rows + (size_t)off * k, w.fix_keys + (size_t)off * k, w.fix_status + off, s))
                    return r;
            }
        }
        std::vector<int> st2((size_t)nbq);
        HIPCHK(hipMemcpyAsync(st2.data(), w.fix_status, (size_t)nbq * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<int> still;
        for (int i = 0; i < nbq; ++i) {
            if (st2[i] == 0) {
                const int q = bad[i];
                HIPCHK(hipMemcpyAsync(d_scores + (size_t)q * k, w.fix_scores + (size_t)i * k, (size_t)k * sizeof(float), hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(d_rows + (size_t)q * k, w.fix_rows + (size_t)i * k, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
                if (d_keys) HIPCHK(hipMemcpyAsync(d_keys + (size_t)q * k, w.fix_keys + (size_t)i * k, (size_t)k * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemsetAsync(d_status + q, 0, sizeof(int), s));
            } else {
                still.push_back(bad[i]);
            }
        }
        HIPCHK(hipStreamSynchronize(s));
        bad.swap(still);
    }
    if (!bad.empty()) return set_err(RQ_EHIP, "internal: %zu queries uncertified after the exact scan", bad.size());
    return repaired;
}

// The blocking host-buffer search in two halves, so that a multi-device parent can enqueue on every device before it
// waits for any: search_begin stages the queries and enqueues the search on the index's own stream, search_end waits,
// repairs uncertified queries and hands the results over.
int rq_search_begin(rq_index* idx, const float* queries, int B, int k, int metric) {
    RQ_ON_DEVICE(idx);
    if (B > idx->h_bcap || k > idx->h_kcap) {
        const int nb = std::max(B, idx->h_bcap), nk = std::max(k, idx->h_kcap);
        if (int r = ensure(idx->h_dq, (size_t)nb * RQ_MAX_DIM)) return r;
        if (int r = ensure(idx->h_dscores, (size_t)nb * nk)) return r;
        if (int r = ensure(idx->h_drows, (size_t)nb * nk)) return r;
        if (int r = ensure(idx->h_dstatus, (size_t)nb)) return r;
        idx->h_bcap = nb; idx->h_kcap = nk;
    }
    hipStream_t s = idx->own_stream;
    idx->hs_small = (size_t)B * k <= 65536;
    if (idx->hs_small) {
        // one device block [rows int64 | scores fp32 | status int32] -> one copy into pinned memory -> one synchronisation
        const size_t off_s = (size_t)B * k * sizeof(int64_t), off_t = off_s + (size_t)B * k * sizeof(float);
        const size_t bytes = off_t + (size_t)B * sizeof(int), qfloats = (size_t)B * idx->dim;
        if (bytes > idx->hs_bytes) {
            if (idx->hs_dev) (void)hipFree(idx->hs_dev);
            if (idx->hs_pin) (void)hipHostFree(idx->hs_pin);
            idx->hs_dev = nullptr; idx->hs_pin = nullptr; idx->hs_bytes = 0;
            HIPCHK(hipMalloc((void**)&idx->hs_dev, bytes));
            HIPCHK(hipHostMalloc((void**)&idx->hs_pin, bytes, hipHostMallocDefault));
            idx->hs_bytes = bytes;
        }
        if (qfloats > idx->hs_qfloats) {
            if (idx->hs_pin_q) (void)hipHostFree(idx->hs_pin_q);
            idx->hs_pin_q = nullptr; idx->hs_qfloats = 0;
            HIPCHK(hipHostMalloc((void**)&idx->hs_pin_q, qfloats * sizeof(float), hipHostMallocDefault));
            idx->hs_qfloats = qfloats;
        }
        int64_t* d_rows = (int64_t*)idx->hs_dev;
        float* d_scores = (float*)(idx->hs_dev + off_s);
        int* d_status = (int*)(idx->hs_dev + off_t);
        std::memcpy(idx->hs_pin_q, queries, qfloats * sizeof(float));
        HIPCHK(hipMemcpyAsync(idx->h_dq, idx->hs_pin_q, qfloats * sizeof(float), hipMemcpyHostToDevice, s));
        if (int r = rq_search_device(idx, idx->h_dq, B, k, metric, d_scores, d_rows, nullptr, d_status, s)) return r;
        if (int r = flush_tails(idx, s)) return r;   // (option "pipeline": the tail must have run before the copy)
        HIPCHK(hipMemcpyAsync(idx->hs_pin, idx->hs_dev, bytes, hipMemcpyDeviceToHost, s));
        return RQ_OK;
    }
    HIPCHK(hipMemcpyAsync(idx->h_dq, queries, (size_t)B * idx->dim * sizeof(float), hipMemcpyHostToDevice, s));
    return rq_search_device(idx, idx->h_dq, B, k, metric, idx->h_dscores, idx->h_drows, nullptr, idx->h_dstatus, s);
}

int rq_search_end(rq_index* idx, int B, int k, int metric, float* out_scores, int64_t* out_rows) {
    RQ_ON_DEVICE(idx);
    hipStream_t s = idx->own_stream;
    if (idx->hs_small) {
        const size_t off_s = (size_t)B * k * sizeof(int64_t), off_t = off_s + (size_t)B * k * sizeof(float);
        int64_t* d_rows = (int64_t*)idx->hs_dev;
        float* d_scores = (float*)(idx->hs_dev + off_s);
        int* d_status = (int*)(idx->hs_dev + off_t);
        HIPCHK(hipStreamSynchronize(s));
        const int* st = (const int*)(idx->hs_pin + off_t);
        bool clean = true;
        for (int q = 0; q < B; ++q) clean = clean && st[q] == 0;
        if (!clean) {   // rare: repair on the device, fetch again
            const int fr = rq_search_fixup_device(idx, idx->h_dq, B, k, metric, d_scores, d_rows, nullptr, d_status, s);
            if (fr < 0) return fr;
            HIPCHK(hipMemcpyAsync(idx->hs_pin, idx->hs_dev, off_t, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        } else scan8_account(idx, k, B, 0);   // the int8 ladder sees the clean calls too
        std::memcpy(out_rows, idx->hs_pin, off_s);
        std::memcpy(out_scores, idx->hs_pin + off_s, off_t - off_s);
        return RQ_OK;
    }
    const int fr = rq_search_fixup_device(idx, idx->h_dq, B, k, metric, idx->h_dscores, idx->h_drows, nullptr, idx->h_dstatus, s);
    if (fr < 0) return fr;
    HIPCHK(hipMemcpyAsync(out_scores, idx->h_dscores, (size_t)B * k * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_rows, idx->h_drows, (size_t)B * k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RQ_OK;
}

extern "C" int rq_search(rq_index* idx, const float* queries, int B, int k, int metric, float* out_scores, int64_t* out_rows) {
    if (int r = check_search_args(idx, queries, B, k, metric, out_scores, out_rows)) return r;
    if (!idx->shards.empty()) return rq_multi_search(idx, queries, B, k, metric, out_scores, out_rows);
    if (int r = rq_search_begin(idx, queries, B, k, metric)) return r;
    return rq_search_end(idx, B, k, metric, out_scores, out_rows);
}

extern "C" int rq_merge_keys_device(const uint64_t* d_keys_in, int n_per_query, int B, int k, float* d_scores, int64_t* d_rows,
                                    uint64_t* d_keys_out, void* stream) {
    if (!d_keys_in || !d_scores || !d_rows || B < 1 || k < 1 || k > RQ_MAX_K || n_per_query < 0) return set_err(RQ_EINVAL, "bad merge arguments");
    HIPCHK(rq_merge_keys_launch(d_keys_in, n_per_query, B, k, d_scores, d_rows, d_keys_out, (hipStream_t)stream));
    return RQ_OK;
}

// ---- development hook: wall-clock (start, end) stamps of every workgroup of the LAST fused launch ------------
extern "C" int rq_debug_stamps(rq_index* idx, int enable, unsigned long long* out, int max_wgs) {
    if (!idx) return set_err(RQ_EINVAL, "null index");
    if (!idx->shards.empty()) return set_err(RQ_EUNSUPPORTED, "debug hooks work on single-device indexes");
    RQ_ON_DEVICE(idx);
    HIPCHK(hipDeviceSynchronize());
    if (enable && !idx->dbg_stamps) { HIPCHK(hipMalloc((void**)&idx->dbg_stamps, 4 * 8192 * sizeof(unsigned long long))); HIPCHK(hipMemset(idx->dbg_stamps, 0, 4 * 8192 * sizeof(unsigned long long))); }
    if (out && idx->dbg_stamps) HIPCHK(hipMemcpy(out, idx->dbg_stamps, (size_t)4 * std::min(max_wgs, 8192) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (!enable && idx->dbg_stamps) { (void)hipFree(idx->dbg_stamps); idx->dbg_stamps = nullptr; }
    return RQ_OK;
}

// ---- measurement hook: plain streaming read of the shard (see rq_read_probe_kernel) ------------------------
extern "C" double rq_debug_read_bandwidth(rq_index* idx, int iters, int nt, int wg_per_cu) {
    if (!idx || iters < 1 || iters > 1000 || wg_per_cu < 1 || wg_per_cu > 32 || !idx->shards.empty()) { set_err(RQ_EINVAL, "bad arguments"); return -1.0; }
    DeviceGuard dg_(idx->device);
    if (!dg_.ok) { set_err(RQ_EHIP, "cannot select device %d", idx->device); return -1.0; }
    if (idx->n == 0) { set_err(RQ_EINVAL, "empty index"); return -1.0; }
    if (flush_all(idx)) return -1.0;
    const int64_t bytes = idx->n * (int64_t)(RQ_DPAD * 2);
    const bool use_nt = nt < 0 ? bytes > ((int64_t)208 << 20) : nt != 0;
    uint32_t* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double gbs = -1.0;
    hipStream_t s = idx->own_stream;
    do {
        if (hipMalloc((void**)&sink, 64) != hipSuccess) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        const int grid = idx->cu_count * wg_per_cu;
        if (rq_read_probe_launch(idx->x, bytes, use_nt, grid, sink, s) != hipSuccess) break;   // warm-up
        if (hipEventRecord(e0, s) != hipSuccess) break;
        bool ok = true;
        for (int i = 0; i < iters && ok; ++i) ok = rq_read_probe_launch(idx->x, bytes, use_nt, grid, sink, s) == hipSuccess;
        if (!ok || hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) break;
        gbs = (double)bytes * iters / (ms * 1e-3) / 1e9;
    } while (0);
    if (gbs < 0) set_err(RQ_EHIP, "read probe failed: %s", hipGetErrorString(hipGetLastError()));
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (sink) (void)hipFree(sink);
    return gbs;
}

// ---- test hook: the scan's per-bin maxima of the last search on `stream` ---------------------------------
extern "C" int64_t rq_debug_pooled(rq_index* idx, void* stream, int query, float* out, int64_t max_bins) {
    if (!idx || !out || query < 0 || !idx->shards.empty()) return set_err(RQ_EINVAL, "bad arguments");
    RQ_ON_DEVICE(idx);
    auto it = idx->ctx.find((hipStream_t)stream);
    if (it == idx->ctx.end() || !it->second.w[0].bins || query >= it->second.w[0].bcap) return set_err(RQ_EINVAL, "no search has run on this stream");
    const Workspace& w = it->second.w[0];
    const int64_t nbins = (idx->n + 63) / 64;
    const int64_t n = std::min(nbins, max_bins);
    HIPCHK(hipDeviceSynchronize());
    // field x of every 8-byte record: the bin's largest approximate score (26 bits, rounded up) | its row
    std::vector<uint32_t> raw((size_t)n);
    HIPCHK(hipMemcpy2D(raw.data(), sizeof(uint32_t), w.bins + (size_t)query * w.bins_stride, sizeof(uint2), sizeof(uint32_t), (size_t)n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) out[i] = rq_rec_m1(raw[(size_t)i]);
    return n;
}

// ---- test hook: the int8 image's worst row error per bin (what the tail lifts its threshold by) -------------
extern "C" int64_t rq_debug_bin_err(rq_index* idx, float* out, int64_t max_bins) {
    if (!idx || !out || !idx->shards.empty()) return set_err(RQ_EINVAL, "bad arguments");
    RQ_ON_DEVICE(idx);
    if (!idx->x8 || !idx->binerr8 || idx->x8_valid != idx->n) return set_err(RQ_EINVAL, "the int8 image is not built (run a search with the int8 scan first)");
    const int64_t n = std::min((idx->n + 63) / 64, max_bins);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, idx->binerr8, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
}

// ---- timing ----------------------------------------------------------------------------------
extern "C" int rq_get_timing(rq_index* idx, rq_timing* out) {
    if (!idx || !out) return set_err(RQ_EINVAL, "null argument");
    if (!idx->shards.empty()) {   // kernel figures summed over the children, call counts of the parent
        rq_timing sum = idx->t;
        for (rq_index* c : idx->shards) {
            rq_timing t;
            if (int r = rq_get_timing(c, &t)) return r;
            sum.scan_ms += t.scan_ms; sum.scan_launches += t.scan_launches; sum.scan_bytes += t.scan_bytes;
            sum.widened += t.widened; sum.exact_scans += t.exact_scans;
        }
        *out = sum;
        return RQ_OK;
    }
    RQ_ON_DEVICE(idx);
    HIPCHK(hipDeviceSynchronize());
    double ms = 0.0;
    for (size_t i = 0; i < idx->ev_used; ++i) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, idx->events[i].first, idx->events[i].second));
        ms += t;
    }
    idx->t.scan_ms = ms;
    idx->t.scan_launches = (int64_t)idx->ev_used;
    idx->t.scan_bytes = idx->ev_bytes;
    *out = idx->t;
    return RQ_OK;
}
extern "C" int rq_reset_timing(rq_index* idx) {
    if (!idx) return set_err(RQ_EINVAL, "null argument");
    if (!idx->shards.empty()) {
        idx->t = rq_timing{};
        for (rq_index* c : idx->shards)
            if (int r = rq_reset_timing(c)) return r;
        return RQ_OK;
    }
    RQ_ON_DEVICE(idx);
    HIPCHK(hipDeviceSynchronize());
    idx->ev_used = 0; idx->ev_bytes = 0;
    idx->t = rq_timing{};
    return RQ_OK;
}

// ---- persistence -----------------------------------------------------------------------------
extern "C" int rq_save(const rq_index* idx, const char* path) {
    if (!idx || !path) return set_err(RQ_EINVAL, "null argument");
    RQ_ON_DEVICE(idx);
    const std::string meta = std::string(path) + ".meta", data = std::string(path) + ".f16";
    FILE* f = fopen(data.c_str(), "wb");
    if (!f) return set_err(RQ_EIO, "cannot write %s", data.c_str());
    const int64_t chunk = 65536;
    std::vector<uint16_t> buf((size_t)chunk * idx->dim);
    for (int64_t off = 0; off < idx->n; off += chunk) {
        const int64_t m = std::min(chunk, idx->n - off);
        if (int r = rq_index_get_rows_f16(idx, off, m, buf.data())) { fclose(f); return r; }
        if (fwrite(buf.data(), (size_t)idx->dim * 2, (size_t)m, f) != (size_t)m) { fclose(f); return set_err(RQ_EIO, "short write to %s", data.c_str()); }
    }
    fclose(f);
    f = fopen(meta.c_str(), "w");
    if (!f) return set_err(RQ_EIO, "cannot write %s", meta.c_str());
    fprintf(f, "rq-index 1\ndim %d\nrows %lld\ndtype f16\n", idx->dim, (long long)idx->n);
    fclose(f);
    return RQ_OK;
}

extern "C" rq_index* rq_load(const char* path, int n_devices, const int* device_ids) {
    if (!path) { set_err(RQ_EINVAL, "null path"); return nullptr; }
    const std::string meta = std::string(path) + ".meta", data = std::string(path) + ".f16";
    FILE* f = fopen(meta.c_str(), "r");
    if (!f) { set_err(RQ_EIO, "cannot read %s", meta.c_str()); return nullptr; }
    int ver = 0, dim = 0;
    long long rows = -1;
    char dtype[16] = "";
    const int got = fscanf(f, "rq-index %d dim %d rows %lld dtype %15s", &ver, &dim, &rows, dtype);
    fclose(f);
    if (got != 4 || ver != 1 || rows < 0 || strcmp(dtype, "f16") != 0) { set_err(RQ_EIO, "%s is not an rq-index v1 meta file", meta.c_str()); return nullptr; }
    rq_index* idx = rq_index_create(dim, n_devices, device_ids);
    if (!idx) return nullptr;
    f = fopen(data.c_str(), "rb");
    if (!f) { set_err(RQ_EIO, "cannot read %s", data.c_str()); rq_index_destroy(idx); return nullptr; }
    if (!idx->shards.empty()) {   // multi-device: stripes no longer than an even share, so that a small collection still uses every device
        const int64_t g = (int64_t)idx->shards.size(), share = ((rows + g - 1) / g + 63) / 64 * 64;
        idx->stripe = std::min<int64_t>(idx->stripe, std::max<int64_t>(share, 4096));
    }
    if (rows > 0 && rq_index_reserve(idx, rows) != RQ_OK) { fclose(f); rq_index_destroy(idx); return nullptr; }
    const int64_t chunk = 65536;
    std::vector<uint16_t> buf((size_t)chunk * dim);
    for (int64_t off = 0; off < rows; off += chunk) {
        const int64_t m = std::min<int64_t>(chunk, rows - off);
        if (fread(buf.data(), (size_t)dim * 2, (size_t)m, f) != (size_t)m) { fclose(f); set_err(RQ_EIO, "short read from %s", data.c_str()); rq_index_destroy(idx); return nullptr; }
        if (rq_index_add_f16(idx, buf.data(), m) != RQ_OK) { fclose(f); rq_index_destroy(idx); return nullptr; }
    }
    fclose(f);
    return idx;
}
