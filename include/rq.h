/* rq.h -- C ABI of the MI355X dense-retrieval backend (librq_hip.so).
 *
 * This is the drop-in boundary for the reference's dense path.  The reference has no native code;
 * the calls below replace the two third-party services its DenseIndex talks to:
 *
 *   reference rag_uq/streaming_index.py          replaced by
 *   ------------------------------------------   ---------------------------------------------
 *   :252-263  chromadb client + collection       rq_index_create / rq_load
 *   :306      collection.get()['ids'] (dedup)    host-side id table (Python), rq_index_size
 *   :326-331  collection.add(embeddings=...)     rq_index_add_f32 / rq_index_add_f16
 *   :355-359  collection.query(n_results=top_k)  rq_search (host buffers) / rq_search_device
 *   :363-368  score = 1 - distance, best first   scores returned as cosine similarity, best first
 *   :373      collection.count()                 rq_index_size
 *   (Chroma persistence directory)               rq_save / rq_load
 *
 * Conventions: plain pointers and sizes, no torch types.  Host buffers are C-contiguous and owned by
 * the caller; device memory is owned by the library.  Every int-returning call returns 0 on success
 * or a negative RQ_E* code; the message is in rq_last_error() (thread local).  Calls block unless
 * the name ends in _device.  One rq_index may be used from one thread at a time (the reference is
 * single-threaded, streaming_index.py has no locks).  There is no CPU fallback: without a gfx950
 * device every call that touches data fails with RQ_ENODEVICE.
 *
 * Score definition (pinned by oracle/dense_oracle.py, "parity unpinned" upstream -- Chroma's own
 * arithmetic is not in the reference tree):
 *   cosine:  s = fp32( dot64(q, x) / (||q||_64 * ||x||_64 + 1e-30) )   on the STORED fp16 row x
 *   ip:      s = fp32( dot64(q, x) )
 *   order:   s descending, then row ascending;  k_eff = min(k, N);  missing entries row = -1.
 */
#ifndef RQ_H
#define RQ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rq_index rq_index;

enum { RQ_OK = 0, RQ_EINVAL = -1, RQ_ENODEVICE = -2, RQ_EHIP = -3, RQ_ENOMEM = -4, RQ_EIO = -5, RQ_EUNSUPPORTED = -6 };
enum { RQ_METRIC_COSINE = 0, RQ_METRIC_IP = 1 };
#define RQ_MAX_DIM 768
#define RQ_MAX_K 1024

/* Number of visible HIP devices (0 if none / no driver). */
int rq_device_count(void);

/* Create an empty index of `dim`-element rows (1 <= dim <= RQ_MAX_DIM).  NULL on error.
 * n_devices == 1: one row shard on device_ids[0] (the form every *_device call works on; shards in other processes
 *   are merged with rq_merge_keys_device).
 * n_devices  > 1: the rows are sharded across device_ids[0..n_devices) inside the library: every appended block is cut
 *   into n_devices contiguous pieces, piece j is stored on device_ids[j]; rq_search enqueues the search on every device
 *   before it waits for any, copies each device's k best (score, row) pairs back and merges them on the host in the
 *   canonical order.  Same rq_index_add_f16 / _f32, rq_search, rq_index_get_rows_f16, rq_save / rq_load, rq_set_option
 *   calls; the device-pointer entry points (*_device) return RQ_EUNSUPPORTED on such an index.  A device may be named
 *   more than once (several shards on one GPU). */
rq_index* rq_index_create(int dim, int n_devices, const int* device_ids);
void rq_index_destroy(rq_index* idx);

int rq_index_dim(const rq_index* idx);
int64_t rq_index_size(const rq_index* idx);
/* Global id of this shard's row 0 (default 0); out_rows / keys carry row_offset + local row. */
int rq_index_set_row_offset(rq_index* idx, int64_t row_offset);
int rq_index_reserve(rq_index* idx, int64_t n_rows);

/* Append n rows (row-major [n][dim]).  _f16: IEEE binary16 bit patterns, stored as given.
 * _f32: rounded to fp16 (round-to-nearest-even); with normalize != 0 each row is first divided by
 * its fp64 L2 norm (cosine is invariant to it and fp16 range is never exceeded).
 * The fp64 norm of the STORED row is computed on device and kept for exact scoring. */
int rq_index_add_f16(rq_index* idx, const uint16_t* rows, int64_t n_rows);
int rq_index_add_f32(rq_index* idx, const float* rows, int64_t n_rows, int normalize);
/* Same, rows already in device memory of the index's GPU.  The call first waits for ALL prior work on
 * the device (the rows may come from any stream of the caller), then appends synchronously. */
int rq_index_add_f16_device(rq_index* idx, const void* d_rows, int64_t n_rows);
int rq_index_add_f32_device(rq_index* idx, const float* d_rows, int64_t n_rows, int normalize);
/* Copy stored rows [row_begin, row_begin+n) back to the host as fp16 bit patterns [n][dim]. */
int rq_index_get_rows_f16(const rq_index* idx, int64_t row_begin, int64_t n_rows, uint16_t* out);

/* Exact top-k of B queries ([B][dim] fp32, host).  out_scores [B][k], out_rows [B][k] (-1 padded).
 * Results are exact (see score definition): an on-device certificate proves it per query, and the
 * queries it cannot prove are re-run with a wider candidate set, finally with a full fp64 scan. */
int rq_search(rq_index* idx, const float* queries, int B, int k, int metric, float* out_scores, int64_t* out_rows);

/* Asynchronous form: all pointers are device pointers, work is enqueued on `stream` (a hipStream_t,
 * NULL = default stream).  d_keys ([B][k] uint64, may be NULL) receives sortable (score, global row)
 * keys for cross-shard merging.  d_status[B]: 0 = certified exact, 1 = not certified -- call
 * rq_search_fixup_device after the stream has the results you need repaired.
 * Calls on different streams may overlap; each stream has its own workspace. */
int rq_search_device(rq_index* idx, const float* d_queries, int B, int k, int metric, float* d_scores, int64_t* d_rows,
                     uint64_t* d_keys, int* d_status, void* stream);
/* Deferred tails (option "pipeline"): the tail of a search (threshold, fp64 re-score, final top-k) is taken off
 * the critical path of `stream` so that the next corpus scan starts at once.
 *   1: the tail runs on an internal stream (two cross-stream events per call);
 *   2: the tail of call i is executed by extra workgroups of the scan launch of call i+1 on the same stream (no
 *      events, scans never overlap each other; calls of <= 64 queries, k <= 128).
 * In both modes the outputs of rq_search_device calls are complete on `stream` only after this call
 * (rq_search_fixup_device implies it); output buffers must stay valid until then.  Mode 1 also reads d_queries
 * until then, mode 2 keeps its own copy. */
int rq_search_flush_device(rq_index* idx, void* stream);
/* "pipeline" = 2 loops that know their next batch: announce the queries of the NEXT rq_search_device call on `stream`
 * before making the current one.  The current call's launch then prepares them (norms, unit-norm fp16 fragments) with 64
 * extra workgroups, and the next call -- when it is given exactly d_next_queries and B -- needs no preparation launch of its
 * own (measured: one 5 us kernel + a launch boundary per batch).  d_next_queries must hold its final contents when the
 * current call is made and stay unchanged until the call that searches it.  Advisory: a hint that cannot be used (more than
 * 64 queries, another pipeline mode, a different pointer / B at the next call, a flush in between) is dropped and the
 * next call prepares its queries itself; results never depend on it.  B = 0 or NULL withdraws a pending hint. */
int rq_search_hint_next_device(rq_index* idx, const float* d_next_queries, int B, void* stream);
/* A TRAIN of searches enqueued by one call (serving loops, the multi-GPU bench: a 125 k-row shard answers a 64-query batch in
 * ~25 us, which a host loop that crosses the language boundary twice per batch cannot feed): batch i = B queries at
 * d_queries[i], searched exactly as rq_search_device(idx, d_queries[i], B, k, metric, d_scores[i], d_rows[i],
 * d_keys ? d_keys[i] : NULL, d_status[i], streams[i % n_streams]) would, after announcing d_queries[i + n_streams] -- the
 * batch the same stream searches next -- with rq_search_hint_next_device.  The pointer tables are HOST arrays of device
 * pointers; d_queries holds n_batches + n_streams entries, the last n_streams of which are only announced (the first
 * batches of the NEXT train; NULL = nothing to announce).  Results are complete per stream after rq_search_flush_device,
 * as for single calls.  Replaces, n_batches times over, the collection.query of reference rag_uq/streaming_index.py:355-359. */
int rq_search_train_device(rq_index* idx, int n_batches, const float* const* d_queries, int B, int k, int metric,
                           float* const* d_scores, int64_t* const* d_rows, uint64_t* const* d_keys, int* const* d_status,
                           void* const* streams, int n_streams);
/* The library keeps one search workspace per caller stream (about 11 MB at 1M rows and 64 queries).  Call this before a
 * stream that has been used for searches is destroyed, or when it will not be used again: waits for the device, runs
 * any deferred tail, frees the stream's workspace.  (Beyond 8 streams the library drops idle workspaces by itself.) */
int rq_stream_release(rq_index* idx, void* stream);
/* Synchronises `stream`, re-runs the queries whose d_status is non-zero with wider candidate sets /
 * the exact scan, patches d_scores/d_rows/d_keys/d_status in place.  Returns the number repaired. */
int rq_search_fixup_device(rq_index* idx, const float* d_queries, int B, int k, int metric, float* d_scores, int64_t* d_rows,
                           uint64_t* d_keys, int* d_status, void* stream);

/* Merge per-shard key lists: d_keys_in [B][n_per_query] (any order, 0 = empty) -> best k per query.
 * d_rows receives the global row ids carried by the keys.  d_keys_out may be NULL. */
int rq_merge_keys_device(const uint64_t* d_keys_in, int n_per_query, int B, int k, float* d_scores, int64_t* d_rows,
                         uint64_t* d_keys_out, void* stream);

/* Tuning / test hooks: "kstage" (1: an LDS stage holds whole rows, 2: half rows), "ring" (LDS stages 2..6; the
 * (kstage, ring, prefetch) triples built are listed in csrc/rq_scan.hip, others fail with RQ_EHIP at search time),
 * "wg_per_cu", "nt" (non-temporal corpus loads: 0, 1, -1 = auto), "slack_bins" (extra bins beyond k, -1 = auto),
 * "eps" (certificate bound, <0 = derived default), "profile" (HIP events on every scan launch, attached to the dispatch; "profile_stride" n: on every n-th; "profile_legacy" 1: hipEventRecord around it),
 * "prefetch" (LDS fragments read ahead of their MFMAs: 1, 4, 6, 12), "fast_tail" (0 = generic sorted tail),
 * "pipeline" (see rq_search_flush_device), "wide_batch" (calls of more than 64 queries: 0 = passes of 64 only, 1 = passes of
 * 256 / 128 / 64, 3 = 128 / 64, 2 = round 1's 8-wave 128-query pass), "wide128" / "wide256" (variant of csrc/rq_scan_wide.hip),
 * "epi" (selection form of the 64-query scan: 1 = row positions inside the scores, 0 = compare / select),
 * "use_hint" (0: rq_search_hint_next_device is ignored),
 * "scan8" (searches may scan an int8 image of the shard instead of its fp16 rows -- half the bytes per pass;
 *   candidates are still re-scored from the fp16 rows in fp64, so results do not change: 0 = never, 1 = for k <= 128 on shards of
 *   200 000 rows and more (default), 2 = always.  The image (+768 B per row) is built by the first search that wants it (no room for it: the fp16 rows stay the operand); a shard whose
 *   worst row quantises with more than 3 % relative error keeps the fp16 scan),
 * "scan8_split" (-1 = default: the queries reach the int8 scan as one int8 image for k <= 32 and as two -- value and residual,
 *   twice the matrix-core work, a third of the candidate rows -- for larger k; 0 / 1: one / two images for every k),
 * "wide8" (default 1: calls of more than 64 queries whose k class runs with one int8 image use 128-query passes over the image;
 *   0: the fp16 passes of "wide_batch"),
 * "thr_mult8" (1.05 .. 2.25, default 1.25: candidate threshold of the int8 scan, T = P - bound - (thr_mult8 - 1) * max(bound,
 *   typical one-image bound); 2.25 certifies by construction, smaller values re-score fewer rows and leave the rare query
 *   whose errors add up to the repair path of rq_search_fixup_device.  When more than 1 in 16 checked queries of a class of
 *   k (<= 32 / larger) needed repair, that class moves one step along one image -> two images -> fp16 scan, until "scan8" or
 *   "scan8_split" is set again),
 * "wide256_8" (default 31: calls of more than 128 queries on the one-image class use 256-query passes over the image, a variant of
 *   csrc/rq_scan_wide.hip; 0 = passes of 128 as in round 2), "bin_bound" (A/B hook, default 1: the tail tests every bin with its own rows'
 *   worst quantisation error instead of the shard's), "stripe_rows" (multi-device index, before the first append: rows per stripe),
 * "tail_local" (A/B hook, default 1: a tail workgroup with more than k re-scored rows publishes only its own k best keys),
 * "poison_cand" (test hook: candidate lists are filled with 0xff..ff keys before every tail).
 * "scan8" = 1 measures where the ladder STARTS when the image is built (64 stored rows searched as queries through every rung, the
 *   fastest rung that certifies wins; "scan8_calibrated_rows", "scan8_calib_ms_<class><rung>", "scan8_calib_unc_<class><rung>" report it).
 * Read-only: "repaired_queries" (queries that came back uncertified and were repaired, all rungs of the ladder), "scan8_used" (searches that scanned the int8 image), "scan8_row_err" (worst row's relative int8 error, -1 = image
 * not built), "scan8_level" (ladder position: class k <= 32 + 10 * class of larger k; 0 one image, 1 two images, 2 fp16 scan),
 * "scan8_wide_one_image" (same encoding: 1 = calls of more than 64 queries of that class scan ONE int8 image per query -- also in a class whose 64-query calls
 *   run on two images, when the image-build measurement found one image eligible; given up for the wide calls alone after too many repairs),
 * "scan8_suspended" (bit 0 / 1: that class is back at the fp16 scan), "hints_used" (searches that found their queries prepared, see rq_search_hint_next_device), "max_row_norm", "max_sub_rel" / "max_sub_abs" (largest share of a stored row that sits in fp16-subnormal elements,
 * which the matrix cores flush), "eps_cosine" / "eps_ip" (the certificate's bound including that term). */
int rq_set_option(rq_index* idx, const char* name, double value);
double rq_get_option(const rq_index* idx, const char* name);

typedef struct rq_timing {
    double scan_ms;          /* sum of HIP-event durations of the scan kernel launches recorded */
    int64_t scan_launches;   /* launches recorded (profile on) */
    int64_t scan_bytes;      /* algorithmic corpus bytes of those launches: rows * 2 * dim_padded each */
    int64_t searches;        /* rq_search* calls */
    int64_t queries;         /* queries searched */
    int64_t widened;         /* queries re-run with a wider candidate set */
    int64_t exact_scans;     /* queries that needed the full fp64 scan */
} rq_timing;
int rq_get_timing(rq_index* idx, rq_timing* out);   /* synchronises the device */
int rq_reset_timing(rq_index* idx);

/* Flat-file persistence: <path>.meta (text), <path>.f16 (rows [N][dim] fp16). */
int rq_save(const rq_index* idx, const char* path);
rq_index* rq_load(const char* path, int n_devices, const int* device_ids);

/* Test hook: copy the scan's approximate per-bin maxima of query `query` of the LAST search enqueued on `stream`
 * (bin b = rows 64 b .. 64 b + 63) to the host.  Returns the number of bins copied. */
int64_t rq_debug_pooled(rq_index* idx, void* stream, int query, float* out, int64_t max_bins);
/* Test hook: the int8 image's worst relative row error of every bin (fp32, rounded up), i.e. what the tail may lift its candidate
 * threshold by per bin (option "bin_bound").  Fails unless the image is built and up to date.  Returns the number of bins copied. */
int64_t rq_debug_bin_err(rq_index* idx, float* out, int64_t max_bins);

/* Measurement hook: GB/s of a plain streaming read (16-byte loads, nothing else) of the stored shard, averaged over
 * `iters` back-to-back passes -- what THIS GPU delivers right now, the yardstick for the scan kernel's rate.
 * nt: non-temporal loads 0 / 1 / -1 = the scan's own rule.  Negative on error. */
double rq_debug_read_bandwidth(rq_index* idx, int iters, int nt, int wg_per_cu);

/* Development hook: enable != 0 makes every workgroup of the fused scan + tail launches (option "pipeline" = 2) record
 * {start, end} wall-clock ticks (100 MHz), HW_ID and XCC_ID; out (may be NULL) receives [max_wgs][4] words of the last
 * such launch.  enable == 0 switches it off again.  Used by tools/gpu_stamps.py. */
int rq_debug_stamps(rq_index* idx, int enable, unsigned long long* out, int max_wgs);

const char* rq_last_error(void);
const char* rq_version(void);

/* ---- encoder pieces (the embedding the reference asks Ollama for, rag_uq/streaming_index.py:267-288) ----
 * The memory-bound parts of the NomicBert (nomic-embed-text) forward pass as fused gfx950 kernels; the GEMMs between them stay
 * with the framework's library (hipBLASLt).  All pointers are device pointers, fp16 unless noted; `stream` is a hipStream_t.
 * Used by embedders.NomicBertEmbedder (efficient-rag-..._amd/embedders.py), csrc/rq_encoder.hip. */
/* d_rope[pos][0..31] = cos(pos * rope_theta^(-d / 32)), d_rope[pos][32..63] = the sines (fp32 [seq][64]): the rotary table. */
int rq_nb_rope_table_f32(float* d_rope, int seq, float rope_theta, void* stream);
/* ctx[b][t][:] = softmax(rot(q) rot(k)^T / 8 + prefix mask) v per head.  d_qkv [batch * seq][3 * heads * 64] (q | k | v of the
 * fused projection), d_len[batch] = valid tokens of each sequence (the first d_len[b] positions; padded rows come out zero),
 * rotary = rotate-half over the 64 dims with d_rope (at least seq rows).  seq <= 512 (RQ_EUNSUPPORTED beyond: use the
 * framework's attention). */
int rq_nb_attention_f16(const void* d_qkv, const int* d_len, const float* d_rope, void* d_ctx, int batch, int seq, int heads, void* stream);
/* The same for a PACKED batch -- no padding rows: the tokens of sequence b are rows [d_offsets[b], d_offsets[b + 1]) of d_qkv and d_ctx
 * (d_offsets: batch + 1 ascending ints, d_offsets[0] = 0), every sequence at most max_seq <= 512 tokens long.  The GEMMs around the
 * attention then run on sum(lengths) rows instead of batch * longest. */
int rq_nb_attention_packed_f16(const void* d_qkv, const int* d_offsets, const float* d_rope, void* d_ctx, int batch, int max_seq, int heads, void* stream);
/* out = LayerNorm(x + res) * gamma + beta over rows of `width` (8..1536, multiple of 8) elements, fp32 statistics; res may be
 * NULL; out may alias x or res. */
int rq_nb_add_layernorm_f16(const void* d_x, const void* d_res, const void* d_gamma, const void* d_beta, void* d_out, int64_t rows, int width,
                            float eps, void* stream);
/* out[t][j] = silu(gate_up[t][j]) * gate_up[t][inter + j]: d_gate_up [rows][2 * inter] (gate | up), d_out [rows][inter]. */
int rq_nb_swiglu_f16(const void* d_gate_up, void* d_out, int64_t rows, int inter, void* stream);
/* d_out[b][:] (fp32) = mean of d_h[b][t][:] over the first d_len[b] tokens (0 for an empty sequence). */
int rq_nb_mean_pool_f16(const void* d_h, const int* d_len, float* d_out, int batch, int seq, int width, void* stream);
/* ... of a packed batch: d_out[b][:] = mean of the rows [d_offsets[b], d_offsets[b + 1]) of d_h. */
int rq_nb_mean_pool_packed_f16(const void* d_h, const int* d_offsets, float* d_out, int batch, int max_seq, int width, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RQ_H */
