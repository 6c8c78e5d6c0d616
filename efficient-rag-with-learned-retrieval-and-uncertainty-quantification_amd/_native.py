"""ctypes binding of librq_hip.so (C ABI declared in include/rq.h).

This is the only place Python touches the native library.  There is no CPU fallback: if the shared
object is missing or no gfx950 device is visible, construction of an index raises.

Replaces the chromadb client calls of reference rag_uq/streaming_index.py:252-263,326-331,355-359,373.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence, Tuple

import numpy as np

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["RQ_LIB_PATH"]) if os.environ.get("RQ_LIB_PATH") else _PKG_DIR / "librq_hip.so"   # override: A/B of builds

METRIC_COSINE = 0
METRIC_IP = 1
MAX_DIM = 768
MAX_K = 1024


class RqError(RuntimeError):
    """A call into librq_hip.so failed; the message is rq_last_error()."""


class rq_timing(C.Structure):
    _fields_ = [
        ("scan_ms", C.c_double),
        ("scan_launches", C.c_int64),
        ("scan_bytes", C.c_int64),
        ("searches", C.c_int64),
        ("queries", C.c_int64),
        ("widened", C.c_int64),
        ("exact_scans", C.c_int64),
    ]


# name -> (restype, argtypes); must list every symbol include/rq.h declares (tests check this)
_SIGNATURES = {
    "rq_device_count": (C.c_int, []),
    "rq_index_create": (C.c_void_p, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rq_index_destroy": (None, [C.c_void_p]),
    "rq_index_dim": (C.c_int, [C.c_void_p]),
    "rq_index_size": (C.c_int64, [C.c_void_p]),
    "rq_index_set_row_offset": (C.c_int, [C.c_void_p, C.c_int64]),
    "rq_index_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "rq_index_add_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "rq_index_add_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "rq_index_add_f16_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "rq_index_add_f32_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "rq_index_get_rows_f16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "rq_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rq_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "rq_search_fixup_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "rq_search_flush_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rq_search_train_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int]),
    "rq_nb_rope_table_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "rq_nb_attention_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rq_nb_add_layernorm_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "rq_nb_swiglu_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "rq_nb_mean_pool_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rq_nb_attention_packed_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rq_nb_mean_pool_packed_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rq_search_hint_next_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "rq_stream_release": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rq_merge_keys_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "rq_debug_pooled": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "rq_debug_bin_err": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]),
    "rq_debug_read_bandwidth": (C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "rq_debug_stamps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rq_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "rq_get_option": (C.c_double, [C.c_void_p, C.c_char_p]),
    "rq_get_timing": (C.c_int, [C.c_void_p, C.POINTER(rq_timing)]),
    "rq_reset_timing": (C.c_int, [C.c_void_p]),
    "rq_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rq_load": (C.c_void_p, [C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "rq_last_error": (C.c_char_p, []),
    "rq_version": (C.c_char_p, []),
}

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen librq_hip.so and declare every entry point.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).
    # Whichever is loaded first serves both, a second copy cannot open the GPU -- so load torch's first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not LIB_PATH.exists():
        raise RqError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {(_PKG_DIR / 'csrc')}`; this backend has no CPU fallback"
        )
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_LOCAL", 0))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load_library().rq_last_error().decode("utf-8", "replace")


def _check(rc: int, what: str) -> int:
    if rc < 0:
        raise RqError(f"{what}: {last_error()} (code {rc})")
    return rc


def device_count() -> int:
    return int(load_library().rq_device_count())


def _ptr(a) -> C.c_void_p:
    """Pointer of a numpy array, a torch tensor (host or device) or a raw integer address."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"cannot take the address of {type(a)!r}")


class NativeIndex:
    """Thin owner of one rq_index handle: one row shard on one GPU, or -- `devices=[...]` with more than one entry -- rows
    sharded across several GPUs inside the library (host-buffer calls only, see include/rq.h rq_index_create)."""

    def __init__(self, dim: int, device: int = 0, _handle: Optional[int] = None, devices: Optional[Sequence[int]] = None):
        self._lib = load_library()
        devs = [int(d) for d in devices] if devices else [int(device)]
        if _handle is None:
            ids = (C.c_int * len(devs))(*devs)
            _handle = self._lib.rq_index_create(int(dim), len(devs), ids)
            if not _handle:
                raise RqError(f"rq_index_create(dim={dim}, devices={devs}): {last_error()}")
        self._h = C.c_void_p(_handle)
        self.dim = int(self._lib.rq_index_dim(self._h))
        self.device = devs[0]
        self.devices = devs

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rq_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.rq_index_size(self._h))

    # -- build ------------------------------------------------------------------------------
    def reserve(self, n_rows: int) -> None:
        _check(self._lib.rq_index_reserve(self._h, int(n_rows)), "rq_index_reserve")

    def set_row_offset(self, off: int) -> None:
        _check(self._lib.rq_index_set_row_offset(self._h, int(off)), "rq_index_set_row_offset")

    def add_f16(self, rows: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows)
        if rows.dtype == np.float16:
            rows = rows.view(np.uint16)
        if rows.dtype != np.uint16 or rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"expected [n][{self.dim}] float16/uint16 rows, got {rows.dtype} {rows.shape}")
        _check(self._lib.rq_index_add_f16(self._h, _ptr(rows), rows.shape[0]), "rq_index_add_f16")

    def add_f32(self, rows: np.ndarray, normalize: bool = True) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"expected [n][{self.dim}] float32 rows, got {rows.shape}")
        _check(self._lib.rq_index_add_f32(self._h, _ptr(rows), rows.shape[0], int(bool(normalize))), "rq_index_add_f32")

    def add_f16_device(self, d_rows, n_rows: int) -> None:
        _check(self._lib.rq_index_add_f16_device(self._h, _ptr(d_rows), int(n_rows)), "rq_index_add_f16_device")

    def add_f32_device(self, d_rows, n_rows: int, normalize: bool = True) -> None:
        _check(self._lib.rq_index_add_f32_device(self._h, _ptr(d_rows), int(n_rows), int(bool(normalize))),
               "rq_index_add_f32_device")

    def get_rows_f16(self, row_begin: int, n_rows: int) -> np.ndarray:
        out = np.empty((int(n_rows), self.dim), dtype=np.uint16)
        _check(self._lib.rq_index_get_rows_f16(self._h, int(row_begin), int(n_rows), _ptr(out)), "rq_index_get_rows_f16")
        return out.view(np.float16)

    # -- search -----------------------------------------------------------------------------
    def search(self, queries: np.ndarray, k: int, metric: int = METRIC_COSINE) -> Tuple[np.ndarray, np.ndarray]:
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [B][{self.dim}] float32 queries, got {q.shape}")
        B = q.shape[0]
        scores = np.empty((B, int(k)), dtype=np.float32)
        rows = np.empty((B, int(k)), dtype=np.int64)
        _check(self._lib.rq_search(self._h, _ptr(q), B, int(k), int(metric), _ptr(scores), _ptr(rows)), "rq_search")
        return scores, rows

    def search_device(self, d_queries, B: int, k: int, metric: int, d_scores, d_rows, d_keys, d_status, stream: int = 0) -> None:
        _check(self._lib.rq_search_device(self._h, _ptr(d_queries), int(B), int(k), int(metric), _ptr(d_scores), _ptr(d_rows),
                                          _ptr(d_keys), _ptr(d_status), C.c_void_p(stream)), "rq_search_device")

    def search_fixup_device(self, d_queries, B: int, k: int, metric: int, d_scores, d_rows, d_keys, d_status, stream: int = 0) -> int:
        return _check(self._lib.rq_search_fixup_device(self._h, _ptr(d_queries), int(B), int(k), int(metric), _ptr(d_scores),
                                                       _ptr(d_rows), _ptr(d_keys), _ptr(d_status), C.c_void_p(stream)),
                      "rq_search_fixup_device")

    def search_hint_next_device(self, d_next_queries, B: int, stream: int = 0) -> None:
        """Announce the queries of the NEXT search_device call on `stream` (include/rq.h: rq_search_hint_next_device)."""
        _check(self._lib.rq_search_hint_next_device(self._h, _ptr(d_next_queries), int(B), C.c_void_p(stream)), "rq_search_hint_next_device")

    @staticmethod
    def make_train(d_queries: Sequence, d_scores: Sequence, d_rows: Sequence, d_keys: Optional[Sequence], d_status: Sequence,
                   streams: Sequence[int], d_announce: Optional[Sequence] = None) -> dict:
        """Pointer tables of a train of searches (include/rq.h rq_search_train_device), built once and re-used: batch i = d_queries[i]
        -> d_scores[i] / d_rows[i] / d_keys[i] / d_status[i] on streams[i % len(streams)]; d_announce = the first len(streams)
        batches of the train that follows (announced to the last launches of this one), or None."""
        n, ns = len(d_queries), len(streams)
        ann = list(d_announce) if d_announce is not None else []
        ann = (ann + [None] * ns)[:ns]
        addr = lambda a: 0 if a is None else (_ptr(a).value or 0)
        tab = lambda seq: (C.c_void_p * len(seq))(*[addr(a) for a in seq])
        return {"n": n, "ns": ns, "q": tab(list(d_queries) + ann), "scores": tab(d_scores), "rows": tab(d_rows),
                "keys": tab(d_keys) if d_keys is not None else None, "status": tab(d_status), "streams": (C.c_void_p * ns)(*[int(s) for s in streams]),
                "keep": (d_queries, d_scores, d_rows, d_keys, d_status, ann)}

    def search_train_device(self, train: dict, B: int, k: int, metric: int = METRIC_COSINE) -> None:
        _check(self._lib.rq_search_train_device(self._h, train["n"], train["q"], int(B), int(k), int(metric), train["scores"], train["rows"],
                                                train["keys"], train["status"], train["streams"], train["ns"]), "rq_search_train_device")

    def search_flush_device(self, stream: int = 0) -> None:
        _check(self._lib.rq_search_flush_device(self._h, C.c_void_p(stream)), "rq_search_flush_device")

    def stream_release(self, stream: int = 0) -> None:
        """Drop the search workspace kept for `stream` (call before destroying a stream that was used for searches)."""
        _check(self._lib.rq_stream_release(self._h, C.c_void_p(stream)), "rq_stream_release")

    def debug_pooled(self, query: int, max_bins: int, stream: int = 0) -> np.ndarray:
        out = np.empty((int(max_bins),), dtype=np.float32)
        n = _check(self._lib.rq_debug_pooled(self._h, C.c_void_p(stream), int(query), _ptr(out), int(max_bins)), "rq_debug_pooled")
        return out[:n]

    def debug_bin_err(self, max_bins: int) -> np.ndarray:
        out = np.empty((int(max_bins),), dtype=np.float32)
        n = _check(self._lib.rq_debug_bin_err(self._h, _ptr(out), int(max_bins)), "rq_debug_bin_err")
        return out[:n]

    def read_bandwidth(self, iters: int = 20, nt: int = -1, wg_per_cu: int = 8) -> float:
        """GB/s of a plain streaming read of the stored shard (measurement yardstick, see include/rq.h)."""
        v = float(self._lib.rq_debug_read_bandwidth(self._h, int(iters), int(nt), int(wg_per_cu)))
        if v < 0:
            raise RqError(f"rq_debug_read_bandwidth: {last_error()}")
        return v

    # -- knobs / timing -----------------------------------------------------------------------
    def set_option(self, name: str, value: float) -> None:
        _check(self._lib.rq_set_option(self._h, name.encode(), float(value)), f"rq_set_option({name})")

    def get_option(self, name: str) -> float:
        return float(self._lib.rq_get_option(self._h, name.encode()))

    def timing(self) -> dict:
        t = rq_timing()
        _check(self._lib.rq_get_timing(self._h, C.byref(t)), "rq_get_timing")
        return {f: getattr(t, f) for f, _ in rq_timing._fields_}

    def reset_timing(self) -> None:
        _check(self._lib.rq_reset_timing(self._h), "rq_reset_timing")

    # -- persistence ------------------------------------------------------------------------
    def save(self, path: str) -> None:
        _check(self._lib.rq_save(self._h, str(path).encode()), "rq_save")

    @classmethod
    def load(cls, path: str, device: int = 0, devices: Optional[Sequence[int]] = None) -> "NativeIndex":
        lib = load_library()
        devs = [int(d) for d in devices] if devices else [int(device)]
        ids = (C.c_int * len(devs))(*devs)
        h = lib.rq_load(str(path).encode(), len(devs), ids)
        if not h:
            raise RqError(f"rq_load({path}): {last_error()}")
        return cls(0, devs[0], _handle=h, devices=devs)


# -- encoder pieces (include/rq.h "encoder pieces", csrc/rq_encoder.hip): torch tensors or raw device addresses -------------
def nb_rope_table(rope, seq: int, rope_theta: float, stream: int = 0) -> None:
    _check(load_library().rq_nb_rope_table_f32(_ptr(rope), int(seq), float(rope_theta), C.c_void_p(stream)), "rq_nb_rope_table_f32")


def nb_attention(qkv, lengths, rope, ctx, batch: int, seq: int, heads: int, stream: int = 0) -> None:
    _check(load_library().rq_nb_attention_f16(_ptr(qkv), _ptr(lengths), _ptr(rope), _ptr(ctx), int(batch), int(seq), int(heads),
                                              C.c_void_p(stream)), "rq_nb_attention_f16")


def nb_attention_packed(qkv, offsets, rope, ctx, batch: int, max_seq: int, heads: int, stream: int = 0) -> None:
    _check(load_library().rq_nb_attention_packed_f16(_ptr(qkv), _ptr(offsets), _ptr(rope), _ptr(ctx), int(batch), int(max_seq), int(heads),
                                                     C.c_void_p(stream)), "rq_nb_attention_packed_f16")


def nb_add_layernorm(x, res, gamma, beta, out, rows: int, width: int, eps: float, stream: int = 0) -> None:
    _check(load_library().rq_nb_add_layernorm_f16(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(out), int(rows), int(width), float(eps),
                                                  C.c_void_p(stream)), "rq_nb_add_layernorm_f16")


def nb_swiglu(gate_up, out, rows: int, inter: int, stream: int = 0) -> None:
    _check(load_library().rq_nb_swiglu_f16(_ptr(gate_up), _ptr(out), int(rows), int(inter), C.c_void_p(stream)), "rq_nb_swiglu_f16")


def nb_mean_pool(h, lengths, out, batch: int, seq: int, width: int, stream: int = 0) -> None:
    _check(load_library().rq_nb_mean_pool_f16(_ptr(h), _ptr(lengths), _ptr(out), int(batch), int(seq), int(width), C.c_void_p(stream)),
           "rq_nb_mean_pool_f16")


def nb_mean_pool_packed(h, offsets, out, batch: int, max_seq: int, width: int, stream: int = 0) -> None:
    _check(load_library().rq_nb_mean_pool_packed_f16(_ptr(h), _ptr(offsets), _ptr(out), int(batch), int(max_seq), int(width), C.c_void_p(stream)),
           "rq_nb_mean_pool_packed_f16")


def merge_keys_device(d_keys_in, n_per_query: int, B: int, k: int, d_scores, d_rows, d_keys_out=None, stream: int = 0) -> None:
    lib = load_library()
    _check(lib.rq_merge_keys_device(_ptr(d_keys_in), int(n_per_query), int(B), int(k), _ptr(d_scores), _ptr(d_rows),
                                    _ptr(d_keys_out), C.c_void_p(stream)), "rq_merge_keys_device")


# -- librq_bm25.so (include/rq_bm25.h, csrc/rq_bm25.cpp): batched CPU BM25 for BM25Index.search_batch -- host cores, no GPU code ----------
BM25_LIB_PATH = _PKG_DIR / "librq_bm25.so"
_BM25_SIGNATURES = {
    "rq_bm25_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "rq_bm25_destroy": (None, [C.c_void_p]),
    "rq_bm25_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "rq_bm25_version": (C.c_char_p, []),
}
_bm25_lib: Optional[C.CDLL] = None


def load_bm25_library() -> C.CDLL:
    global _bm25_lib
    if _bm25_lib is None:
        if not BM25_LIB_PATH.exists():
            raise RqError(f"{BM25_LIB_PATH} is missing: build it with `make -C {(_PKG_DIR / 'csrc')}`")
        lib = C.CDLL(str(BM25_LIB_PATH))
        for name, (res, args) in _BM25_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bm25_lib = lib
    return _bm25_lib


def bm25_available() -> bool:
    try:
        load_bm25_library()
        return True
    except (RqError, OSError):
        return False


def bm25_create(indptr: np.ndarray, rows: np.ndarray, contrib: np.ndarray, n_tokens: int, n_docs: int) -> int:
    """Handle over BORROWED arrays (the caller keeps them alive until bm25_destroy)."""
    assert indptr.dtype == np.int64 and rows.dtype == np.int32 and contrib.dtype == np.float64
    assert indptr.flags.c_contiguous and rows.flags.c_contiguous and contrib.flags.c_contiguous and len(indptr) == n_tokens + 1
    h = load_bm25_library().rq_bm25_create(_ptr(indptr), _ptr(rows), _ptr(contrib), int(n_tokens), int(n_docs))
    if not h:
        raise RqError("rq_bm25_create: malformed posting arrays")
    return h


def bm25_destroy(handle: int) -> None:
    if handle and _bm25_lib is not None:
        _bm25_lib.rq_bm25_destroy(C.c_void_p(handle))


def bm25_topk(handle: int, q_indptr: np.ndarray, q_tokens: np.ndarray, n_queries: int, k: int, n_threads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    out_rows = np.empty((n_queries, k), np.int32)
    out_scores = np.empty((n_queries, k), np.float64)
    q_indptr = np.ascontiguousarray(q_indptr, np.int64)
    q_tokens = np.ascontiguousarray(q_tokens, np.int32)
    rc = load_bm25_library().rq_bm25_topk(C.c_void_p(handle), _ptr(q_indptr), _ptr(q_tokens) if len(q_tokens) else None, int(n_queries), int(k),
                                          _ptr(out_rows), _ptr(out_scores), int(n_threads))
    if rc != 0:
        raise RqError(f"rq_bm25_topk failed (code {rc})")
    return out_rows, out_scores
