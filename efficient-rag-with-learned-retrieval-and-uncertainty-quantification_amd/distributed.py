"""Row-sharded search across GPUs: one process per GPU, one exchange step.

The reference has no multi-device code at all; this is the north star's addition.  Rank r holds a
contiguous block of corpus rows (global id = row_offset + local row).  Every rank searches every
query batch on its shard, then the per-shard top-k lists travel as packed 64-bit keys

    key = mono32(score) << 32 | (0xffffffff - global_row)        (0 = empty slot)

so that "larger key" == "better rank" under the canonical order (score desc, row asc).  One
`all_gather` (RCCL over xGMI when the backend is nccl; gloo on CPU in the tests) moves B*k*8 bytes
per rank; the merge is a top-k over world*k keys per query -- on the GPU (`rq_merge_keys_device`) or
on the host (`merge_keys_host`).  No all-reduce, no all-to-all.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from . import _native

_SIGN = np.uint32(0x80000000)


def mono32(scores: np.ndarray) -> np.ndarray:
    """Order-preserving float32 -> uint32 map (csrc/rq_device.h rq_mono32)."""
    u = np.ascontiguousarray(scores, dtype=np.float32).view(np.uint32)
    return np.where(u & _SIGN, ~u, u | _SIGN).astype(np.uint32)


def unmono32(k: np.ndarray) -> np.ndarray:
    k = np.ascontiguousarray(k, dtype=np.uint32)
    u = np.where(k & _SIGN, k & np.uint32(0x7FFFFFFF), ~k).astype(np.uint32)
    return u.view(np.float32)


def pack_keys(scores: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """(fp32 scores, int64 global rows, -1 = empty) -> uint64 keys."""
    rows = np.asarray(rows, dtype=np.int64)
    valid = rows >= 0
    if valid.any() and int(rows[valid].max()) >= 0xFFFFFFFF:
        raise ValueError("global row ids must be below 2^32 - 1")
    inv = (np.uint64(0xFFFFFFFF) - np.where(valid, rows, 0).astype(np.uint64))
    keys = (mono32(scores).astype(np.uint64) << np.uint64(32)) | inv
    return np.where(valid, keys, np.uint64(0))


def unpack_keys(keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    keys = np.asarray(keys, dtype=np.uint64)
    scores = unmono32((keys >> np.uint64(32)).astype(np.uint32)).copy()
    rows = (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
    empty = keys == 0
    scores[empty] = 0.0
    rows[empty] = -1
    return scores, rows


def merge_keys_host(keys: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """keys [B][n] (any order) -> canonical top-k (scores [B][k], rows [B][k], -1 padded)."""
    keys = np.asarray(keys, dtype=np.uint64)
    B, n = keys.shape
    top = np.zeros((B, k), dtype=np.uint64)
    srt = np.sort(keys, axis=1)[:, ::-1]
    m = min(k, n)
    top[:, :m] = srt[:, :m]
    return unpack_keys(top)


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: [lo, hi)."""
    per = (n_rows + world - 1) // world
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


class ShardedDenseSearcher:
    """One rank's view of a row-sharded index.  `index` is this rank's NativeIndex (row_offset set)."""

    def __init__(self, index: _native.NativeIndex, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.index = index
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", index.device)

    def search_device(self, d_queries, k: int, metric: int = _native.METRIC_COSINE, stream=None):
        """queries: float32 CUDA tensor [B][dim].  Returns merged (scores [B][k], global rows [B][k]) CUDA tensors.
        Blocks only on the certificate check of this rank's shard (a [B] int32 read-back)."""
        torch, dist = self.torch, self.dist
        B = d_queries.shape[0]
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        scores = torch.empty((B, k), device=self.device)
        rows = torch.empty((B, k), device=self.device, dtype=torch.int64)
        keys = torch.empty((B, k), device=self.device, dtype=torch.int64)
        status = torch.empty((B,), device=self.device, dtype=torch.int32)
        with torch.cuda.stream(st):
            self.index.search_device(d_queries, B, k, metric, scores, rows, keys, status, st.cuda_stream)
            self.index.search_fixup_device(d_queries, B, k, metric, scores, rows, keys, status, st.cuda_stream)
            if self.world == 1:
                return scores, rows
            gathered = torch.empty((self.world, B, k), device=self.device, dtype=torch.int64)
            dist.all_gather_into_tensor(gathered, keys, group=self.group)
            merged_in = gathered.permute(1, 0, 2).contiguous()
            _native.merge_keys_device(merged_in, self.world * k, B, k, scores, rows, None, st.cuda_stream)
        return scores, rows


def gather_merge_host(local_scores: np.ndarray, local_rows: np.ndarray, k: int, group=None) -> Tuple[np.ndarray, np.ndarray]:
    """Host-side exchange + merge (any backend, gloo included): every rank contributes its local
    top-k (global row ids) and receives the global canonical top-k."""
    import torch
    import torch.distributed as dist

    keys = pack_keys(local_scores, local_rows)
    world = dist.get_world_size(group)
    t = torch.from_numpy(keys.view(np.int64).copy())
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    allk = np.concatenate([o.numpy().view(np.uint64) for o in out], axis=1)
    return merge_keys_host(allk, k)


class MultiDeviceIndex(_native.NativeIndex):
    """Row shards on several GPUs of ONE process (the in-process counterpart of ShardedDenseSearcher).

    Since round 2 this is the library's own multi-device index (`rq_index_create(dim, n_devices, device_ids)`,
    include/rq.h): contiguous pieces of every appended block per device, one stream per device, every device enqueued
    before any is waited for, k keys per shard copied back, canonical host merge -- all inside librq_hip.so.  The class
    is kept as the name `DenseIndex(devices=[...])` constructs."""

    def __init__(self, dim: int, devices: Sequence[int]):
        if not devices:
            raise ValueError("devices must name at least one GPU")
        super().__init__(dim, devices=[int(d) for d in devices])
