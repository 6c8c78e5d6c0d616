"""The N > 1 exchange step rehearsed on the CPU: 2 ranks, gloo, 127.0.0.1.

Each rank owns a contiguous row block; local top-k comes from the oracle here (the HIP kernels need a
GPU -- their shard results are checked against the same oracle in test_gpu_parity.py), the packed
keys travel through a real all_gather and are merged on the host.  The merged answer must equal the
oracle's answer over the whole corpus."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_rows, dim, k, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import dense_oracle as orc
    import rag_uq_amd  # noqa: F401
    from rag_uq_amd import distributed as d

    dist.init_process_group("gloo", rank=rank, world_size=world)
    x16 = orc.synthetic_corpus(n_rows, dim, seed=21)
    x16[40:50] = x16[3]                      # duplicates straddling the shard boundary exercise the row tie-break
    x16[n_rows // 2 - 2: n_rows // 2 + 2] = x16[3]
    q = orc.synthetic_queries(6, dim, seed=22)
    q[1] = x16[3].astype(np.float32)
    q[2] = 0
    lo, hi = d.shard_bounds(n_rows, world, rank)
    ls, lr = orc.dense_topk(q, x16[lo:hi], k, row_offset=lo)
    ms, mr = d.gather_merge_host(ls, lr, k)
    gs, gr = orc.dense_topk(q, x16, k)
    ok = bool(np.array_equal(mr, gr) and np.array_equal(ms, gs))
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = ok


@pytest.mark.parametrize("n_rows,k", [(301, 10), (64, 40)])
def test_two_rank_gloo_gather_and_merge(n_rows, k):
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, 48, k, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {0: True, 1: True}


def test_shard_bounds_cover_everything():
    from rag_uq_amd import distributed as d
    for n, w in [(10, 3), (1_000_000, 8), (5, 8), (0, 2)]:
        spans = [d.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
