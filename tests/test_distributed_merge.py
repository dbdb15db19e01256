"""The N>1 path on CPU: two processes (gloo, 127.0.0.1) run the same gather + merge step bench.py runs over RCCL.
Each rank fabricates the 64-key candidate list its shard would produce (keys are what vg_scan_topk_device emits:
order-preserving float image << 32 | local position), the ranks exchange them with ONE all_gather and rank 0 merges.
The merged (global position, distance) list must equal a single-process top-k over the concatenated shards,
including ties that straddle the shard border."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

K = 20
ROWS = [700, 900]          # rows per rank (ragged on purpose)
EMPTY = 0xFFFFFFFFFFFFFFFF


def shard_distances(rank):
    rng = np.random.default_rng(100 + rank)
    d = rng.integers(0, 40, ROWS[rank]).astype(np.float32)        # many exact ties, also across shards
    d[5] = -np.inf if rank == 1 else d[5]
    d[7] = np.nan
    d[9] = np.inf
    return d


def keys_of(d, k):
    """what one shard's scan returns: k best (distance, position) packed, ascending, EMPTY padded to 64"""
    order = sorted((float(v), i) for i, v in enumerate(d) if v < np.inf)[:k]
    out = np.full(64, EMPTY, dtype=np.uint64)
    for j, (v, i) in enumerate(order):
        b = int(np.float32(v).view(np.uint32))
        s = b ^ (0xFFFFFFFF if b >> 31 else 0x80000000)
        out[j] = (s << 32) | i
    return out


def worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    pkg = g.load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("vg_shard", os.path.join(ROOT, "sqlite-vector_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    local = torch.from_numpy(keys_of(shard_distances(rank), K).view(np.int64).copy())
    gathered = torch.empty((world, 64), dtype=torch.int64)
    res = shard.gather_and_merge(pkg, dist, local, gathered, shard.row_offsets(ROWS), K)
    dist.barrier()
    if rank == 0:
        q.put((res[0].tolist(), res[1].tolist()))
    dist.destroy_process_group()


def test_two_rank_gather_and_merge_equals_single_process():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    pos, dd = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    allv = np.concatenate([shard_distances(0), shard_distances(1)])
    want = sorted((float(v), i) for i, v in enumerate(allv) if v < np.inf)[:K]
    assert pos == [i for _, i in want]
    assert dd == [v for v, _ in want]
    assert dd[0] == -np.inf and pos[0] == ROWS[0] + 5


# ---- the batched exchange (config C5 sharded by rows): nq queries, k keys per (rank, query)

NQ = 5


def batch_distances(rank, q):
    rng = np.random.default_rng(1000 + 10 * rank + q)
    return rng.integers(0, 25, ROWS[rank]).astype(np.float32)


def batch_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    pkg = g.load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("vg_shard", os.path.join(ROOT, "sqlite-vector_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    local = np.stack([keys_of(batch_distances(rank, i), K)[:K] for i in range(NQ)])          # [nq, k]
    gathered = torch.empty((world, NQ, K), dtype=torch.int64)
    res = shard.gather_and_merge_batch(pkg, dist, torch.from_numpy(local.view(np.int64).copy()), gathered,
                                       shard.row_offsets(ROWS), K)
    dist.barrier()
    if rank == 0:
        q.put((res[0].tolist(), res[2].tolist()))
    dist.destroy_process_group()


def test_two_rank_batched_gather_and_merge():
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    pos, cnt = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i in range(NQ):
        allv = np.concatenate([batch_distances(0, i), batch_distances(1, i)])
        want = sorted((float(v), j) for j, v in enumerate(allv))[:K]
        assert cnt[i] == K and pos[i] == [j for _, j in want]
