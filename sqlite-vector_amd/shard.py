"""Row-range sharding across GPUs, one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).

The scan shards by contiguous row range and needs exactly one exchange per query: every rank's 64 packed candidate
keys (512 B).  This module holds that exchange + merge so that bench.py, tests (gloo on CPU, world_size 2) and any
host program share one implementation.  No distance is computed here.
"""
import numpy as np


def row_offsets(rows_per_rank):
    """global scan position of each rank's first row (rank r holds rows [off[r], off[r] + rows_per_rank[r]))"""
    off, acc = [], 0
    for n in rows_per_rank:
        off.append(acc)
        acc += int(n)
    return off


def gather_and_merge(pkg, dist, local_keys, gathered, offsets, k, dst=0, host_buf=None, sync=None):
    """One exchange step of a sharded query.

    local_keys : int64 tensor [64] on this rank's device (output of vg_scan_topk_device viewed as int64)
    gathered   : int64 tensor [world, 64] on the same device (reused across queries)
    Returns (global_positions, distances) as numpy arrays on rank `dst`, None on the other ranks.
    The collective is a single all_gather_into_tensor (direct one-hop exchange on the xGMI mesh; the payload is
    latency-bound, 512 B per rank).
    """
    dist.all_gather_into_tensor(gathered.view(-1), local_keys)
    if dist.get_rank() != dst:
        return None
    if host_buf is not None:
        host_buf.copy_(gathered, non_blocking=True)
        if sync is not None:
            sync()
        keys = host_buf.numpy()
    else:
        keys = gathered.cpu().numpy()
    return pkg.merge_keys(keys.view(np.uint64), offsets, k)


def gather_and_merge_batch(pkg, dist, local_keys, gathered, offsets, k, dst=0):
    """The same exchange for a batch of queries (config C5 sharded by rows).

    local_keys : int64 tensor [nq, k] on this rank's device - vg_scan_topk_batch_keys of this shard, EMPTY padded
    gathered   : int64 tensor [world, nq, k] on the same device
    One all_gather_into_tensor of nq * k * 8 bytes per rank (160 KB at 1024 x 20); rank `dst` merges every query by
    (distance, shard, position).  Returns (global_positions [nq, k], distances [nq, k], counts [nq]) on `dst`.
    """
    dist.all_gather_into_tensor(gathered.view(-1), local_keys.reshape(-1))
    if dist.get_rank() != dst:
        return None
    keys = gathered.cpu().numpy().view(np.uint64)
    return pkg.merge_keys_batch(keys, offsets, k)
