#!/usr/bin/env python3
"""What `bench.py --gpus N` must answer for query 0 (tests/golden/bench_c4_expected.json): the N ranks' seeded shards (seed 42 + r,
12.5M x 384 f32 each, bench.make_shard) scanned ONE AFTER THE OTHER on one device with the plain kernel, their 64 candidate keys
merged on the host per world size 1, 2, 4, 8.  Run on a GPU box:  python tools/make_bench_c4_expected.py > gpurun_out/<tag>/bench_c4_expected.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    import bench
    pkg = g.load_package()
    rows, dim, k = 12_500_000, 384, 20
    q = np.random.default_rng(43).standard_normal((1, dim), dtype=np.float32)[0]
    keys = []
    for r in range(8):
        c = bench.make_shard(pkg, torch, pkg.F32, dim, rows, 42 + r, 0)
        c.set_scan_filter(0)
        kk = np.full(64, pkg.KEY_EMPTY, dtype=np.uint64)
        cnt = pkg.C.c_int(0)
        pkg._check(pkg.lib().vg_scan_topk_keys(c.h, pkg.L2, pkg._ptr(q), k, pkg._ptr(kk), pkg.C.byref(cnt)))
        keys.append(kk)
        c.close()
        torch.cuda.empty_cache()
    out = {"rows_per_rank": rows, "dim": dim, "k": k, "query_seed": 43, "shard_seeds": "42 + rank", "per_world": {}}
    for n in (1, 2, 4, 8):
        pos, dist = pkg.merge_keys(np.stack(keys[:n]), [i * rows for i in range(n)], k)
        out["per_world"][str(n)] = {"rowids": [int(p) + 1 for p in pos], "dist_bits": [int(x) for x in np.asarray(dist, dtype=np.float32).view(np.uint32)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
