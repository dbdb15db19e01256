#!/usr/bin/env python3
"""vg_shards' candidate exchange, both forms timed per query on whatever devices this box has:

    python tools/shards_gather_bench.py [--rows-per-device 4000000] [--dim 384] [--queries 50]

  host   every shard copies its 64 keys back behind its scan; one host thread waits for the S streams
  rccl   one grouped ncclAllGather of 64 keys per shard over RCCL / xGMI on the scan streams + one copy from device 0

One line of JSON: per-query milliseconds (p50 / mean) for both, the devices used, and whether the results were bit-identical.
On a 1-GPU box the RCCL communicator has one rank (a functional check); the comparison the design question needs is the
multi-device one."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-device", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--queries", type=int, default=50)
    ap.add_argument("--k", type=int, default=20)
    args = ap.parse_args()
    pkg = g.load_package()
    ndev = pkg.device_count()
    devs = list(range(ndev))
    rng = np.random.default_rng(1)
    block = rng.standard_normal((1 << 18, args.dim), dtype=np.float32)
    sh = pkg.Shards(pkg.F32, args.dim, devs, block_rows=65536)
    total = args.rows_per_device * ndev
    done = 0
    while done < total:
        take = min(block.shape[0], total - done)
        sh.append(block[:take])
        done += take
    sh.set_scan_filter(0)
    qs = rng.standard_normal((args.queries + 5, args.dim), dtype=np.float32)
    out = {"devices": devs, "rows": total, "dim": args.dim, "k": args.k, "queries": args.queries}
    res = {}
    for mode in ("host", "rccl"):
        sh.set_gather(mode)
        for i in range(5):
            sh.scan_topk(pkg.L2, qs[i], args.k)
        lat, answers = [], []
        for i in range(args.queries):
            t0 = time.perf_counter()
            ids, d = sh.scan_topk(pkg.L2, qs[5 + i], args.k)
            lat.append(time.perf_counter() - t0)
            answers.append((ids.tolist(), d.tolist()))
        res[mode] = answers
        out[mode] = {"p50_ms": float(np.median(lat) * 1e3), "mean_ms": float(np.mean(lat) * 1e3)}
    st = sh.gather_stats()
    out["rccl_served"] = st["rccl"]
    out["identical_results"] = res["host"] == res["rccl"]
    print(json.dumps(out))
    sh.close()


if __name__ == "__main__":
    main()
