#!/usr/bin/env python3
"""bench.py - vectors scanned / second for the sqlite-vector hot path on MI355X.

(Round 6: this file holds the contract - argument parsing, the N-rank launch and self check, main() and the line's `summary`; the shared
machinery - workloads, seeded corpora, timed runners, CPU legs, roofline pricing - is bench_common.py, the `also.*` sub-results and the other
workloads are bench_extras.py.  `python bench.py` is the only entry point.)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--workload c1|c2|c3|c5|c3b|c5h|c5f|c5q|c5l|stage|sql] [--no-also]

Default line (BASELINE.json configs[1]): 10M x 384 f32, L2, top-20, single query, corpus resident in HBM, answered by
the PLAIN f32 scan kernel (vg_scan_kernel; the shadow-copy filter is switched off for this corpus), so that
`roofline` is SURVEY 8(d)'s figure: algorithmic bytes N*D*4 = 15.36 GB per launch / the kernel's mean duration from HIP
events recorded around it on its own stream inside the timed region, against the 8 TB/s HBM3E peak.  `roofline.traffic` is the
PMC FETCH_SIZE figure of profiles/pmc_traffic.json - emitted only while that file's kernel-source hash matches this build.
A "step" is ONE complete query: upload the query, scan the whole shard, reduce to k candidates, bring the k keys back
and decode them - what vector_full_scan's xFilter costs once the corpus is staged.

The same run appends (N = 1, default workload only; --no-also skips them, --also filter,c3,c5,matrix,c4,long,clustered,c1 picks):
  filter_scan  the SAME queries over the SAME corpus through the product's default path for a corpus of this size: the lower-bound
               filter over an int8 shadow copy + exact f32 re-evaluation of the candidates (vg_scan_filter.h).  It answers the
               f32 question with the f32 scan's rowids and distance bits but STREAMS int8: its rate is priced on the bytes it
               streams (dtype_streamed / frac_on_streamed) and is never reported under dtype f32 / roofline.frac.
  also.c3      10M x 768 uint8 cosine top-20 (configs[2]; bytes = an f32 U[0,1) source quantized with the reference's formula):
               own roofline (7.68 GB per launch, the PLAIN kernel) and cpu_baseline; tie_order (the reference's result order -
               the SQL surface's default for integer types - next to (distance, position) order, with the replay counters);
               nibble_filter_probe + filter_scan (the high-nibble filter, probed on a prefix, same answers, half the bytes)
  also.c5      1024 queries x 10M x 384 f32 dot top-20 (configs[4]): own roofline (7.864 TFLOP per launch against the
               157.3 TF f32 MFMA peak, the f32 matrix-core kernel) and cpu_baseline; filter_batch (the product's default for a
               corpus of this size: bf16 matrix-core filter + exact f32 re-evaluation, priced on the bf16 peak)
  also.long_rows       1024 queries x 10M x 1536 f32 dot top-20 (not a BASELINE config): the default path for such rows (int8 matrix-core
               filter, a tile's K in three ring parts; priced on the int8 peak) with `against_single_scans` (what such batches were
               before) and `bf16_ksplit_batch` (round 4's K-split bf16 kernel, the same batches, bit-identical answers)
  also.kernel_matrix   f16 / bf16 / int8 x L2 / cosine at 10M x 384 through their PLAIN kernels: frac of the HBM peak each
  also.c4_one_gpu      north_star's target sentence: 100M x 384 f32 L2 resident on ONE device, the plain kernel

N > 1: configs[3] - the corpus row-sharded over the ranks, 12.5M x 384 f32 rows per rank (8 ranks = the stated 100M rows; weak
scaling: every rank count uses that shard size), each step = local scan -> all_gather of the per-shard candidate keys over
RCCL/xGMI -> rank 0 merges.  value = rows scanned by ALL ranks / max-over-ranks time; roofline.per_rank lists every rank's kernel.
`python bench.py --gpus N` WITHOUT a rank environment starts the N ranks itself (torch.distributed.run, 127.0.0.1) and refuses,
non-zero and without printing a line, when fewer than N devices are visible or when a launcher's WORLD_SIZE disagrees with --gpus.
(VG_BENCH_SHARE_DEVICES=1: the ranks share the visible devices and exchange over gloo - a functional check of the N-rank path on a
smaller box; --selftest-launch: the launch + exchange + merge plumbing with fabricated keys, no device - tests/test_bench_launch.py.)

cpu_baseline (rank 0, every N) = the reference's own kernel + top-k loop (oracle/_ref/libref_avx2.so, built from /root/reference
by oracle/Makefile) on ONE host core - the reference is single-threaded - over a bounded sample, timed in this run; all_cores =
a row-split of ONE matrix over every logical core of the host - thread i runs that loop over its range, a query's answer is the merge of
the per-range lists (checked against the unsplit scan).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

from bench_common import *                              # noqa: F401,F403  (tools/ use bench.make_shard, bench.kernel_source_hash ...)
from bench_extras import *                              # noqa: F401,F403



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU shard (default: 10M on one GPU, 12.5M per rank on several = config C4's shard)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["stage", "sql"])
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=1024, help="queries per batch (workload c5)")
    ap.add_argument("--no-also", action="store_true", help="default workload: skip the filter_scan / c3 / c5 sub-results")
    ap.add_argument("--also", default="filter,c3,c5,matrix,c4,long,clustered,c1", help="default workload: which sub-results to append (filter,c3,c5,matrix,c4,long,clustered,c1)")
    ap.add_argument("--inprocess", action="store_true",
                    help="--gpus N in ONE process: the product's own multi-device form (vg_shards: block-cyclic deal over the N devices, "
                         "candidate gather by host copies and by one grouped RCCL all-gather), timed per query, same JSON contract")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="no GPU work: every rank fabricates its candidate keys and runs the N-rank exchange + merge + timing plumbing "
                         "over gloo (what tests/test_bench_launch.py drives on a CPU-only box)")
    return ap.parse_args()


C4_EXPECTED = os.path.join(ROOT, "tests", "golden", "bench_c4_expected.json")


def c4_self_check(args, runner, dist, n_rows, n_gpus, rank, k, share, torch):
    """one untimed step with query 0; rank 0 compares the merged (rowid, distance bits) with the recorded one-device answer and
    tells the others.  None when no recorded answer applies (other shard size / k / file absent)."""
    try:
        exp = json.load(open(C4_EXPECTED))
    except (OSError, ValueError):
        return None
    want = exp.get("per_world", {}).get(str(n_gpus))
    if want is None or exp.get("rows_per_rank") != n_rows or exp.get("k") != k:
        return None
    runner.step(0)
    flag = torch.zeros(1, dtype=torch.int64, device="cpu" if (share or dist is None) else "cuda")
    res = {"query": 0, "expected_from": "tests/golden/bench_c4_expected.json (the same seeded shards scanned one by one on ONE device, merged on the host)"}
    if rank == 0:
        got_ids = [int(p) + 1 for p in runner.last["pos"]]
        got_bits = [int(x) for x in np.asarray(runner.last["dist"], dtype=np.float32).view(np.uint32)]
        res["rowids_match"] = got_ids == want["rowids"]
        res["distance_bits_match"] = got_bits == want["dist_bits"]
        if not (res["rowids_match"] and res["distance_bits_match"]):
            res["failed"] = True
            res["got_rowids"], res["want_rowids"] = got_ids, want["rowids"]
            flag[0] = 1
    if dist is not None:
        dist.broadcast(flag, src=0)
    if int(flag.item()) != 0:
        res["failed"] = True
    return res


def self_launch(args):
    """`python bench.py --gpus N` with no rank environment: start the N ranks ourselves (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1) and hand their exit code back.  Fails loudly - before anything is
    launched - when fewer than N devices are visible (VG_BENCH_SHARE_DEVICES=1: the ranks share the visible devices round
    robin and exchange over gloo instead of RCCL - a functional check of the N-rank path on a smaller box, not a measurement)."""
    import socket
    import subprocess
    if not args.selftest_launch:
        import __graft_entry__ as g
        have = g.load_package().device_count()
        if have < args.gpus and os.environ.get("VG_BENCH_SHARE_DEVICES") != "1":
            print("bench.py: --gpus %d but only %d HIP device(s) visible; refusing to report an n_gpus=%d line from fewer devices"
                  % (args.gpus, have, args.gpus), file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def selftest_launch(args, rank, world):
    """The N-rank plumbing without a device: rendezvous, the per-step exchange + merge of shard.py on fabricated candidate
    keys, barrier-bracketed timing, MAX over ranks, one JSON line from rank 0 with n_gpus = N.  Measures nothing."""
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    shard = load_shard_module()
    dist.init_process_group(backend="gloo")
    k, n_rows = args.k, 1000
    rng = np.random.default_rng(7 + rank)
    d = np.sort(rng.random(64).astype(np.float32))
    bits = d.view(np.uint32).astype(np.uint64)
    keys = ((bits ^ np.uint64(0x80000000)) << np.uint64(32)) | np.arange(64, dtype=np.uint64)
    keys[k:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    local = torch.from_numpy(keys.view(np.int64).copy())
    gathered = torch.empty((world, 64), dtype=torch.int64)
    offsets = [i * n_rows for i in range(world)]
    res = None
    for _ in range(args.warmup):
        res = shard.gather_and_merge(pkg, dist, local, gathered, offsets, k)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = shard.gather_and_merge(pkg, dist, local, gathered, offsets, k)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        assert res is not None and len(res[0]) == k and np.all(np.diff(res[1]) >= 0)
        print(json.dumps({"metric": "selftest: N-rank launch + candidate exchange + merge (no scan, no device)", "value": None,
                          "unit": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": float(t.item()) / args.steps * 1e3, "data": "fabricated candidate keys",
                          "config": {"workload": "selftest-launch", "backend": "gloo"}}))
    dist.destroy_process_group()
    return 0


def rccl_record(torch, dist, share, n_dev):
    """what the collective layer saw, for the N > 1 line: the backend, the RCCL version torch was built against, the communicator's size,
    and an all_gather of (rank, device) pairs run once, untimed - a SCALE record then states by itself which ranks exchanged candidates"""
    rec = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices_visible": n_dev}
    try:
        rec["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())       # (torch's name for RCCL on ROCm)
    except Exception as e:
        rec["nccl_version"] = "unavailable: %r" % (e,)
    try:
        dev = "cpu" if share else "cuda"
        mine = torch.tensor([dist.get_rank(), torch.cuda.current_device()], dtype=torch.int64, device=dev)
        got = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(got, mine)                         # (the list form: gloo has no all_gather_into_tensor for this shape)
        if not share:
            torch.cuda.synchronize()
        pairs = [t.cpu().tolist() for t in got]
        rec["ranks_in_communicator"] = len(set(p[0] for p in pairs))
        rec["rank_devices"] = [p[1] for p in sorted(pairs)]
    except Exception as e:
        rec["ranks_in_communicator"] = "all_gather failed: %r" % (e,)
    return rec


def main():
    args = parse()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.inprocess:
        return bench_inprocess(args)
    if "RANK" not in os.environ and args.gpus > 1:
        return self_launch(args)                 # N ranks of this same script; their rank 0 prints the line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but launched with WORLD_SIZE=%d: the two must agree (no line is printed)" % (args.gpus, world),
                  file=sys.stderr)
        return 2
    if args.selftest_launch:
        return selftest_launch(args, rank, world)
    import torch
    import torch.distributed as dist

    # VG_BENCH_FORCE_DIST=1 runs the collective path even with one rank (RCCL smoke test on a 1-GPU box)
    use_dist = world > 1 or (os.environ.get("VG_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    n_dev = torch.cuda.device_count()
    share = os.environ.get("VG_BENCH_SHARE_DEVICES") == "1" and n_dev < world
    if n_dev < world and not share:
        if rank == 0:
            print("bench.py: %d ranks but only %d HIP device(s) visible" % (world, n_dev), file=sys.stderr)
        return 2
    device_index = local_rank % max(1, n_dev)
    torch.cuda.set_device(device_index)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")          # RCCL refuses two ranks on one device: functional check only
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
    n_gpus = world

    import __graft_entry__ as g
    pkg = g.load_package()
    pkg.AUTO_RELOAD = False                  # (the switches are read at corpus creation; this file re-reads them where it changes the environment)
    shard = load_shard_module()
    if args.workload == "c1":
        if not args.rows:
            args.rows = 10_000
        return bench_sql(args, pkg, torch)
    if args.workload == "stage":
        return bench_stage(args, pkg, torch)
    if args.workload == "sql":
        return bench_sql_dropin(args, pkg, torch)
    vt, np_dtype, dim, metric, desc = WORKLOADS[args.workload]
    if args.workload == "c5f":
        os.environ["VG_F32_FILTER"] = "1"
        os.environ["VG_BATCH_Q8"] = "0"
    k = args.k
    # one GPU: config C2 / C3 / C5 as stated (10M rows).  Several GPUs: config C4's shard, 12.5M rows per rank - 8 ranks scan
    # the stated 100M rows; every rank count keeps that shard size (weak scaling)
    n_rows = args.rows if args.rows else (10_000_000 if n_gpus == 1 else 12_500_000)
    if n_gpus > 1 and args.workload == "c2":
        desc = "%gMx384 f32 L2 top-20, corpus row-sharded across %d MI355X (%gM rows per rank) + RCCL candidate gather" % (
            n_rows * n_gpus / 1e6, n_gpus, n_rows / 1e6)
    elif n_rows != 10_000_000:
        desc = desc.replace("10M", "%gM" % (n_rows / 1e6))
    the_dist = dist if use_dist else None

    corpus = make_shard(pkg, torch, vt, dim, n_rows, 42 + rank, device_index)
    corpus.set_rowid_base(1 + rank * n_rows)
    corpus.set_profiling(True)
    if args.workload in ("c5", "c3b", "c5h", "c5f", "c5q", "c5l"):
        if args.workload == "c5":
            corpus.set_scan_filter(0)                                 # the f32 matrix-core kernel (the filters are the default for a corpus of this size)
        out = run_batched(args, pkg, torch, corpus, args.workload, n_rows, dim, metric, k, desc, the_dist, shard, n_gpus, rank,
                          share=share)
        rccl = rccl_record(torch, dist, share, n_dev) if use_dist else None      # (a collective: every rank)
        if out is not None:
            if rccl is not None:
                out["rccl"] = rccl
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = batch_cpu_baseline(args, vt, np_dtype, dim, metric, k, sample=(pkg, torch, n_rows, 42 + rank))
            print(json.dumps(out))
        corpus.close()
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    # queries: a different one every step, pre-generated on the host
    rng = np.random.default_rng(43)
    nq = args.steps + args.warmup
    queries = rng.standard_normal((nq, dim), dtype=np.float32) if vt == pkg.F32 else c3_queries(nq, dim)

    # THE line: the plain scan kernel on SURVEY 8(d)'s basis.  The shadow-copy filter is switched off for this corpus
    # (it is the product's default for f32 corpora of this size and is reported on its own below).
    corpus.set_scan_filter(0)
    runner = SingleQueryRunner(pkg, torch, the_dist, shard, corpus, vt, dim, metric, k, n_rows, n_gpus, queries, share=share)
    check = None
    if args.workload == "c2":
        # The N-rank answer checked against an answer that was NOT computed by N ranks: query 0 over the same seeded shards, scanned
        # one after the other on ONE device and merged on the host (tools/make_bench_c4_expected.py -> tests/golden/bench_c4_expected.json).
        # A wrong merged top-20 ends the run without a line.
        check = c4_self_check(args, runner, the_dist, n_rows, n_gpus, rank, k, share, torch)
        if check is not None and check.get("failed"):
            if rank == 0:
                print("bench.py: SELF CHECK FAILED - the %d-rank top-%d of query 0 is not the one-device answer: %s" % (n_gpus, k, json.dumps(check)), file=sys.stderr)
            if use_dist:
                dist.destroy_process_group()
            return 3
    out, _ = single_query_line(args, pkg, runner, corpus, args.workload, vt, dim, metric, k, n_rows, n_gpus, desc)
    if check is not None:
        out["self_check"] = check
    if use_dist:
        # every rank's own dominant-kernel time (HIP events on its stream): the line's roofline is rank 0's, the others ride along
        mine = torch.tensor([out["roofline"]["kernel_ms"]], dtype=torch.float64, device="cpu" if share else "cuda")
        allk = [torch.zeros_like(mine) for _ in range(n_gpus)]
        dist.all_gather(allk, mine)
        per = [float(t.item()) for t in allk]
        ab = out["roofline"]["algorithmic_bytes_per_launch"]
        out["roofline"]["per_rank"] = [{"rank": i, "kernel_ms": ms, "achieved": ab / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                                        "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0} for i, ms in enumerate(per)]
        out["rccl"] = rccl_record(torch, dist, share, n_dev)
        if share:
            out["config"]["note"] = "VG_BENCH_SHARE_DEVICES=1: %d ranks share %d device(s), exchange over gloo - a functional check, not a measurement" % (n_gpus, n_dev)
    plain_last = dict(runner.last)
    if rank == 0 and not args.no_cpu_baseline:
        try:
            want = max(args.cpu_sample_rows, min(4_000_000, (os.cpu_count() or 1) * 8192))
            sample = corpus_sample(pkg, torch, vt, dim, n_rows, 42 + rank, want)
            out["cpu_baseline"] = cpu_baseline(vt, np_dtype, dim, metric, k, args.cpu_sample_rows, rows=sample, queries=queries[args.warmup:args.warmup + 8])
        except Exception as e:                                        # the checker is optional on a bare box
            out["cpu_baseline"] = {"value": None, "unit": "vectors/s", "cores": 0, "kind": "port", "sample": "unavailable: %r" % (e,)}

    also_set = set() if args.no_also else set(x for x in args.also.split(",") if x)
    if n_gpus == 1 and args.workload == "c2" and "filter" in also_set:
        # ---- the same queries through the filter scan (the product's default path for this corpus)
        out["filter_scan"] = filter_scan_object(args, pkg, corpus, runner, metric, vt, dim, n_rows, plain_last)
    if n_gpus == 1 and args.workload == "c2" and (also_set & {"c3", "c5", "matrix", "c4", "long", "clustered", "c1"}):
        # ---- configs[2] over its own corpus, then configs[4] over the f32 corpus (the MFMA-bound batch after the HBM-bound
        # lines: they are not timed on a package it has just heated), then the plain-kernel matrix and 100M x 384 on this device
        also = {}
        if "c4" in also_set:
            # north_star's target configuration runs FIRST among the extras, next to the C2 corpus (15 + 154 GB): the same kernel
            # measured 0.834 of the peak at the END of this run (round 3) and 0.861 in a process of its own on the same day
            # (profiles/r6a_scan_10m_vs_100m_*.jsonl; UTCL1 misses 0.02 % of the requests at either size, the batch order makes no
            # difference) - what it ran behind, the MFMA-bound batches and the kernel matrix, is what it was paying for
            also["c4_one_gpu"] = also_c4_one_gpu(args, pkg, torch, shard, k, device_index)
        if "c3" in also_set:
            also["c3"] = also_c3(args, pkg, torch, shard, also_set, n_rows, k, nq, device_index)
        if "c5" in also_set:
            also["c5"] = also_c5(args, pkg, torch, corpus, n_rows, k)
        corpus.close()
        corpus = None
        torch.cuda.empty_cache()
        if "matrix" in also_set:
            also["kernel_matrix"] = also_kernel_matrix(args, pkg, torch, shard, n_rows, k, device_index)
        if "long" in also_set:
            also["long_rows"] = also_long_rows(args, pkg, torch, k, device_index)
        if "clustered" in also_set:
            also["clustered"] = also_clustered(args, pkg, torch, k, device_index)
        if "c1" in also_set:
            also["c1"] = also_c1(args, pkg, torch)
        out["also"] = also
    if rank == 0:
        if "also" in out:
            out["summary"] = make_summary(out)       # LAST key of the line: what north_star asks for, where a tail of the line still shows it
        print(json.dumps(out))
    if corpus is not None:
        corpus.close()
    if use_dist:
        dist.barrier()                 # (rank 0 was timing the CPU baseline meanwhile)
        dist.destroy_process_group()
    return 0


def make_summary(out):
    """a compact digest (< 1 KB) of the line's extras - the fractions north_star's target sentence is about - as the line's last key"""
    def g(d, *path):
        for p in path:
            if not isinstance(d, dict) or p not in d:
                return None
            d = d[p]
        return float("%.4g" % d) if isinstance(d, float) else d
    a = out.get("also", {})
    c5 = a.get("c5", {})
    return {
        "c2_10m_frac_of_hbm_peak": g(out, "roofline", "frac"), "c2_kernel_ms": g(out, "roofline", "kernel_ms"),
        "c4_100m_one_gpu_frac_of_hbm_peak": g(a, "c4_one_gpu", "roofline", "frac"), "c4_100m_kernel_ms": g(a, "c4_one_gpu", "roofline", "kernel_ms"),
        "c3_u8_cosine_frac_of_hbm_peak": g(a, "c3", "roofline", "frac"),
        "c5_f32_mfma": {"ms_per_step": g(c5, "ms_per_step"), "frac_of_f32_mfma_peak": g(c5, "roofline", "frac")},
        "c5_bf16_filter": {"ms_per_step": g(c5, "filter_batch", "ms_per_step"), "frac_of_bf16_peak": g(c5, "filter_batch", "frac_of_bf16_peak")},
        "c5_default_int8_filter": {"ms_per_step": g(c5, "int8_filter_batch", "ms_per_step"), "frac_of_int8_peak": g(c5, "int8_filter_batch", "frac_of_int8_peak"),
                                   "batch_path": g(c5, "int8_filter_batch", "batch_path"), "hbm_traffic_bytes_per_batch": g(c5, "int8_filter_batch", "traffic"),
                                   "bit_identical_to_bf16_filter": g(c5, "int8_filter_batch", "last_batch_bit_identical_to_the_bf16_filter"),
                                   "max_rel_vs_reference_kernel": g(c5, "int8_filter_batch", "last_batch_max_rel_difference_from_the_reference_kernel"),
                                   "small_batches_ms_per_batch": g(c5, "int8_filter_batch", "small_batches_ms_per_batch")},
        "long_rows_1536": {"ms_per_step": g(a, "long_rows", "ms_per_step"), "frac": g(a, "long_rows", "roofline", "frac"),
                           "of": "int8 peak" if g(a, "long_rows", "roofline", "batch_path") == 7 else "bf16 peak",
                           "batch_path": g(a, "long_rows", "roofline", "batch_path"), "hbm_traffic_bytes_per_batch": g(a, "long_rows", "roofline", "traffic"),
                           "bf16_ksplit_ms_per_step": g(a, "long_rows", "bf16_ksplit_batch", "ms_per_step"),
                           "bit_identical_to_bf16_ksplit": g(a, "long_rows", "bf16_ksplit_batch", "last_batch_bit_identical_to_the_default_path")},
        # what a user of vector_full_scan / vector_quantize_scan gets by default at this size: the filter scans
        "c2_default_filter_scan": {"ms_per_step": g(out, "filter_scan", "ms_per_step"), "kernel_ms": g(out, "filter_scan", "kernel_ms"),
                                   "frac_on_streamed": g(out, "filter_scan", "frac_on_streamed"),
                                   "exact_evaluations_per_query": g(out, "filter_scan", "exact_evaluations_per_query"),
                                   "outside_kernel_us": g(out, "filter_scan", "caller_view", "outside_kernel_us")},
        "c3_default_nibble_filter_scan": {"ms_per_step": g(a, "c3", "filter_scan", "ms_per_step"), "kernel_ms": g(a, "c3", "filter_scan", "kernel_ms"),
                                          "frac_on_streamed": g(a, "c3", "filter_scan", "frac_on_streamed"),
                                          "exact_evaluations_per_query": g(a, "c3", "filter_scan", "exact_evaluations_per_query"),
                                          "outside_kernel_us": g(a, "c3", "filter_scan", "caller_view", "outside_kernel_us")},
        "clustered": g(a, "clustered", "summary"),
        "c1_sql_p50_ms": g(a, "c1", "p50_query_latency_ms"),
        "cpu_reference_1_core_vectors_per_s": g(out, "cpu_baseline", "value"),
        "cpu_reference_all_cores": {"vectors_per_s": g(out, "cpu_baseline", "all_cores", "value"), "cores": g(out, "cpu_baseline", "all_cores", "cores"),
                                    "GB_per_s": g(out, "cpu_baseline", "all_cores", "GB_per_s"), "x_one_core": g(out, "cpu_baseline", "all_cores", "x_one_core"),
                                    "limited_by": g(out, "cpu_baseline", "all_cores", "limited_by"), "cpu_model": g(out, "cpu_baseline", "host", "cpu_model")},
    }




if __name__ == "__main__":
    sys.exit(main() or 0)
