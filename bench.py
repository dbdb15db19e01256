#!/usr/bin/env python3
"""bench.py - vectors scanned / second for the sqlite-vector hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--workload c1|c2|c3|c5|c3b|c5h|c5f|c5q|c5l|stage|sql] [--no-also]

Default line (BASELINE.json configs[1]): 10M x 384 f32, L2, top-20, single query, corpus resident in HBM, answered by
the PLAIN f32 scan kernel (vg_scan_kernel; the shadow-copy filter is switched off for this corpus), so that
`roofline` is SURVEY 8(d)'s figure: algorithmic bytes N*D*4 = 15.36 GB per launch / the kernel's mean duration from HIP
events recorded around it on its own stream inside the timed region, against the 8 TB/s HBM3E peak.  `roofline.traffic` is the
PMC FETCH_SIZE figure of profiles/pmc_traffic.json - emitted only while that file's kernel-source hash matches this build.
A "step" is ONE complete query: upload the query, scan the whole shard, reduce to k candidates, bring the k keys back
and decode them - what vector_full_scan's xFilter costs once the corpus is staged.

The same run appends (N = 1, default workload only; --no-also skips them, --also filter,c3,c5,matrix,c4,long picks):
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling

WORKLOADS = {
    # name: (type enum, numpy dtype, dim, metric enum, description)
    # configs[0], the reference's own CPU-runnable case, driven through SQL (python sqlite3 + load_extension): the
    # same statements against this repo's vector.so (GPU) and the reference's vector.so (oracle/_ref, CPU)
    "c1": (1, np.float32, 384, 1, "10kx384 f32 L2 top-20 through SQL: SELECT ... FROM vector_full_scan(...)"),
    "c2": (1, np.float32, 384, 1, "10Mx384 f32 L2 top-20 single-query"),
    "c3": (4, np.uint8, 768, 3, "10Mx768 u8 quantized cosine top-20 single-query"),
    # batched queries on the matrix cores (config #5); a step is one batch of --batch queries (the same batch on every shard)
    "c5": (1, np.float32, 384, 4, "batched 1024 queries x 10Mx384 f32 dot top-20 (MFMA Q x C^T + fused top-k)"),
    # the quantized counterpart of c5 (not a BASELINE config): config #3's corpus, a batch of queries, int8 matrix cores
    "c3b": (4, np.uint8, 768, 3, "batched 1024 queries x 10Mx768 u8 quantized cosine top-20 (int8 MFMA Q x C^T + fused top-k)"),
    # c5 over an f16 corpus (not a BASELINE config): matrix cores as a filter, the reference's f64 arithmetic for survivors
    "c5h": (2, np.float16, 384, 4, "batched 1024 queries x 10Mx384 f16 dot top-20 (default path: int8 MFMA filter over the shadow copy + exact f64 re-evaluation; VG_BATCH_Q8=0: the f16 MFMA filter)"),
    # c5 answered through the bf16 filter (VG_F32_FILTER=1: bf16 shadow copy on the matrix cores, f32 exact re-evaluation of the
    # survivors) instead of the f32 MFMA kernel - same question, same f32 distances, the GEMM at the bf16 rate
    "c5f": (1, np.float32, 384, 4, "batched 1024 queries x 10Mx384 f32 dot top-20 (bf16 MFMA filter over a shadow copy + exact f32 re-evaluation + fused top-k)"),
    # c5 through the product's DEFAULT path for a batch of this size (round 5): the int8 shadow copy on the integer matrix cores as the filter
    # (vg_batch_q8.hip), exact f32 re-evaluation of the pairs that pass - priced on the int8 MFMA rate, the instruction it runs on
    "c5q": (1, np.float32, 384, 4, "batched 1024 queries x 10Mx384 f32 dot top-20 (int8 MFMA filter over the int8 shadow copy + exact f32 re-evaluation)"),
    # long rows (not a BASELINE config): 1536-dimensional f32 embeddings - the K dimension split over the wavefronts of
    # a workgroup (vg_batch_hl.hip), bf16 shadow copy on the matrix cores, exact f32 re-evaluation; reported next to one scan per query
    "c5l": (1, np.float32, 1536, 4, "batched 1024 queries x 10Mx1536 f32 dot top-20 (default path: int8 MFMA filter, a tile's K in three ring parts + exact f32 re-evaluation; VG_BATCH_Q8=0: the K-split bf16 MFMA filter)"),
}
F16_MFMA_PEAK_TF = 2500.0      # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)
I8_MFMA_PEAK_TOPS = 3944.0     # int8 MFMA: no spec figure in the guide, its micro-benchmark ceiling (>= 3944 TOP/s)


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


def shard_blocks(pkg, torch, vt, dim, n_rows, seed):
    """the synthetic shard's rows as device tensors, block by block (the same seeded stream every time it is walked).
    f32 / f16 / bf16: N(0,1); uint8: SURVEY 8(d)'s C3 data - an f32 U[0,1) source quantized with the reference's formula
    (offset = min = 0, scale = 255 / (max - min) = 255, sqlite-vector.c:517-548): (uint8)(v * 255 + 0.5); int8: N(0,1) * 40 rounded"""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    block = 1_000_000
    for r0 in range(0, n_rows, block):
        nr = min(block, n_rows - r0)
        if vt == pkg.F32:
            t = torch.randn((nr, dim), generator=gen, device="cuda", dtype=torch.float32)
        elif vt in (pkg.F16, pkg.BF16):
            t = torch.randn((nr, dim), generator=gen, device="cuda", dtype=torch.float32).to(torch.float16 if vt == pkg.F16 else torch.bfloat16)
        elif vt == pkg.U8:
            t = torch.rand((nr, dim), generator=gen, device="cuda", dtype=torch.float32).mul_(255.0).add_(0.5).floor_().clamp_(0, 255).to(torch.uint8)
        else:
            t = torch.randn((nr, dim), generator=gen, device="cuda", dtype=torch.float32).mul_(40.0).round_().clamp_(-128, 127).to(torch.int8)
        torch.cuda.synchronize()
        yield r0, t
        del t


def make_shard(pkg, torch, vt, dim, n_rows, seed, device):
    """synthetic shard (shard_blocks) generated on the device in blocks and handed to the C-ABI as a raw device pointer"""
    corpus = pkg.Corpus(vt, dim, device=device, capacity=n_rows)
    es = pkg.TYPE_SIZE[vt]
    for r0, t in shard_blocks(pkg, torch, vt, dim, n_rows, seed):
        corpus.append_device(t.data_ptr(), t.shape[0], dim * es)
    torch.cuda.empty_cache()
    return corpus


_SAMPLES = {}


def corpus_sample(pkg, torch, vt, dim, n_rows, seed, want):
    """the first `want` rows of the synthetic shard (shard_blocks: the same seeded device stream the GPU corpus was built from), copied
    back to the host: SURVEY 8(d)'s "same inputs" - the CPU legs time the reference on rows the GPU scanned, not on a numpy look-alike"""
    want = int(min(want, n_rows))
    key = (vt, dim, seed)
    have = _SAMPLES.get(key)
    if have is not None and have.shape[0] >= want:
        return have[:want]
    parts, got = [], 0
    for r0, t in shard_blocks(pkg, torch, vt, dim, n_rows, seed):
        take = min(t.shape[0], want - got)
        h = t[:take].cpu()
        parts.append(h.view(torch.int16).numpy().view(np.uint16) if vt in (pkg.F16, pkg.BF16) else h.numpy())
        got += take
        if got >= want:
            break
    _SAMPLES[key] = np.ascontiguousarray(np.concatenate(parts))
    return _SAMPLES[key]


def rows_at(pkg, torch, vt, dim, n_rows, seed, positions):
    """{position: row} for a handful of positions of the synthetic shard, regenerated from its seeded device stream"""
    want = sorted(set(int(p) for p in positions))
    out, i = {}, 0
    for r0, t in shard_blocks(pkg, torch, vt, dim, n_rows, seed):
        while i < len(want) and want[i] < r0 + t.shape[0]:
            out[want[i]] = t[want[i] - r0].cpu().numpy().copy()
            i += 1
        if i >= len(want):
            break
    return out


def host_description():
    """what the CPU legs ran on: CPU model, logical CPUs of the host, the process' affinity mask, the cgroup CPU quota, free memory"""
    d = {"logical_cpus": os.cpu_count()}
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except Exception:
        cpus = list(range(os.cpu_count() or 1))
    d["affinity_cpus"] = len(cpus)
    d["affinity_cpus_list"] = cpus
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        models = [ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("model name")]
        d["cpu_model"] = models[0] if models else None
        d["sockets"] = len(set(ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("physical id"))) or None
    except Exception:
        d["cpu_model"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                d["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    per = float(f2.read().split()[0])
                d["cgroup_cpu_max"] = "%d %d" % (q, per)
                if q > 0:
                    quota = q / per
            break
        except Exception:
            continue
    d["cgroup_cpu_quota_cores"] = quota
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable"):
                    d["mem_available_bytes"] = int(ln.split()[1]) * 1024
    except Exception:
        pass
    return d


def cpu_baseline(vt, np_dtype, dim, metric, k, sample_rows, seconds=10.0, all_cores=True, rows=None, queries=None):
    """reference kernel + reference top-k loop, one core, bounded sample (`seconds` of CPU work).  rows / queries: rows of the GPU's own
    corpus (corpus_sample) and the queries the GPU timed; without them (a box without torch) a numpy stream of the same distribution."""
    from oracle import orc
    same_inputs = rows is not None and queries is not None
    if not same_inputs:
        rng = np.random.default_rng(42)
        if vt == 1:
            rows = rng.standard_normal((sample_rows, dim), dtype=np.float32)
            queries = rng.standard_normal((4, dim), dtype=np.float32)
        else:                                        # f32 U[0,1) quantized with the reference's formula
            rows = quantize_unit_uniform_np(rng.random((sample_rows, dim), dtype=np.float32))
            queries = quantize_unit_uniform_np(rng.random((4, dim), dtype=np.float32))
    queries = np.ascontiguousarray(queries)
    one = np.ascontiguousarray(rows[:sample_rows])
    sample_rows = one.shape[0]
    q = queries[0]
    kind, ref = "port", None
    if orc.have_ref():
        ref = orc.RefKernels("avx2")
        kind = "reference"
        work = lambda v, qq: ref.scan_topk(metric, vt, qq, v, k)         # noqa: E731
        label = "oracle/_ref/libref_avx2.so (reference distance-avx2.c kernel via dispatch table, backend %s)" % ref.backend_name
    else:
        work = lambda v, qq: orc.scan_topk_reference(orc.AVX2, metric, vt, qq, v, None, k)   # noqa: E731
        label = "oracle/liboracle.so (C restatement, AVX2 order, scalar)"
    work(one, q)                                 # warm (page in)
    reps, t0 = 0, time.perf_counter()
    while True:
        work(one, queries[reps % len(queries)])
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds or reps >= 40:
            break
    out = {"value": sample_rows * reps / el, "unit": "vectors/s", "cores": 1, "kind": kind,
           "sample": "%d queries over %s, %dx%d %s, top-%d, %s; host has %d logical cores" %
                     (reps, "the first rows of the GPU's own corpus (copied back) with the queries the GPU timed" if same_inputs
                      else "a numpy sample of the corpus' distribution", sample_rows, dim, np.dtype(np_dtype).name, k, label, os.cpu_count())}
    if not all_cores:
        return out
    # A ROW SPLIT of one corpus over the host cores THIS PROCESS MAY USE (a generous upper bound - the reference itself is one
    # thread): oracle.c's pthread harness (orc_scan_topk_threads_pinned) - thread i, pinned to the i-th CPU of the process' affinity mask, loops
    # the reference's kernel inside the reference's top-k loop over its own range of the rows; a query's answer is the merge of the per-range
    # lists, checked once against the unsplit scan.  No interpreter in the timed loop.  Every thread's private copy holds its range several
    # times over (>= 64K rows: ~100 MB at 1.5 KB rows), so the timed scans stream from memory, not from a cache-resident 12 MB slice.
    host = host_description()
    out["host"] = {kk: vv for kk, vv in host.items() if kk != "affinity_cpus_list"}
    try:
        if ref is None:
            raise RuntimeError("needs oracle/_ref (the reference's own kernel)")
        cpus = host["affinity_cpus_list"]
        nthreads = len(cpus)
        quota = host.get("cgroup_cpu_quota_cores")
        if quota and quota < nthreads:                       # a CFS quota below the mask: more runnable threads than the quota only get throttled
            nthreads = max(1, int(quota))
        per = min(rows.shape[0] // nthreads, 16384)
        if per < 1024:
            raise RuntimeError("sample too small for %d threads" % nthreads)
        row_bytes = int(rows.strides[0])
        repeat = max(1, -(-65536 // per))
        budget = min(48 << 30, int(host.get("mem_available_bytes") or (8 << 30)) // 3)        # private copies: at most a third of what is free
        repeat = max(1, min(repeat, budget // max(1, nthreads * per * row_bytes)))
        big = rows[:per * nthreads]
        # fewer threads than CPUs in the mask (a quota): spread them evenly over the mask - over the sockets and core complexes, each with its
        # own memory channels and caches - instead of packing them onto the first few (16 threads on CPUs 0-15 of a 2 x 64-core host: 3.6 x one core)
        pick = [cpus[(i * len(cpus)) // nthreads] for i in range(nthreads)]
        rate, per, ids, d, cnt, pinned = ref.scan_topk_all_cores(metric, vt, queries[:8], big, k, nthreads, 4.0, cpus=pick, repeat=repeat)
        cand = sorted((float(dd), int(i) - 1 + t * per) for t in range(nthreads) for i, dd in zip(ids[t][:cnt[t]], d[t][:cnt[t]]))[:k]
        whole = work(big, q)
        merged_ok = [c[1] + 1 for c in cand] == np.asarray(whole[0]).tolist()[:k] or sorted(c[0] for c in cand) == sorted(np.asarray(whole[1]).tolist()[:k])
        out["all_cores"] = {"value": rate, "unit": "vectors/s", "cores": nthreads, "threads_pinned": int(pinned),
                            "GB_per_s": rate * row_bytes / 1e9, "x_one_core": rate / out["value"] if out["value"] else None,
                            "rows_per_thread_per_timed_scan": int(per * repeat), "private_copy_MB_per_thread": per * repeat * row_bytes / 1e6,
                            "rows": int(per * nthreads), "cpus_used": pick if nthreads <= 32 else "%d CPUs, every %dth of the mask" % (nthreads, max(1, len(cpus) // nthreads)),
                            "merged_lists_equal_the_unsplit_scan": bool(merged_ok),
                            "limited_by": ("cgroup cpu.max quota of %.1f cores (affinity mask: %d CPUs)" % (quota, len(cpus))) if (quota and quota < len(cpus))
                                          else "the affinity mask (%d of the host's %d logical CPUs)" % (len(cpus), os.cpu_count() or 0),
                            "note": "a row split of ONE %d x %d matrix (%s) over %d pinned pthreads (oracle/oracle.c orc_scan_topk_threads_pinned): thread i "
                                    "loops the reference's single-threaded kernel + top-k over its range (P = %d rows, held %d x in memory its own CPU "
                                    "touched first), 4 s; a query's answer = the merge of the %d lists (done once, untimed)"
                                    % (per * nthreads, dim, "rows of the GPU's corpus" if same_inputs else "numpy sample", nthreads, per, repeat, nthreads)}
    except Exception as e:
        out["all_cores"] = {"value": None, "note": "unavailable: %r" % (e,)}
    return out


def run_batched(args, pkg, torch, corpus, workload, n_rows, dim, metric, k, desc, dist=None, shard=None, n_gpus=1, rank=0, share=False):
    """config #5: each step = one batch of queries through the batched scan (host queries in, host (position,
    distance) lists out).  The dominant kernel is MFMA-bound: flops = 2 * Q * N * D per launch.
    N > 1: every rank scans its own row-range shard with the same batch, ONE all_gather of nq x k keys per rank
    (160 KB at 1024 x 20), rank 0 merges every query (shard.gather_and_merge_batch) - SURVEY 8e.
    Returns the result line (a dict) on rank 0, None elsewhere."""
    nq = args.batch
    rng = np.random.default_rng(44)
    steps, warmup = min(args.steps, 10), min(args.warmup, 2)
    quantized = corpus.vtype in (pkg.U8, pkg.I8)
    half = corpus.vtype == pkg.F16
    filt = workload in ("c5f", "c5l")
    q8 = workload == "c5q"
    if quantized:
        batches = [rng.integers(0, 256, (nq, dim)).astype(np.uint8) for _ in range(2)]
    elif half:
        batches = [rng.standard_normal((nq, dim), dtype=np.float32).astype(np.float16) for _ in range(2)]
    else:
        batches = [rng.standard_normal((nq, dim), dtype=np.float32) for _ in range(2)]
    use_dist = dist is not None
    offsets = [i * n_rows for i in range(n_gpus)]
    xdev = "cpu" if share else "cuda"                       # (share: ranks on one device exchange over gloo, host tensors)
    gathered = torch.empty((n_gpus, nq, k), dtype=torch.int64, device=xdev) if use_dist else None
    last = {}

    def step(i):
        if not use_dist:
            last["res"] = corpus.scan_topk_batch(metric, batches[i % 2], k)
            return
        keys, _ = corpus.scan_topk_batch_keys(metric, batches[i % 2], k)
        local = torch.from_numpy(keys.view(np.int64)).to(xdev)
        res = shard.gather_and_merge_batch(pkg, dist, local, gathered, offsets, k, dst=0)
        if res is not None:
            last["res"] = res

    for i in range(warmup):
        step(i)
    corpus.set_profiling(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_launch, kern_ms, _ = corpus.profile_mean_ms()
    # priced on the rate of the instruction the batch really ran on: the int8 filter (path 7) is the default for f32 / f16 / bf16 corpora
    # of this size, whatever the workload's name says
    q8 = q8 or corpus.last_batch_path() == 7
    peak = I8_MFMA_PEAK_TOPS if (quantized or q8) else (F16_MFMA_PEAK_TF if (half or filt) else F32_MFMA_PEAK_TF)
    single = None
    if workload == "c5l" and not use_dist:               # what the batch replaces: one scan per query (the plain kernel; HBM-bound)
        path = corpus.last_batch_path()
        corpus.set_profiling(False)
        for i in range(3):                               # (the first scans make the corpus' shadow copy for the filter scan)
            corpus.scan_topk(metric, batches[1][i], k)
        t1 = time.perf_counter()
        for i in range(8):
            corpus.scan_topk(metric, batches[0][i], k)
        one_ms = (time.perf_counter() - t1) / 8 * 1e3
        single = {"batch_path": path, "one_scan_per_query_ms": one_ms, "batch_ms_per_query": elapsed / steps * 1e3 / nq,
                  "speedup_over_single_scans": one_ms / (elapsed / steps * 1e3 / nq)}
    flops = 2.0 * nq * n_rows * dim
    tf = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    run_batched.last_result = last.get("res")
    if rank != 0:
        return None
    line = {
        "metric": "vectors scanned/sec (query x vector pairs), batched %s" % ("quantized cosine top-20 over Nx768 u8" if quantized else
                                                                              ("dot top-20 over Nx384 f16" if half else "dot top-20 over Nx%d f32" % dim)),
        "value": nq * n_rows * n_gpus * steps / elapsed, "unit": "vectors/s", "n_gpus": n_gpus, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8" if quantized else ("f16" if half else "f32"), "data": "synthetic",
        "config": {"workload": desc, "rows_per_gpu": n_rows, "dim": dim, "k": k, "queries_per_batch": nq,
                   "sharding": "row-range shard per GPU, RCCL all_gather of nq x k candidate keys per rank" if n_gpus > 1 else "single shard",
                   "backend": pkg.backend_name()},
        "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TOP/s" if (quantized or q8) else "TFLOP/s",
                     "frac": tf / peak, "traffic": None,
                     "kernel": ("vg_batch_q8_kernel<%d> + vg_batch_hx_kernel" % ((dim + 31) // 32)) if q8 else ("vg_batch_i8_kernel<%d>" % ((dim + 31) // 32)) if quantized else
                               (("vg_batch_h_kernel<%d>" % ((dim + 15) // 16)) if (half or filt) else ("vg_batch_kernel<%d>" % ((dim + 7) // 8))),
                     "kernel_ms": kern_ms, "launches_timed": n_launch, "flops_per_launch": flops,
                     "note": "kernel_ms = pre-pass + main pass + merges of one batch on one shard" +
                             ("; query images + staged filter / exact-evaluation / merge launches of one batch; peak = the int8 MFMA rate the filter runs at" if q8
                              else ("; peak = the bf16 MFMA rate the filter runs at" if filt else "")),
                     "batch_path": corpus.last_batch_path()}}
    if workload == "c5l" and q8:
        line["roofline"]["kernel"] = "vg_batch_q8_kernel<16 k-steps x %d K-parts, 8 wavefronts x 32 queries> + vg_batch_hx_kernel" % (((dim + 31) // 32 + 15) // 16)
    elif workload == "c5l":
        line["roofline"]["kernel"] = "vg_batch_hl_kernel<%d k-steps per wavefront> + vg_batch_hx_kernel" % (((dim * 2 + 31) // 32 + 3) // 4)
    # HBM bytes per BATCH from the PMC pass over the same batches (tools/pmc_batch.sh --json: FETCH_SIZE summed over every launch of one
    # vg_scan_topk_batch call), while the batch kernels' sources are the ones that pass ran on
    tb, tsrc = batch_traffic(batch_traffic_entry(corpus.last_batch_path(), "u8" if quantized else ("f16" if half else "f32"), metric, nq, dim, n_rows))
    line["roofline"]["traffic"] = tb
    if tsrc:
        line["roofline"]["traffic_source"] = tsrc
    if single is not None:
        line["against_single_scans"] = single
    return line


def sql_latency(ext_path, rows, queries, k, warmup, steps):
    """p50 / mean seconds of `SELECT rowid, distance FROM vector_full_scan('t','v',?,k)` through `ext_path`"""
    import sqlite3
    db = sqlite3.connect(":memory:", isolation_level=None)
    db.enable_load_extension(True)
    db.load_extension(ext_path)
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(rows.shape[0])])
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % rows.shape[1])
    backend = db.execute("SELECT vector_backend()").fetchone()[0]
    sql = "SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, %d)" % k
    res = None
    for i in range(warmup):
        res = db.execute(sql, (queries[i].tobytes(),)).fetchall()
    lat = []
    t0 = time.perf_counter()
    for i in range(steps):
        ts = time.perf_counter()
        res = db.execute(sql, (queries[warmup + i].tobytes(),)).fetchall()
        lat.append(time.perf_counter() - ts)
    elapsed = time.perf_counter() - t0
    res = db.execute(sql, (queries[0].tobytes(),)).fetchall()          # same query for every build: result check
    db.close()
    return elapsed, float(np.median(lat)), backend, res


def bench_sql(args, pkg, torch):
    """config #1: what a user of the reference types, unchanged, with this repo's vector.so loaded instead."""
    vt, np_dtype, dim, metric, desc = WORKLOADS["c1"]
    n_rows = args.rows if args.rows else 10_000
    k, steps, warmup = args.k, args.steps, args.warmup
    rng = np.random.default_rng(42)
    rows = rng.standard_normal((n_rows, dim), dtype=np.float32)
    queries = np.random.default_rng(43).standard_normal((steps + warmup, dim), dtype=np.float32)
    elapsed, p50, backend, res = sql_latency(pkg.EXT_PATH[:-3], rows, queries, k, warmup, steps)
    # kernel-level roofline of the same shape, HIP events around the scan kernel (the extension's own corpus is private)
    corpus = pkg.Corpus(vt, dim, capacity=n_rows)
    corpus.append(rows)
    for i in range(warmup):
        corpus.scan_topk(metric, queries[i], k)
    corpus.set_profiling(True)
    for i in range(steps):
        corpus.scan_topk(metric, queries[warmup + i], k)
    n_launch, scan_ms, merge_ms = corpus.profile_mean_ms()
    algo_bytes = n_rows * dim * 4
    achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    out = {
        "metric": "vectors scanned/sec, L2 top-20 over Nx384 f32 (SQL level)",
        "value": n_rows * steps / elapsed, "unit": "vectors/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "p50_query_latency_ms": p50 * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "rows_per_gpu": n_rows, "dim": dim, "k": k, "backend": backend},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": corpus.kernel_name(metric),
                     "kernel_ms": scan_ms, "merge_kernel_ms": merge_ms, "launches_timed": n_launch,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "a 15 MB scan is launch-latency bound, not HBM bound; see the c2 line for the roofline"},
    }
    corpus.close()
    if not args.no_cpu_baseline:
        try:
            from oracle import orc
            base = {}
            for which in ("cpu", "avx2"):
                ref = orc.ref_extension_path(which)
                if ref:
                    el, rp50, rbackend, rres = sql_latency(ref, rows, queries, k, min(warmup, 2), min(steps, 20))
                    base[which] = {"value": n_rows * min(steps, 20) / el, "p50_query_latency_ms": rp50 * 1e3, "backend": rbackend,
                                   "same_rowids_as_gpu": [r[0] for r in rres] == [r[0] for r in res]}
            if base:
                best = base.get("cpu") or base["avx2"]
                out["cpu_baseline"] = {"value": best["value"], "unit": "vectors/s", "cores": 1, "kind": "reference",
                                       "sample": "the same SQL through the reference's own vector.so built by oracle/Makefile "
                                                 "(stock flags = what its Makefile ships; 'avx2' = same sources with -mavx2)",
                                       "builds": base}
            else:
                out["cpu_baseline"] = {"value": None, "unit": "vectors/s", "cores": 0, "kind": "reference",
                                       "sample": "oracle/_ref/*/vector.so not built"}
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "vectors/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
    print(json.dumps(out))


def bench_sql_dropin(args, pkg, torch):
    """--workload sql: what the drop-in costs THROUGH SQL at a non-toy size (VERDICT r3 missing #3).  A file database with
         t384 : N x 384 f32 rows                 -> vector_full_scan (the first scan stages the table into HBM)
         t768 : N x 768 f32 rows, vector_quantize -> uint8, vector_quantize_preload -> vector_quantize_scan
       the same statements through this repo's vector.so and through the reference's own (oracle/_ref/avx2/vector.so, built by
       oracle/Makefile), one connection each, in this run.  Reported per leg: cold first scan (staging included) split into the
       staging loop (sqlite3_step + BLOB copy) / the engine's append calls (pinned copy + host-link back pressure) / the rest
       (derived passes, first launches), rows per second staged, warm p50, the reference's per-query time (it re-reads the table
       every query, sqlite-vector.c:2071-2113) and the number of queries after which the staging pass has paid for itself."""
    import sqlite3
    import tempfile
    from oracle import orc
    n = args.rows if args.rows else 1_000_000
    k = args.k
    tmp = tempfile.mkdtemp(prefix="vgsql_")
    path = os.path.join(tmp, "bench.db")
    rng = np.random.default_rng(42)

    def connect(ext):
        db = sqlite3.connect(path, isolation_level=None)
        db.enable_load_extension(True)
        db.load_extension(ext)
        return db

    t0 = time.perf_counter()
    db = sqlite3.connect(path, isolation_level=None)
    db.execute("PRAGMA journal_mode=OFF")
    db.execute("PRAGMA synchronous=OFF")
    for name, dim, gen in (("t384", 384, lambda m: rng.standard_normal((m, 384), dtype=np.float32)),
                           ("t768", 768, lambda m: rng.random((m, 768), dtype=np.float32))):
        db.execute("CREATE TABLE %s (id INTEGER PRIMARY KEY, v BLOB)" % name)
        db.execute("BEGIN")
        for r0 in range(0, n, 50_000):
            blk = gen(min(50_000, n - r0))
            db.executemany("INSERT INTO %s(id, v) VALUES (?, ?)" % name, [(r0 + i + 1, blk[i].tobytes()) for i in range(blk.shape[0])])
        db.execute("COMMIT")
    db.close()
    build_s = time.perf_counter() - t0
    q384 = np.random.default_rng(43).standard_normal((40, 384), dtype=np.float32)
    q768 = np.random.default_rng(44).random((40, 768), dtype=np.float32)

    def leg(ext, is_gpu, table, dim, queries, quantized, n_warm):
        db = connect(ext)
        out = {"backend": db.execute("SELECT vector_backend()").fetchone()[0]}
        kind = "FLOAT32"
        db.execute("SELECT vector_init('%s', 'v', 'type=%s,dimension=%d,distance=%s')" % (table, kind, dim, "COSINE" if quantized else "L2"))
        stats0 = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0]) if is_gpu else None
        if quantized:
            ts = time.perf_counter()
            db.execute("SELECT vector_quantize('%s', 'v')" % table)
            out["vector_quantize_s"] = time.perf_counter() - ts
            ts = time.perf_counter()
            db.execute("SELECT vector_quantize_preload('%s', 'v')" % table)
            out["vector_quantize_preload_s"] = time.perf_counter() - ts
            sql = "SELECT rowid, distance FROM vector_quantize_scan('%s', 'v', ?, %d)" % (table, k)
        else:
            sql = "SELECT rowid, distance FROM vector_full_scan('%s', 'v', ?, %d)" % (table, k)
        ts = time.perf_counter()
        first = db.execute(sql, (queries[0].tobytes(),)).fetchall()
        out["first_scan_s"] = time.perf_counter() - ts
        if is_gpu:
            st = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])
            d = {kk: st[kk] - stats0[kk] for kk in st}
            out["staging"] = {"rows": d["rows_staged"], "seconds_in_staging_loops": d["seconds_staging"],
                              "of_which_in_engine_append_calls": d["seconds_in_engine_append"],
                              "of_which_sqlite3_step_and_blob_copy": d["seconds_staging"] - d["seconds_in_engine_append"],
                              "count_star_s": d.get("seconds_count_star"), "hbm_reserve_s": d.get("seconds_hbm_reserve"),
                              "parallel_reader_passes": d.get("parallel_reader_passes"),
                              "rows_per_s": d["rows_staged"] / d["seconds_staging"] if d["seconds_staging"] > 0 else None,
                              "GB_per_s": d["rows_staged"] * dim * (1 if quantized else 4) / d["seconds_staging"] / 1e9 if d["seconds_staging"] > 0 else None}
        lat = []
        for i in range(n_warm):
            ts = time.perf_counter()
            db.execute(sql, (queries[1 + i].tobytes(),)).fetchall()
            lat.append(time.perf_counter() - ts)
        out["warm_p50_ms"] = float(np.median(lat)) * 1e3
        out["warm_queries"] = n_warm
        out["first_rowids"] = [r[0] for r in first]
        db.close()
        return out

    # what a process pays ONCE, whatever the table: the engine's first use (HIP context, the library's code objects, first launches)
    db0 = connect(pkg.EXT_PATH[:-3])
    db0.execute("CREATE TABLE t0 (id INTEGER PRIMARY KEY, v BLOB)")
    db0.executemany("INSERT INTO t0(id, v) VALUES (?, ?)", [(i + 1, np.full(8, i, np.float32).tobytes()) for i in range(64)])
    db0.execute("SELECT vector_init('t0', 'v', 'type=FLOAT32,dimension=8,distance=L2')")
    ts = time.perf_counter()
    db0.execute("SELECT rowid FROM vector_full_scan('t0', 'v', ?, 3)", (np.zeros(8, np.float32).tobytes(),)).fetchall()
    engine_first_use_s = time.perf_counter() - ts
    db0.execute("DROP TABLE t0")
    db0.close()
    res = {"rows": n, "k": k, "db_file_GB": os.path.getsize(path) / 1e9, "db_build_s_untimed": build_s,
           "engine_first_use_s_once_per_process": engine_first_use_s, "legs": {}}
    ref_ext = orc.ref_extension_path("avx2")
    for name, table, dim, queries, quantized in (("full_scan_f32_384", "t384", 384, q384, False), ("quantize_scan_u8_768", "t768", 768, q768, True)):
        g_leg = leg(pkg.EXT_PATH[:-3], True, table, dim, queries, quantized, 30)
        entry = {"gpu": g_leg}
        if ref_ext and not args.no_cpu_baseline:
            if quantized:                                  # the reference quantizes into the same shadow table: start from a clean one
                dbc = connect(pkg.EXT_PATH[:-3])
                dbc.execute("SELECT vector_init('%s', 'v', 'type=FLOAT32,dimension=%d,distance=COSINE')" % (table, dim))
                dbc.execute("SELECT vector_quantize_cleanup('%s', 'v')" % table)
                dbc.close()
            r_leg = leg(ref_ext, False, table, dim, queries, quantized, 4)
            entry["reference_avx2_one_core"] = r_leg
            entry["same_rowids_first_query"] = r_leg["first_rowids"] == g_leg["first_rowids"]
            per_ref = r_leg["warm_p50_ms"] / 1e3
            cold_extra = g_leg["first_scan_s"] + g_leg.get("vector_quantize_preload_s", 0.0) - g_leg["warm_p50_ms"] / 1e3
            ref_extra = r_leg.get("vector_quantize_preload_s", 0.0)
            gain = per_ref - g_leg["warm_p50_ms"] / 1e3
            entry["break_even_queries"] = (cold_extra - ref_extra) / gain if gain > 0 else None
            entry["warm_speedup"] = per_ref / (g_leg["warm_p50_ms"] / 1e3)
        for l in entry.values():
            if isinstance(l, dict):
                l.pop("first_rowids", None)
        res["legs"][name] = entry
    try:
        os.remove(path)
        os.rmdir(tmp)
    except OSError:
        pass
    g384 = res["legs"]["full_scan_f32_384"]["gpu"]
    out = {"metric": "vectors scanned/sec through SQL, warm (vector_full_scan over a staged Nx384 f32 table)", "value": n / (g384["warm_p50_ms"] / 1e3),
           "unit": "vectors/s", "n_gpus": 1, "steps": g384["warm_queries"], "warmup": 1, "ms_per_step": g384["warm_p50_ms"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "the drop-in through SQL: %dx384 f32 vector_full_scan + %dx768 -> uint8 vector_quantize / preload / vector_quantize_scan in a file database" % (n, n)},
           "sql": res}
    print(json.dumps(out))
    return 0


class SingleQueryRunner:
    """one resident shard + the per-step plumbing of a single-query scan (query upload -> scan + candidate reduction ->
    [RCCL gather] -> k keys to the host -> merge); run() times K steps the way the contract prescribes"""

    def __init__(self, pkg, torch, dist, shard, corpus, vt, dim, metric, k, n_rows, n_gpus, queries, share=False):
        self.pkg, self.torch, self.dist, self.shard, self.corpus = pkg, torch, dist, shard, corpus
        self.share = share                            # ranks share a device: the exchange runs over gloo on host tensors
        self.metric, self.k, self.n_rows, self.n_gpus = metric, k, n_rows, n_gpus
        es = pkg.TYPE_SIZE[vt]
        nq = queries.shape[0]
        # a real (non-null) stream: handle 0 would mean "use the corpus' own stream" to the C-ABI
        self.stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)
        qpad = ((dim * es + 15) // 16) * 16
        self.d_query = torch.zeros(qpad, dtype=torch.uint8, device="cuda")
        # every query of the run zero-padded in ONE pinned host tensor: a step uploads its row (the upload stays in the
        # timed region, the numpy -> torch conversion does not have to)
        h = torch.zeros((nq, qpad), dtype=torch.uint8)
        h[:, : dim * es] = torch.from_numpy(queries.view(np.uint8).reshape(nq, dim * es))
        self.h_queries = h.pin_memory()
        self.d_keys = torch.empty(64, dtype=torch.int64, device="cuda")
        self.h_keys = torch.empty((n_gpus, 64), dtype=torch.int64).pin_memory()
        self.d_all = torch.empty((n_gpus, 64), dtype=torch.int64, device="cpu" if share else "cuda") if dist is not None else None
        self.h_local = torch.empty(64, dtype=torch.int64).pin_memory() if share else None
        self.offsets = [i * n_rows for i in range(n_gpus)]
        self.last = {}

    def step(self, i):
        pkg, stream = self.pkg, self.stream
        self.d_query.copy_(self.h_queries[i], non_blocking=True)
        self.corpus.scan_topk_device(self.metric, self.d_query.data_ptr(), self.k, self.d_keys.data_ptr(), stream.cuda_stream)
        if self.dist is not None and self.share:
            self.h_local.copy_(self.d_keys, non_blocking=True)
            stream.synchronize()
            res = self.shard.gather_and_merge(pkg, self.dist, self.h_local, self.d_all, self.offsets, self.k, dst=0)
            if res is not None:
                self.last["pos"], self.last["dist"] = res
        elif self.dist is not None:
            # the path's only exchange: 64 keys per rank, one RCCL all_gather, rank 0 merges (shard.py)
            res = self.shard.gather_and_merge(pkg, self.dist, self.d_keys, self.d_all, self.offsets, self.k, dst=0,
                                              host_buf=self.h_keys, sync=stream.synchronize)
            if res is not None:
                self.last["pos"], self.last["dist"] = res
            else:
                stream.synchronize()      # lockstep with rank 0: the pinned query buffer is reused next step
        else:
            self.h_keys[0].copy_(self.d_keys, non_blocking=True)
            stream.synchronize()
            self.last["pos"], self.last["dist"] = pkg.merge_keys(self.h_keys.numpy().view(np.uint64), self.offsets, self.k)

    def run(self, warmup, steps):
        """W untimed steps, then exactly K steps between barrier + synchronize; returns (elapsed max over ranks, latencies)"""
        torch, dist = self.torch, self.dist
        for i in range(warmup):
            self.step(i)
        self.corpus.set_profiling(True)               # reset the event ring: only timed steps are averaged
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        lat = []
        t0 = time.perf_counter()
        for i in range(steps):
            ts = time.perf_counter()
            self.step(warmup + i)
            lat.append(time.perf_counter() - ts)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if self.share else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, lat


KERNEL_SOURCES = ("vg_scan.h", "vg_accum.h", "vg_half.h", "vg_device.h", "vg_lists.h", "vg_scan_filter.h", "vg_scan_filter_n4.h",
                  "vg_api.hip", "vg_filter.hip")


def kernel_source_hash():
    """sha256 over the sources the single-query scan kernels are compiled from (what a PMC pass has to be re-run for)"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "sqlite-vector_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


BATCH_KERNEL_SOURCES = ("vg_batch.hip", "vg_batch_h.hip", "vg_batch_hl.hip", "vg_batch_q8.hip", "vg_batch_i8.hip", "vg_batch_h_defs.h", "vg_batch_common.h",
                        "vg_batch_api.hip", "vg_accum.h", "vg_half.h")
METRIC_NAMES = {1: "l2", 2: "l2sq", 3: "cosine", 4: "dot", 5: "l1"}


def batch_traffic_entry(path, dtype, metric, nq, dim, n_rows):
    """key of a batched workload in profiles/pmc_traffic.json: batch path (vg_corpus_last_batch_path), element type, metric, batch size, row length @ rows"""
    return "batch_path%d_%s_%s_%dq_%d@%d" % (path, dtype, METRIC_NAMES.get(metric, str(metric)), nq, dim, n_rows)


def batch_kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for name in BATCH_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "sqlite-vector_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def batch_traffic(entry):
    """HBM bytes per BATCH (all launches of one vg_scan_topk_batch call) from the PMC pass recorded in profiles/pmc_traffic.json under
    `entry` - only while the batch kernels' sources are the ones that pass was made on"""
    try:
        now = batch_kernel_source_hash()
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(entry)
        if not ent:
            return None, None
        if ent.get("kernel_source_hash") != now:
            return None, "stale: measured on batch kernel sources %s, this build is %s" % (ent.get("kernel_source_hash"), now)
        return ent["bytes_per_batch"], ent["source"]
    except Exception:
        return None, None


def pmc_traffic(kernel_name, n_rows):
    """HBM bytes per launch measured by the PMC pass committed under profiles/ (same kernel, same N) - only when that pass was
    made on THESE kernel sources (pmc_traffic.json records the source hash of its build; tools/measure.sh refreshes it).
    Returns (bytes or None, source / reason)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tab = json.load(f)
        ent = tab.get("%s@%d" % (kernel_name, n_rows))
        if ent:
            have, now = ent.get("kernel_source_hash"), kernel_source_hash()
            if have != now:
                return None, "stale: profiles/pmc_traffic.json entry was measured on kernel sources %s, this build is %s - re-run tools/measure.sh pmc" % (have, now)
            return ent["bytes_per_launch"], ent["source"]
    except Exception:
        pass
    return None, None


def single_query_line(args, pkg, runner, corpus, workload, vt, dim, metric, k, n_rows, n_gpus, desc):
    """time the scan as it is currently switched on `corpus` and price it on SURVEY 8(d)'s algorithmic bytes"""
    es = pkg.TYPE_SIZE[vt]
    elapsed, lat = runner.run(args.warmup, args.steps)
    n_launch, scan_ms, merge_ms, prepass_ms = corpus.profile_mean_ms_ex()
    kname = corpus.kernel_name(metric)
    algo_bytes = n_rows * dim * es                              # per launch (one shard): corpus read once
    achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic, source = pmc_traffic(kname, n_rows)
    out = {
        "metric": "vectors scanned/sec, L2 top-20 over Nx384 f32" if workload == "c2" else
                  "vectors scanned/sec, quantized cosine top-20 over Nx768 u8",
        "value": n_rows * n_gpus * args.steps / elapsed,
        "unit": "vectors/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "p50_query_latency_ms": float(np.median(lat) * 1e3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if vt == pkg.F32 else "u8", "data": "synthetic",
        "config": {"workload": desc, "rows_per_gpu": n_rows, "dim": dim, "k": k,
                   "sharding": "row-range shard per GPU, RCCL all_gather of 64 candidate keys per rank" if n_gpus > 1 else "single shard",
                   "backend": pkg.backend_name()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": kname, "kernel_ms": scan_ms, "merge_kernel_ms": merge_ms,
                     "launches_timed": n_launch, "algorithmic_bytes_per_launch": algo_bytes},
    }
    if source:
        out["roofline"]["traffic_source"] = source
    if n_gpus == 1:
        out["caller_view"] = caller_view(args, corpus, runner, metric, scan_ms)
    return out, prepass_ms


def caller_view(args, corpus, runner, metric, kernel_ms):
    """what a caller of the product API pays per query: vg_scan_topk (host query in, host rowids + distances out) with the
    profiling events OFF - the timed steps above carry four event records per query, which is what kernel_ms is measured with"""
    try:
        nq = runner.h_queries.shape[0]
        qs = [runner.h_queries[i].numpy() for i in range(nq)]
        corpus.set_profiling(False)
        for i in range(min(5, nq)):
            corpus.scan_topk(metric, qs[i], runner.k)
        t0 = time.perf_counter()
        for i in range(args.steps):
            corpus.scan_topk(metric, qs[(args.warmup + i) % nq], runner.k)
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        corpus.set_profiling(True)
        return {"ms_per_query": ms, "outside_kernel_us": (ms - kernel_ms) * 1e3,
                "what": "vg_scan_topk end to end (host query in, host top-k out), profiling events off, the same %d queries; "
                        "outside_kernel_us = this minus kernel_ms above" % args.steps}
    except Exception as e:
        return {"error": repr(e)}


def filter_scan_object(args, pkg, corpus, runner, metric, vt, dim, n_rows, plain_last):
    """the SAME queries over the SAME corpus through the lower-bound filter scan (the product's default path for a corpus of this
    size): priced on the bytes it streams, never under the line's dtype / roofline.frac"""
    try:
        corpus.set_scan_filter(1)
        if n_rows < (1 << 20):
            os.environ.setdefault("VG_SCAN_FILTER_MIN_MB", "0")    # (a reduced --rows run: the filter regardless of the size rule)
        runner.step(0)                                             # builds the shadow copy + norms (not timed)
        corpus.filter_exact_evals()
        felapsed, flat = runner.run(args.warmup, args.steps)
        fn_launch, fscan_ms, fmerge_ms, fpre_ms = corpus.profile_mean_ms_ex()
        evals = corpus.filter_exact_evals()
        fname = corpus.kernel_name(metric)
        es = pkg.TYPE_SIZE[vt]
        if "_n4_" in fname:        # uint8 / int8: the high-nibble shadow row + (sum x^2, sum of low nibbles, their centred norm)
            kind, per_row = "high nibbles (4 bit)", ((dim + 31) // 32) * 16 + 16
        elif "_q8_" in fname:      # the int8 shadow row + (scale, residual norm, cached f32 norm)
            kind, per_row = "int8", ((dim + 15) // 16) * 16 + 12
        elif vt == pkg.F32:        # the bf16 shadow row + the cached f32 norm
            kind, per_row = "bf16", ((dim * 2 + 15) // 16) * 16 + 4
        else:
            kind, per_row = "rows", ((dim * es + 15) // 16) * 16 + 4
        streamed = n_rows * per_row
        ftraffic, fsource = pmc_traffic(fname, n_rows)
        same = (list(runner.last["pos"]) == list(plain_last["pos"]) and
                np.array_equal(np.asarray(runner.last["dist"]), np.asarray(plain_last["dist"])))
        caller = caller_view(args, corpus, runner, metric, fscan_ms)
        return {
            "what": "the same %d queries through the filter scan: %s shadow copy as a lower-bound filter + exact re-evaluation of the "
                    "candidates with the plain kernel's arithmetic (same rowids and distance bits as the plain scan)" % (args.steps, kind),
            "value": n_rows * args.steps / felapsed, "unit": "vectors/s", "ms_per_step": felapsed / args.steps * 1e3,
            "p50_query_latency_ms": float(np.median(flat) * 1e3),
            "kernel": fname, "kernel_ms": fscan_ms, "prepass_ms": fpre_ms, "merge_kernel_ms": fmerge_ms, "launches_timed": fn_launch,
            "dtype_streamed": kind, "streamed_bytes_per_launch": streamed,
            "achieved_on_streamed_GBs": streamed / (fscan_ms * 1e-3) / 1e9 if fscan_ms > 0 else 0.0,
            "frac_on_streamed": streamed / (fscan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fscan_ms > 0 else 0.0,
            "traffic": ftraffic, "traffic_source": fsource,
            "exact_evaluations_per_query": evals / float(args.warmup + args.steps),
            "last_query_same_answer_as_plain_scan": bool(same),
            "extra_hbm_bytes": n_rows * per_row,
            "caller_view": caller,
        }
    except Exception as e:
        return {"error": repr(e)}


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


def bench_inprocess(args):
    """`--gpus N --inprocess`: config C4 the way the SQLite extension holds it - ONE process, vg_shards dealing the corpus over N
    devices block-cyclically, every query = N scans in flight + the candidate gather (host copies, then one grouped RCCL all-gather)
    + the host merge.  Fewer than N devices visible: the shards share device 0 (logical shards - a functional run, labelled)."""
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    n = args.gpus
    have = torch.cuda.device_count()
    shared = have < n
    devices = [0] * n if shared else list(range(n))
    vt, np_dtype, dim, metric, _ = WORKLOADS["c2"]
    per = args.rows if args.rows else (12_500_000 if not shared else 1_250_000)
    total = per * n
    sh = pkg.Shards(vt, dim, devices)
    sh.reserve(total)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    pinned = torch.empty((500_000, dim), dtype=torch.float32).pin_memory()
    for r0 in range(0, total, 500_000):
        nr = min(500_000, total - r0)
        t = torch.randn((nr, dim), generator=gen, device="cuda", dtype=torch.float32)
        pinned[:nr].copy_(t)
        torch.cuda.synchronize()
        sh.append(pinned[:nr].numpy())
        del t
    sh.set_scan_filter(0)
    lib = pkg.lib()
    handles = [lib.vg_shards_shard(sh.h, i) for i in range(n)]
    steps, warmup, k = args.steps, args.warmup, args.k
    qs = np.random.default_rng(43).standard_normal((steps + warmup, dim), dtype=np.float32)
    forms = {}
    first = None
    for form in ("host", "rccl"):
        sh.set_gather(form)
        for i in range(warmup):
            sh.scan_topk(metric, qs[i], k)
        import ctypes as C
        for h in handles:
            lib.vg_set_profiling(C.c_void_p(h), 1)
        before = sh.gather_stats()
        lat = []
        t0 = time.perf_counter()
        for i in range(steps):
            ts = time.perf_counter()
            ids, dist = sh.scan_topk(metric, qs[warmup + i], k)
            lat.append(time.perf_counter() - ts)
        elapsed = time.perf_counter() - t0
        after = sh.gather_stats()
        if first is None:
            first = (ids.tolist(), dist.tolist())
        per_dev = []
        for i, h in enumerate(handles):
            nl, a, b = C.c_int(0), C.c_float(0), C.c_float(0)
            lib.vg_profile_mean_ms(C.c_void_p(h), C.byref(nl), C.byref(a), C.byref(b))
            rows_i = lib.vg_corpus_rows(C.c_void_p(h))
            gb = rows_i * dim * 4 / 1e9
            per_dev.append({"device": devices[i], "rows": rows_i, "kernel_ms": a.value, "frac_of_8TBs": gb / a.value / 8.0 if a.value > 0 else None})
        served = "rccl" if after["rccl"] - before["rccl"] == steps else "host"
        forms[form] = {"ms_per_query": elapsed / steps * 1e3, "p50_ms": float(np.median(lat)) * 1e3, "vectors_per_s": total * steps / elapsed,
                       "gather_that_served": served, "same_answer_as_first_form": (ids.tolist(), dist.tolist()) == first or form == "host",
                       "per_device": per_dev}
    main_form = forms["host"]
    out = {"metric": "vectors scanned/sec + p50 query latency, L2 top-20 over Nx384 f32", "value": main_form["vectors_per_s"], "unit": "vectors/s",
           "n_gpus": n, "steps": steps, "warmup": warmup, "ms_per_step": main_form["ms_per_query"], "p50_query_latency_ms": main_form["p50_ms"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%gMx384 f32 L2 top-20 single-query, ONE process: vg_shards over %d %s (%gM rows each, block-cyclic deal), plain kernel"
                                  % (total / 1e6, n, "LOGICAL shards on one device (functional run, not a scaling measurement)" if shared else "devices", per / 1e6),
                      "rows_per_gpu": per, "dim": dim, "k": k, "sharding": "in-process vg_shards", "backend": pkg.backend_name()},
           "gather_forms": forms}
    print(json.dumps(out))
    sh.close()
    return 0


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
        got = torch.empty((dist.get_world_size(), 2), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(got, mine)
        if not share:
            torch.cuda.synchronize()
        pairs = got.cpu().tolist()
        rec["ranks_in_communicator"] = len(set(p[0] for p in pairs))
        rec["rank_devices"] = [p[1] for p in sorted(pairs)]
    except Exception as e:
        rec["ranks_in_communicator"] = "all_gather failed: %r" % (e,)
    return rec


def load_shard_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vg_shard", os.path.join(ROOT, "sqlite-vector_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    return shard


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
                                   "max_rel_vs_reference_kernel": g(c5, "int8_filter_batch", "last_batch_max_rel_difference_from_the_reference_kernel")},
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


def c3_queries(nq, dim):
    """queries of config C3: f32 U[0,1) quantized like the corpus"""
    return quantize_unit_uniform_np(np.random.default_rng(43).random((nq, dim), dtype=np.float32))


def quantize_unit_uniform_np(v):
    """the reference's uint8 quantizer (sqlite-vector.c:517-548) with the parameters a U[0,1) source gets: offset = min = 0,
    scale = 255 / (max - min) = 255: (uint8)(v * 255 + 0.5)"""
    return np.clip(np.floor(v * np.float32(255.0) + np.float32(0.5)), 0, 255).astype(np.uint8)


def also_c3(args, pkg, torch, shard, also_set, n_rows, k, nq, device_index):
    try:
        v3, t3, d3, m3, desc3 = WORKLOADS["c3"]
        c3 = make_shard(pkg, torch, v3, d3, n_rows, 42, device_index)
        q3 = c3_queries(nq, d3)
        c3.set_scan_filter(0)              # the line: the plain kernel on SURVEY 8(d)'s 7.68 GB; the nibble filter on its own below
        c3.set_tie_order(pkg.TIE_POSITION)
        r3 = SingleQueryRunner(pkg, torch, None, shard, c3, v3, d3, m3, k, n_rows, 1, q3)
        line, _ = single_query_line(args, pkg, r3, c3, "c3", v3, d3, m3, k, n_rows, 1, desc3)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(v3, t3, d3, m3, k, args.cpu_sample_rows, seconds=5.0, all_cores=False,
                                                rows=corpus_sample(pkg, torch, v3, d3, n_rows, 42, args.cpu_sample_rows), queries=q3[args.warmup:args.warmup + 8])
        # what the reference's result order costs (its rowids among equal distances; the default of the SQL surface for
        # quantized scans): the same host entry point (vg_scan_topk: host query in, host rowids out) in both orders
        tie = {}
        for name, mode in (("position", pkg.TIE_POSITION), ("reference", pkg.TIE_REFERENCE)):
            c3.set_tie_order(mode)
            c3.scan_topk(m3, q3[0], k)
            t0 = time.perf_counter()
            for i in range(20):
                c3.scan_topk(m3, q3[(1 + i) % nq], k)
            tie["ms_per_query_%s" % name] = (time.perf_counter() - t0) / 20 * 1e3
        tie["reference_over_position"] = tie["ms_per_query_reference"] / tie["ms_per_query_position"]
        if hasattr(c3, "tie_stats"):
            tie["reference_path_counters"] = c3.tie_stats()
        # ... and what a query WITH a tie costs: the same corpus under L1 - integer sums, so equal distances among the 21 best are
        # routine - where most queries go through the fused replay (prefix pass + the candidates the scan emitted + host replay)
        before = c3.tie_stats() if hasattr(c3, "tie_stats") else None
        for name, mode in (("position", pkg.TIE_POSITION), ("reference", pkg.TIE_REFERENCE)):
            c3.set_tie_order(mode)
            for i in range(12):                  # (untimed: the first tie also loads the emitting kernels' code object, ~ms, once per process)
                c3.scan_topk(5, q3[(30 + i) % nq], k)
            t0 = time.perf_counter()
            for i in range(20):
                c3.scan_topk(5, q3[(2 + i) % nq], k)
            tie["l1_ms_per_query_%s" % name] = (time.perf_counter() - t0) / 20 * 1e3
        tie["l1_reference_over_position"] = tie["l1_ms_per_query_reference"] / tie["l1_ms_per_query_position"]
        # ... and CHECKED: the last of those tie-heavy queries against the reference's own kernel + slot loop over the whole corpus in
        # scan order (oracle/_ref; rowids and distance bits at every rank) - the fused replay is timed above, this says it is right
        try:
            from oracle import orc
            if orc.have_ref() and not args.no_cpu_baseline:
                c3.set_tie_order(pkg.TIE_REFERENCE)
                answers = [(qi, c3.scan_topk(5, q3[qi], k)) for qi in [(2 + i) % nq for i in range(20)]]
                tied = [a for a in answers if np.any(np.diff(np.asarray(a[1][1], dtype=np.float32)) == 0)]       # equal distances inside the top k
                picks = ([tied[0]] if tied else []) + [answers[-1]]
                host = np.empty((n_rows, d3), dtype=np.uint8)
                for r0, t in shard_blocks(pkg, torch, v3, d3, n_rows, 42):
                    host[r0:r0 + t.shape[0]] = t.cpu().numpy()
                ref = orc.RefKernels("avx2")
                checked = []
                for qi, (got_ids, got_d) in picks:
                    t0 = time.perf_counter()
                    want_ids, want_d = ref.scan_topk(5, v3, q3[qi], host, k)
                    ref_s = time.perf_counter() - t0
                    same = (np.asarray(got_ids).tolist() == np.asarray(want_ids).tolist() and
                            np.array_equal(np.asarray(got_d, dtype=np.float32).view(np.uint32), np.asarray(want_d, dtype=np.float32).view(np.uint32)))
                    d32 = np.asarray(want_d, dtype=np.float32)
                    checked.append({"query": int(qi), "rowids_and_distance_bits": bool(same), "ties_among_the_%d" % k: int(np.sum(d32[1:] == d32[:-1])),
                                    "reference_scan_s": ref_s})
                    if not same:
                        raise SystemExit("bench.py: the reference-order answer of L1 query %d differs from the reference's own scan: %r vs %r" % (
                            qi, np.asarray(got_ids).tolist(), np.asarray(want_ids).tolist()))
                del host
                tie["l1_queries_checked_against_the_reference"] = {"queries_with_ties_inside_the_top_k": len(tied), "checked": checked}
        except SystemExit:
            raise
        except Exception as e:
            tie["l1_queries_checked_against_the_reference"] = {"error": repr(e)}
        if before is not None:
            after = c3.tie_stats()
            tie["l1_reference_path_counters"] = {kk: after[kk] - before[kk] for kk in after}
        c3.set_tie_order(pkg.TIE_POSITION)
        tie["what"] = ("vg_scan_topk end to end, top-%d, 20 queries each; reference = the same scan with one more list slot, the "
                       "reference's slot algorithm replayed on the host only for queries whose k+1 best distances hold a tie" % k)
        line["tie_order"] = tie
        if "filter" in also_set:
            # what the product does with this corpus by default: the high-nibble filter is PROBED (a 2M-row prefix) and kept
            # only if the data is selective under it - independent random bytes are not
            try:
                c3.set_scan_filter(-1)
                c3.filter_exact_evals()
                c3.scan_topk(m3, q3[0], k)                   # the probing scan
                probe_evals = c3.filter_exact_evals()
                for i in range(3):
                    c3.scan_topk(m3, q3[1 + i], k)
                line["nibble_filter_probe"] = {
                    "candidates_in_the_probed_prefix": probe_evals, "prefix_rows": min(n_rows, 1 << 21),
                    "kernel_after_the_probe": c3.kernel_name(m3),
                    "filter_in_use": bool(c3.kernel_name(m3).startswith("scan_filter")),
                }
                if line["nibble_filter_probe"]["filter_in_use"]:
                    c3.set_scan_filter(0)
                    r3.run(args.warmup, args.steps)           # (the plain answers of the same query sequence)
                    plain3 = dict(r3.last)
                    line["filter_scan"] = filter_scan_object(args, pkg, c3, r3, m3, v3, d3, n_rows, plain3)
            except Exception as e:
                line["nibble_filter_probe"] = {"error": repr(e)}
        c3.close()
        return line
    except Exception as e:
        return {"error": repr(e)}


def _time_scans(corpus, metric, qs, k, n):
    """(ms per scan as a caller sees it, scan kernel ms, pre-pass ms, kernel name) over n single scans (profiling events on: kernel time by HIP events)"""
    for i in range(3):
        corpus.scan_topk(metric, qs[i % len(qs)], k)
    corpus.set_profiling(True)
    t0 = time.perf_counter()
    for i in range(n):
        corpus.scan_topk(metric, qs[(3 + i) % len(qs)], k)
    ms = (time.perf_counter() - t0) / n * 1e3
    _, scan_ms, _, pre_ms = corpus.profile_mean_ms_ex()
    return ms, scan_ms, pre_ms, corpus.kernel_name(metric)


def also_clustered(args, pkg, torch, k, device_index):
    """`also.clustered`: the DEFAULT paths on data that is not iid - 10M x 384 f32 drawn from 4 096 Gaussian clusters and
    L2-normalised (tests/datagen.py: what a table of sentence embeddings looks like), cosine and dot, queries near cluster centres; the same
    corpus quantized to uint8 (the reference's formula over the corpus' own min / max: ~46 of the 256 levels are used); and an ADVERSARIAL
    corpus (every row within 1e-3 of every query: no bound separates anything - the selectivity guard must hand the query to the plain
    kernel).  Per leg: the plain kernel, the default single-query path (filter scan) with its exact evaluations per query, the default
    1024-query batch.  The cost of the filter paths is a property of the data: these figures stand next to the N(0,1) ones, not under them."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datagen as dgen
    out = {"what": "default paths on clustered unit-norm data (4096 clusters, within-cluster noise norm %.1f, queries at noise %.1f) and on an adversarial corpus; "
                   "generator: tests/datagen.py clustered_block / adversarial_block" % (dgen.CLUSTER_NOISE, dgen.QUERY_NOISE), "legs": {}}
    n_rows, dim, blk, nq_batch = (args.rows or 10_000_000), 384, 500_000, args.batch
    dev = "cuda:%d" % device_index
    try:
        centres = dgen.clustered_centres(torch, 42, dim, device=dev)
        qs = dgen.clustered_queries(torch, centres, 42, max(nq_batch, 32))
        c = pkg.Corpus(pkg.F32, dim, capacity=n_rows, device=device_index)
        lo, hi = float("inf"), float("-inf")
        for b in range(n_rows // blk):
            t = dgen.clustered_block(torch, centres, 42, b, blk)
            lo, hi = min(lo, float(t.min())), max(hi, float(t.max()))
            torch.cuda.synchronize()
            c.append_device(t.data_ptr(), blk, dim * 4)
            del t
        for mname, metric in (("cosine", pkg.COSINE), ("dot", pkg.DOT)):
            leg = {}
            c.set_scan_filter(0)
            ms, scan_ms, _, kn = _time_scans(c, metric, qs, k, 20)
            leg["plain"] = {"ms_per_step": ms, "kernel": kn, "kernel_ms": scan_ms, "frac_of_hbm_peak": n_rows * dim * 4 / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if scan_ms > 0 else None}
            plain_ans = c.scan_topk(metric, qs[0], k)
            c.set_scan_filter(-1)
            c.scan_topk(metric, qs[0], k)
            c.filter_exact_evals()
            ms, scan_ms, pre_ms, kn = _time_scans(c, metric, qs, k, 20)
            ev = c.filter_exact_evals() / 23.0
            dflt_ans = c.scan_topk(metric, qs[0], k)
            leg["default_single"] = {"ms_per_step": ms, "kernel": kn, "kernel_ms": scan_ms, "prepass_ms": pre_ms, "exact_evaluations_per_query": ev,
                                     "same_answer_as_plain_scan": bool(np.array_equal(plain_ans[0], dflt_ans[0]) and np.array_equal(plain_ans[1], dflt_ans[1]))}
            for i in range(2):
                c.scan_topk_batch(metric, qs[:nq_batch], k)
            c.batch_filter_exact_evals()
            t0 = time.perf_counter()
            for i in range(5):
                c.scan_topk_batch(metric, qs[:nq_batch], k)
            bms = (time.perf_counter() - t0) / 5 * 1e3
            leg["default_batch_%d" % nq_batch] = {"ms_per_step": bms, "batch_path": c.last_batch_path(), "exact_evaluations_per_query": c.batch_filter_exact_evals() / float(5 * nq_batch),
                                                  "frac_of_int8_peak": 2.0 * nq_batch * n_rows * dim / (bms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS if c.last_batch_path() == 7 else None}
            out["legs"]["f32_%s" % mname] = leg
        c.close()
        del c
        torch.cuda.empty_cache()
        # ---- the same clusters at 768 elements, quantized to uint8 with the reference's formula (sqlite-vector.c:1258-1268: scale = 255 / (max - min), offset = min)
        dim8 = 768
        centres8 = dgen.clustered_centres(torch, 43, dim8, device=dev)
        lo8, hi8 = float("inf"), float("-inf")
        for b in range(0, n_rows // blk, 5):                               # (min / max over a fifth of the blocks: the quantizer's parameters)
            t = dgen.clustered_block(torch, centres8, 43, b, blk)
            lo8, hi8 = min(lo8, float(t.min())), max(hi8, float(t.max()))
            del t
        scale8 = 255.0 / (hi8 - lo8)

        def q8(t):
            return torch.clamp(torch.floor((t - lo8) * scale8 + 0.5), 0, 255).to(torch.uint8)
        c8 = pkg.Corpus(pkg.U8, dim8, capacity=n_rows, device=device_index)
        levels = 0
        for b in range(n_rows // blk):
            t = q8(dgen.clustered_block(torch, centres8, 43, b, blk))
            if b == 0:
                levels = int(torch.unique(t).numel())
            torch.cuda.synchronize()
            c8.append_device(t.data_ptr(), blk, dim8)
            del t
        qs8 = q8(dgen.clustered_block(torch, centres8, 43 + 977, 0, max(nq_batch, 32), noise=dgen.QUERY_NOISE)).cpu().numpy()
        leg = {"uint8_levels_in_use": levels}
        c8.set_scan_filter(0)
        ms, scan_ms, _, kn = _time_scans(c8, pkg.COSINE, qs8, k, 20)
        leg["plain"] = {"ms_per_step": ms, "kernel": kn, "kernel_ms": scan_ms, "frac_of_hbm_peak": n_rows * dim8 / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if scan_ms > 0 else None}
        c8.set_scan_filter(-1)
        c8.scan_topk(pkg.COSINE, qs8[0], k)
        c8.filter_exact_evals()
        ms, scan_ms, pre_ms, kn = _time_scans(c8, pkg.COSINE, qs8, k, 20)
        leg["default_single"] = {"ms_per_step": ms, "kernel": kn, "kernel_ms": scan_ms, "prepass_ms": pre_ms, "exact_evaluations_per_query": c8.filter_exact_evals() / 23.0,
                                 "nibble_filter_in_use": "_n4_" in kn and c8.filter_guard_cooldown() == 0,
                                 "guard_sent_the_queries_to_the_plain_kernel": c8.filter_guard_cooldown() > 0}
        for i in range(2):
            c8.scan_topk_batch(pkg.COSINE, qs8[:nq_batch], k)
        t0 = time.perf_counter()
        for i in range(3):
            c8.scan_topk_batch(pkg.COSINE, qs8[:nq_batch], k)
        bms = (time.perf_counter() - t0) / 3 * 1e3
        leg["default_batch_%d" % nq_batch] = {"ms_per_step": bms, "batch_path": c8.last_batch_path(),
                                              "frac_of_int8_peak": 2.0 * nq_batch * n_rows * dim8 / (bms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS}
        out["legs"]["u8_768_cosine"] = leg
        c8.close()
        del c8
        torch.cuda.empty_cache()
        # ---- adversarial: 2M rows all within ~1e-3 of each other and of the queries
        na = min(n_rows, 2_000_000)
        ca = pkg.Corpus(pkg.F32, dim, capacity=na, device=device_index)
        for b in range(na // blk):
            t = dgen.adversarial_block(torch, 7, b, blk, dim, device=dev)
            torch.cuda.synchronize()
            ca.append_device(t.data_ptr(), blk, dim * 4)
            del t
        qa = dgen.adversarial_block(torch, 7, 9999, 32, dim, device=dev).cpu().numpy()
        leg = {"rows": na}
        ca.set_scan_filter(0)
        pms, pscan, _, pkn = _time_scans(ca, pkg.COSINE, qa, k, 20)
        plain_ans = ca.scan_topk(pkg.COSINE, qa[0], k)
        ca.set_scan_filter(-1)
        for i in range(4):                                                 # (the guard needs a few queries to see that the bound does not separate)
            ca.scan_topk(pkg.COSINE, qa[i], k)
        dms, dscan, _, dkn = _time_scans(ca, pkg.COSINE, qa, k, 20)
        dflt_ans = ca.scan_topk(pkg.COSINE, qa[0], k)
        leg.update({"plain_ms_per_step": pms, "plain_kernel": pkn, "default_ms_per_step": dms,
                    "default_over_plain": dms / pms if pms > 0 else None, "guard_handed_the_queries_to_the_plain_kernel": ca.filter_guard_cooldown() > 0,
                    "same_answer_as_plain_scan": bool(np.array_equal(plain_ans[0], dflt_ans[0]) and np.array_equal(plain_ans[1], dflt_ans[1]))})
        out["legs"]["adversarial_f32_cosine"] = leg
        ca.close()
        torch.cuda.empty_cache()
        L = out["legs"]
        out["summary"] = {
            "f32_cosine": {"plain_ms": round(L["f32_cosine"]["plain"]["ms_per_step"], 4), "default_single_ms": round(L["f32_cosine"]["default_single"]["ms_per_step"], 4),
                           "single_evals_per_query": round(L["f32_cosine"]["default_single"]["exact_evaluations_per_query"], 1),
                           "batch_ms": round(L["f32_cosine"]["default_batch_%d" % nq_batch]["ms_per_step"], 3),
                           "batch_evals_per_query": round(L["f32_cosine"]["default_batch_%d" % nq_batch]["exact_evaluations_per_query"], 1)},
            "f32_dot": {"plain_ms": round(L["f32_dot"]["plain"]["ms_per_step"], 4), "default_single_ms": round(L["f32_dot"]["default_single"]["ms_per_step"], 4),
                        "batch_ms": round(L["f32_dot"]["default_batch_%d" % nq_batch]["ms_per_step"], 3)},
            "u8_768_cosine": {"plain_ms": round(L["u8_768_cosine"]["plain"]["ms_per_step"], 4), "default_single_ms": round(L["u8_768_cosine"]["default_single"]["ms_per_step"], 4),
                              "nibble_filter_in_use": L["u8_768_cosine"]["default_single"]["nibble_filter_in_use"],
                              "batch_ms": round(L["u8_768_cosine"]["default_batch_%d" % nq_batch]["ms_per_step"], 3)},
            "adversarial_default_over_plain": round(L["adversarial_f32_cosine"]["default_over_plain"], 3),
        }
    except Exception as e:
        out["error"] = repr(e)
    return out


def also_long_rows(args, pkg, torch, k, device_index):
    """not a BASELINE config: 1024 queries x 10M x 1536 f32 dot top-20 - rows longer than a wavefront's registers hold, the K dimension
    split over a workgroup's wavefronts (vg_batch_hl.hip) - next to one scan per query, which is what such batches were until round 4"""
    try:
        vt, np_dtype, dim, metric, desc = WORKLOADS["c5l"]
        n_rows = args.rows if args.rows else 10_000_000
        c = make_shard(pkg, torch, vt, dim, n_rows, 77, device_index)
        c.set_profiling(True)
        try:
            line = run_batched(args, pkg, torch, c, "c5l", n_rows, dim, metric, k, desc if n_rows == 10_000_000 else desc.replace("10M", "%gM" % (n_rows / 1e6)))
            first = run_batched.last_result
            # the same batches through round 4's path for such rows (the K-split bf16 kernel), priced on the bf16 peak
            try:
                os.environ["VG_BATCH_Q8"] = "0"
                pkg.reload_switches()
                c.close()
                c = make_shard(pkg, torch, vt, dim, n_rows, 77, device_index)
                c.set_profiling(True)
                old = run_batched(args, pkg, torch, c, "c5l", n_rows, dim, metric, k, desc)
                ores = run_batched.last_result
                line["bf16_ksplit_batch"] = {
                    "what": "VG_BATCH_Q8=0: vg_batch_hl_kernel (bf16 shadow copy, K split over a workgroup's wavefronts) + exact f32 re-evaluation",
                    "ms_per_step": old["ms_per_step"], "kernel": old["roofline"]["kernel"], "kernel_ms": old["roofline"]["kernel_ms"],
                    "frac_of_bf16_peak": old["roofline"]["frac"], "batch_path": old["roofline"].get("batch_path"),
                    "default_path_speedup": old["ms_per_step"] / line["ms_per_step"],
                    "last_batch_bit_identical_to_the_default_path": bool(np.array_equal(np.asarray(first[0]), np.asarray(ores[0])) and
                                                                         np.array_equal(np.asarray(first[1], dtype=np.float32).view(np.uint32), np.asarray(ores[1], dtype=np.float32).view(np.uint32)))}
            except Exception as e:                                    # noqa: BLE001
                line["bf16_ksplit_batch"] = {"error": repr(e)}
            finally:
                os.environ.pop("VG_BATCH_Q8", None)
                pkg.reload_switches()
        finally:
            c.close()
            torch.cuda.empty_cache()
        return {kk: line[kk] for kk in ("metric", "value", "unit", "ms_per_step", "config", "roofline", "against_single_scans", "bf16_ksplit_batch") if kk in line}
    except Exception as e:
        return {"error": repr(e)}


def also_c5(args, pkg, torch, corpus, n_rows, k):
    try:
        corpus.set_scan_filter(0)
        v5, t5, d5, m5, desc5 = WORKLOADS["c5"]
        line = run_batched(args, pkg, torch, corpus, "c5", n_rows, d5, m5, k, desc5)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = batch_cpu_baseline(args, v5, t5, d5, m5, k, seconds=5.0, sample=(pkg, torch, n_rows, 42))
        # the same batches through the bf16 filter (VG_F32_FILTER=1, the shadow copy the filter scan above has made): the
        # GEMM at the bf16 rate over HALF the bytes, every survivor re-evaluated with the f32 single-scan arithmetic.
        # Priced on the bf16 MFMA peak and reported next to the f32 MFMA line, never as its roofline.
        plain_res = run_batched.last_result
        try:
            os.environ["VG_F32_FILTER"] = "1"
            os.environ["VG_BATCH_Q8"] = "0"
            pkg.reload_switches()
            fl = run_batched(args, pkg, torch, corpus, "c5f", n_rows, d5, m5, k, WORKLOADS["c5f"][4])
            fres = run_batched.last_result
            same_ids = bool(np.array_equal(np.asarray(fres[0]), np.asarray(plain_res[0])))
            d_f, d_p = np.asarray(fres[1], dtype=np.float64), np.asarray(plain_res[1], dtype=np.float64)
            line["filter_batch"] = {
                "what": "the same batches through vg_batch_h_kernel over the bf16 shadow copy (matrix cores as a lower-bound "
                        "filter) + exact f32 re-evaluation of the survivors (VG_F32_FILTER=1 VG_BATCH_Q8=0: round 4's default path)",
                "value": fl["value"], "unit": "vectors/s", "ms_per_step": fl["ms_per_step"], "dtype_streamed": "bf16",
                "kernel": fl["roofline"]["kernel"], "kernel_ms": fl["roofline"]["kernel_ms"],
                "achieved_TFLOPs_of_the_QxNxD_product": fl["roofline"]["achieved"], "peak_bf16_TFLOPs": F16_MFMA_PEAK_TF,
                "frac_of_bf16_peak": fl["roofline"]["frac"], "speedup_over_f32_mfma_kernel": line["ms_per_step"] / fl["ms_per_step"],
                "last_batch_same_rowids_as_f32_mfma_kernel": same_ids,
                "last_batch_rowid_slots_that_differ": "%d of %d (near-ties: the two kernels' distances differ by summation order; "
                                                      "tests/test_gpu_fullsize.py checks both against the reference's own kernel)" % (
                    int(np.sum(np.asarray(fres[0]) != np.asarray(plain_res[0]))), int(np.asarray(fres[0]).size)),
                "last_batch_max_rel_distance_difference": float(np.max(np.abs(d_f - d_p) / np.maximum(np.abs(d_p), 1e-30))) if d_f.shape == d_p.shape else None,
            }
            line["filter_batch"]["traffic"] = fl["roofline"].get("traffic")      # HBM bytes per batch (PMC FETCH_SIZE pass); the tile-major bf16 copy is 7.68 GB
            line["filter_batch"]["traffic_source"] = fl["roofline"].get("traffic_source")
        except Exception as e:
            line["filter_batch"] = {"error": repr(e)}
        finally:
            os.environ.pop("VG_F32_FILTER", None)
            os.environ.pop("VG_BATCH_Q8", None)
            pkg.reload_switches()
        # the product's DEFAULT path for this batch (round 5): the int8 shadow copy on the integer matrix cores as the filter, 64 queries per
        # wavefront (vg_batch_q8.hip), the same exact f32 re-evaluation behind it - priced on the int8 MFMA rate
        try:
            corpus.set_scan_filter(-1)
            corpus.batch_filter_exact_evals()
            ql = run_batched(args, pkg, torch, corpus, "c5q", n_rows, d5, m5, k, WORKLOADS["c5q"][4])
            qres = run_batched.last_result
            fres = locals().get("fres")
            ib = {"what": "the same batches through the default path: vg_batch_q8_kernel over the int8 shadow copy (3.84 GB streamed) + vg_batch_hx_kernel "
                          "(exact f32 re-evaluation of the pairs that pass), staged over growing row ranges",
                  "batch_path": ql["roofline"].get("batch_path"), "value": ql["value"], "unit": "vectors/s", "ms_per_step": ql["ms_per_step"],
                  "dtype_streamed": "int8", "kernel": ql["roofline"]["kernel"], "kernel_ms": ql["roofline"]["kernel_ms"],
                  "achieved_TOPs_of_the_QxNxD_product": ql["roofline"]["achieved"], "peak_int8_TOPs": I8_MFMA_PEAK_TOPS, "frac_of_int8_peak": ql["roofline"]["frac"],
                  "speedup_over_f32_mfma_kernel": line["ms_per_step"] / ql["ms_per_step"],
                  "speedup_over_bf16_filter": (line["filter_batch"]["ms_per_step"] / ql["ms_per_step"]) if "ms_per_step" in line.get("filter_batch", {}) else None,
                  "last_batch_bit_identical_to_the_bf16_filter": bool(fres is not None and np.array_equal(np.asarray(qres[0]), np.asarray(fres[0])) and
                                                                      np.array_equal(np.asarray(qres[1]), np.asarray(fres[1]))),
                  "traffic": ql["roofline"].get("traffic"), "traffic_source": ql["roofline"].get("traffic_source"),
                  "exact_evaluations_per_query": None}
            try:
                ib["exact_evaluations_per_query"] = corpus.batch_filter_exact_evals() / float(args.batch * (min(args.steps, 10) + min(args.warmup, 2)))
            except Exception:
                pass
            # the last batch's winners against the REFERENCE's own kernel: the rows the GPU returned for 8 of its queries, regenerated from the seeded
            # stream, distance-avx2.c's dot through the dispatch table (oracle/_ref) - the f32 bar is 1e-5 relative
            try:
                from oracle import orc
                if orc.have_ref():
                    ref = orc.RefKernels("avx2")
                    qb = batch_queries(v5, args.batch, d5, which=(min(args.steps, 10) - 1) % 2)
                    ids = np.asarray(qres[0])
                    pick = list(range(0, args.batch, max(1, args.batch // 8)))[:8]
                    need = sorted(set(int(r) - 1 for qi in pick for r in ids[qi][:k]))
                    got = rows_at(pkg, torch, v5, d5, n_rows, 42, need)
                    worst = 0.0
                    for qi in pick:
                        for j in range(k):
                            dref = ref.distance(m5, v5, qb[qi], got[int(ids[qi][j]) - 1])
                            worst = max(worst, abs(float(np.asarray(qres[1])[qi][j]) - dref) / max(abs(dref), 1e-30))
                    ib["last_batch_max_rel_difference_from_the_reference_kernel"] = worst
                    ib["reference_check"] = "%d queries x %d returned rows, reference distance-avx2.c dot on the same rows (oracle/_ref/libref_avx2.so)" % (len(pick), k)
            except Exception as e:
                ib["reference_check"] = "unavailable: %r" % (e,)
            line["int8_filter_batch"] = ib
        except Exception as e:
            line["int8_filter_batch"] = {"error": repr(e)}
        finally:
            corpus.set_scan_filter(0)
        return line
    except Exception as e:
        return {"error": repr(e)}


MATRIX_TYPES = {2: ("f16", np.float16), 3: ("bf16", None), 5: ("i8", np.int8)}


def also_kernel_matrix(args, pkg, torch, shard, n_rows, k, device_index):
    """the element types the driver's lines never touch - f16, bf16, int8 - through their PLAIN scan kernels (filter off), L2 and
    cosine, 10M x 384, priced like the headline: N x D x elem bytes per launch / the kernel's mean HIP-event time / 8 TB/s"""
    out = {"what": "plain scan kernels (scan_filter=0), %d x 384, top-%d, %d timed single queries each: algorithmic bytes N*D*elem / "
                   "mean kernel time (HIP events on the launch stream) / %.0f GB/s" % (n_rows, k, args.steps, HBM_PEAK_GBS), "rows": []}
    dim = 384
    for vt, (tag, _) in MATRIX_TYPES.items():
        try:
            c = make_shard(pkg, torch, vt, dim, n_rows, 60 + vt, device_index)
            c.set_scan_filter(0)
            c.set_tie_order(pkg.TIE_POSITION)
            c.set_profiling(True)
            nq = args.steps + args.warmup
            qf = np.random.default_rng(61).standard_normal((nq, dim), dtype=np.float32)
            if vt == 2:
                q = qf.astype(np.float16)
            elif vt == 3:
                q = torch.from_numpy(qf).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
            else:
                q = np.clip(np.rint(qf * 40.0), -128, 127).astype(np.int8)
            for metric, mname in ((1, "l2"), (3, "cosine")):
                r = SingleQueryRunner(pkg, torch, None, shard, c, vt, dim, metric, k, n_rows, 1, q)
                elapsed, _ = r.run(args.warmup, args.steps)
                n_launch, scan_ms, merge_ms, _ = c.profile_mean_ms_ex()
                ab = n_rows * dim * pkg.TYPE_SIZE[vt]
                ach = ab / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
                out["rows"].append({"dtype": tag, "metric": mname, "kernel": c.kernel_name(metric), "kernel_ms": scan_ms,
                                    "ms_per_step": elapsed / args.steps * 1e3, "launches_timed": n_launch,
                                    "algorithmic_bytes_per_launch": ab, "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
            c.close()
            torch.cuda.empty_cache()
        except Exception as e:
            out["rows"].append({"dtype": tag, "error": repr(e)})
    return out


def also_c1(args, pkg, torch):
    """configs[0]: 10k x 384 f32 L2 top-20 through SQL (vector_full_scan), this repo's vector.so next to the reference's - the
    `--workload c1` line, captured"""
    import contextlib
    import io
    try:
        a2 = argparse.Namespace(**vars(args))
        a2.rows, a2.steps, a2.warmup = 10_000, 50, 5
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench_sql(a2, pkg, torch)
        return json.loads(buf.getvalue().strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}


def also_c4_one_gpu(args, pkg, torch, shard, k, device_index):
    """north_star's target sentence, literally: single-query f32 L2 over 100M x 384 - resident on ONE device (153.6 GB of its
    288 GB), the plain kernel, 10 timed queries"""
    try:
        n = 100_000_000
        free, _ = torch.cuda.mem_get_info()
        if free < n * 384 * 4 + (8 << 30):
            return {"skipped": "needs %.1f GB of free device memory, %.1f GB free" % (n * 1536 / 1e9 + 8.6, free / 1e9)}
        vt, _, dim, metric, _ = WORKLOADS["c2"]
        c = make_shard(pkg, torch, vt, dim, n, 42, device_index)
        c.set_scan_filter(0)
        c.set_profiling(True)
        steps, warmup = 10, 2
        q = np.random.default_rng(43).standard_normal((steps + warmup, dim), dtype=np.float32)
        r = SingleQueryRunner(pkg, torch, None, shard, c, vt, dim, metric, k, n, 1, q)
        a2 = argparse.Namespace(**vars(args))
        a2.steps, a2.warmup = steps, warmup
        line, _ = single_query_line(a2, pkg, r, c, "c2", vt, dim, metric, k, n, 1,
                                    "100Mx384 f32 L2 top-20 single-query, the whole corpus resident on ONE MI355X (plain kernel)")
        c.close()
        torch.cuda.empty_cache()
        return line
    except Exception as e:
        return {"error": repr(e)}


def bench_stage(args, pkg, torch):
    """--workload stage: the passes in front of the scans, each priced on its own bytes.
       staging    host rows -> HBM through vg_corpus_append (pinned double buffer + H2D) and through vg_corpus_append_records (the
                  reference's persisted [int64 rowid | vector] records, de-interleaved on the device): bound by the host link
       minmax     vector_quantize pass 1 over the resident f32 corpus: reads N*D*4 bytes              } kernel time from HIP events
       quantize   vector_quantize pass 2: reads N*D*4, writes N*D (the D2H of the result is not in it) } on the corpus stream, against
       q8_shadow  the filter scans' int8 shadow copy: reads N*D*4, writes N*(D+8)                      } the 8 TB/s HBM peak"""
    n = args.rows if args.rows else 4_000_000
    dim = 384
    rng = np.random.default_rng(5)
    host = rng.random((1 << 20, dim), dtype=np.float32)
    out = {"metric": "staging + quantization throughput", "unit": "GB/s", "n_gpus": 1, "steps": 1, "warmup": 0, "data": "synthetic",
           "dtype": "f32", "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "config": {"workload": "stage: %d x %d f32 rows staged from host memory, then quantized on the device" % (n, dim),
                      "backend": pkg.backend_name()}}
    # warm-up on a throwaway corpus: the first launch of a kernel loads its code object (milliseconds on the host, inside any
    # event bracket around that launch) - every pass below is timed on its second use
    w = pkg.Corpus(pkg.F32, dim, capacity=8192)
    w.append(host[:8192])
    lo_w, hi_w, _ = w.minmax()
    w.quantize_rows(255.0 / max(hi_w - lo_w, 1e-6), lo_w, pkg.QUANT_U8, 0, 8192)
    os.environ["VG_SCAN_FILTER_MIN_MB"] = "0"
    pkg.reload_switches()
    w.set_scan_filter(1)
    w.scan_topk(1, host[0], args.k)
    os.environ.pop("VG_SCAN_FILTER_MIN_MB", None)
    pkg.reload_switches()
    w.close()
    c = pkg.Corpus(pkg.F32, dim, capacity=n)
    c.append(host[:4096])                                   # warm: pinned buffers, stream
    c.minmax()
    c.clear()
    t0 = time.perf_counter()
    done = 0
    while done < n:
        take = min(host.shape[0], n - done)
        c.append(host[:take])
        done += take
    lo, hi, neg = c.minmax()                                # (appends are only enqueued: this waits for them on the same stream)
    mm_ms, mm_rows = c.pass_ms("minmax")
    stage_s = time.perf_counter() - t0 - mm_ms * 1e-3
    out["staging"] = {"rows": n, "bytes": n * dim * 4, "seconds": stage_s, "achieved": n * dim * 4 / stage_s / 1e9, "unit": "GB/s",
                      "bound": "host link", "note": "vg_corpus_append: host memcpy into a pinned bounce buffer + enqueued H2D, overlapped; 1 host thread"}
    # the reference's persisted record format: [int64 LE rowid | dim bytes], stride 8 + dim, de-interleaved by vg_repack_kernel
    try:
        nrec, dq = 2_000_000, 768
        rec = np.zeros((1 << 19, 8 + dq), dtype=np.uint8)
        rec[:, 8:] = rng.integers(0, 256, (1 << 19, dq), dtype=np.uint8)
        rec[:, :8] = np.arange(1, (1 << 19) + 1, dtype="<i8").view(np.uint8).reshape(-1, 8)
        cq = pkg.Corpus(pkg.U8, dq, capacity=nrec)
        cq.append_records(rec[:1024], 1024)
        cq.minmax()
        cq.clear()
        t1 = time.perf_counter()
        done = 0
        while done < nrec:
            take = min(rec.shape[0], nrec - done)
            cq.append_records(rec[:take], take)
            done += take
        cq.minmax()
        qmm_ms, _ = cq.pass_ms("minmax")
        rs = time.perf_counter() - t1 - qmm_ms * 1e-3
        out["staging_records"] = {"rows": nrec, "bytes": nrec * (8 + dq), "seconds": rs, "achieved": nrec * (8 + dq) / rs / 1e9,
                                  "unit": "GB/s", "bound": "host link",
                                  "note": "vg_corpus_append_records: [rowid | vector] records of vector0_<t>_<c>, de-interleaved on the device"}
        cq.close()
    except Exception as e:
        out["staging_records"] = {"error": repr(e)}
    res = {}

    def priced(kernel, ms, nbytes, rows):
        ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"kernel": kernel, "kernel_ms": ms, "rows": rows, "bytes": nbytes, "bound": "hbm", "achieved": ach,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}

    res["minmax"] = priced("vg_minmax_kernel<f32>", mm_ms, mm_rows * dim * 4, mm_rows)
    scale = 255.0 / (hi - lo) if hi > lo else 1.0
    c.quantize_rows(scale, lo, pkg.QUANT_U8, 0, n)
    q_ms, q_rows = c.pass_ms("quantize")
    res["quantize"] = priced("vg_quantize_kernel<f32>", q_ms, q_rows * dim * 5, q_rows)
    try:
        os.environ["VG_SCAN_FILTER_MIN_MB"] = "0"
        pkg.reload_switches()
        c.set_scan_filter(1)
        c.set_profiling(True)
        c.scan_topk(1, host[0], args.k)                      # the first filter scan builds the shadow copy
        s_ms, s_rows = c.pass_ms("q8_shadow")
        res["q8_shadow"] = priced("vg_to_q8_reg_kernel<f32>", s_ms, s_rows * (dim * 4 + dim + 8), s_rows)
    except Exception as e:
        res["q8_shadow"] = {"error": repr(e)}
    finally:
        os.environ.pop("VG_SCAN_FILTER_MIN_MB", None)
        pkg.reload_switches()
    out["quantize"] = res
    out["value"] = out["staging"]["achieved"]
    out["roofline"] = dict(res["quantize"])
    c.close()
    print(json.dumps(out))
    return 0


def batch_queries(vt, nq, dim, which=0):
    """the batches run_batched times (its seeded stream, batch `which` of two)"""
    rng = np.random.default_rng(44)
    out = None
    for _ in range(which + 1):
        if vt in (4, 5):
            out = rng.integers(0, 256, (nq, dim)).astype(np.uint8)
        elif vt == 2:
            out = rng.standard_normal((nq, dim), dtype=np.float32).astype(np.float16)
        else:
            out = rng.standard_normal((nq, dim), dtype=np.float32)
    return out


def batch_cpu_baseline(args, vt, np_dtype, dim, metric, k, seconds=10.0, sample=None):
    """the reference has no batched entry point: its batch is Q independent scans, so its (query, vector) pair rate is its
    single-scan rate for the batch's metric - timed over rows of the GPU's corpus with queries of the timed batch (sample = (pkg, torch,
    rows of the corpus, its seed))"""
    try:
        rows = queries = None
        if sample is not None and vt != 3:
            pkg, torch, n_rows, seed = sample
            rows = corpus_sample(pkg, torch, vt, dim, n_rows, seed, args.cpu_sample_rows)
            queries = batch_queries(vt, args.batch, dim)[:8]
            if vt == 2:
                queries = queries.view(np.uint16)
        out = cpu_baseline(vt, np_dtype, dim, metric, k, args.cpu_sample_rows, seconds=seconds, all_cores=False, rows=rows, queries=queries)
        out["sample"] += "; a batch of Q queries costs the reference Q such scans: (query, vector) pairs/s = this rate"
        return out
    except Exception as e:
        return {"value": None, "unit": "vectors/s", "cores": 0, "kind": "port", "sample": "unavailable: %r" % (e,)}


if __name__ == "__main__":
    sys.exit(main() or 0)
