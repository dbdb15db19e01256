"""bench_common.py - what bench.py (the driver's contract line) and bench_extras.py (the sub-results, the other workloads) share: the
workload table, the seeded corpora, the timed runners, the CPU legs, the roofline pricing.  Split out of bench.py in round 6 (one 110 KB file
before); `python bench.py` is still the only entry point."""
import argparse
import json
import os
import sys
import time

import numpy as np


ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
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


def load_shard_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vg_shard", os.path.join(ROOT, "sqlite-vector_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    return shard


def c3_queries(nq, dim):
    """queries of config C3: f32 U[0,1) quantized like the corpus"""
    return quantize_unit_uniform_np(np.random.default_rng(43).random((nq, dim), dtype=np.float32))


def quantize_unit_uniform_np(v):
    """the reference's uint8 quantizer (sqlite-vector.c:517-548) with the parameters a U[0,1) source gets: offset = min = 0,
    scale = 255 / (max - min) = 255: (uint8)(v * 255 + 0.5)"""
    return np.clip(np.floor(v * np.float32(255.0) + np.float32(0.5)), 0, 255).astype(np.uint8)


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
