"""GPU parity tests proper: the HIP scan, called through the C-ABI (libvectorgpu.so), against the pinned CPU
oracle on the same seeded inputs.

Bars (BASELINE.json north_star):
  * int8 / uint8: distances BIT-EXACT with the reference's distance-avx2.c path, identical rowid order
    (ties resolved by scan position, the GPU contract - see DESIGN.md);
  * f32: distances within 1e-5 relative of the reference arithmetic (plus an absolute term for the metrics that
    cancel: dot / cosine), top-k identical to the (distance, position) order of the GPU's own distances.
"""
import numpy as np
import pytest

import datagen as dg

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


@pytest.fixture(scope="module")
def pkg():
    # torch bundles its own HIP runtime: when a test needs both, torch must initialise it first so that
    # libvectorgpu.so binds to the same libamdhip64 instance (bench.py has the same order)
    try:
        import torch
        torch.cuda.init()
    except Exception:
        pass
    import __graft_entry__ as g
    p = g.load_package()
    if p.device_count() < 1:
        pytest.fail("GPU tests need a HIP device (the product has no CPU fallback)")
    return p


def _abs_scale(vt, metric, q, rows):
    """magnitude of the terms being summed - the natural absolute error scale for cancelling metrics"""
    qf = dg.storage_to_f64(vt, q)
    rf = dg.storage_to_f64(vt, rows)
    if metric == dg.DOT:
        return np.abs(rf * qf).sum(axis=1)
    return np.ones(rows.shape[0])          # cosine is O(1)


def _check_float_distances(got, want, vt, metric, q, rows):
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    both_nan = np.isnan(got) & np.isnan(want)
    same_inf = np.isinf(want) & (got == want)
    fin = ~(both_nan | same_inf)
    assert not np.isnan(got[fin]).any() and not np.isinf(got[fin]).any(), "NaN/Inf mismatch"
    err = np.abs(got[fin] - want[fin])
    tol = REL_TOL * np.abs(want[fin])
    if metric in (dg.DOT, dg.COSINE):
        tol = tol + REL_TOL * _abs_scale(vt, metric, q, rows)[fin] * (1.0 if metric == dg.DOT else 1.0)
    # values both sides clamp to exactly 0 (|d| <= 8 eps) may straddle the clamp
    tol = np.maximum(tol, 8 * np.finfo(np.float32).eps * 1.01)
    bad = err > tol
    assert not bad.any(), (np.nonzero(bad)[0][:5], got[fin][bad][:5], want[fin][bad][:5])


DIMS_F32 = (1, 3, 4, 5, 16, 35, 100, 128, 384, 768, 1000, 1024, 1536)
DIMS_INT = (1, 3, 15, 16, 17, 35, 100, 384, 768, 1000, 1536, 2048)


@pytest.mark.parametrize("dim", DIMS_F32)
def test_f32_all_metrics_vs_oracle(pkg, orc, dim):
    n = 2500
    rows = dg.corpus(dg.F32, n, dim, 300 + dim)
    q = dg.query(dg.F32, dim, 301 + dim)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for metric in dg.ALL_METRICS:
        want = orc.scan_distances(orc.AVX2, metric, dg.F32, q, rows)
        got = c.scan_distances(metric, q)
        _check_float_distances(got, want, dg.F32, metric, q, rows)
        for k in (1, 20, 64):
            ids, dist = c.scan_topk(metric, q, k)
            oids, odist, _ = orc.topk_ordered(got, None, k)        # selection must be exact on the GPU's own floats
            assert ids.tolist() == oids.tolist(), (dim, metric, k)
            assert np.array_equal(dist, odist)
    c.close()


@pytest.mark.parametrize("vt", [dg.U8, dg.I8])
@pytest.mark.parametrize("dim", DIMS_INT)
def test_int8_bit_exact_vs_oracle(pkg, orc, vt, dim):
    n = 2500
    for low in (False, True):                      # low entropy -> many exact ties
        rows = dg.corpus(vt, n, dim, 400 + dim, low_entropy=low)
        q = dg.query(vt, dim, 401 + dim, low_entropy=low)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for metric in dg.ALL_METRICS:
            want = orc.scan_distances(orc.AVX2, metric, vt, q, rows)
            got = c.scan_distances(metric, q)
            assert dg.same_float_bits(got, want), (dg.TYPE_NAMES[vt], dg.METRIC_NAMES[metric], dim, low,
                                                   np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0][:5])
            for k in (1, 20, 64):
                ids, dist = c.scan_topk(metric, q, k)
                oids, odist, _ = orc.topk_ordered(want, None, k)
                assert ids.tolist() == oids.tolist(), (dim, metric, k, low)
                assert np.array_equal(dist, odist)
                # the reference's own (history dependent) slot algorithm returns the same distance sequence
                rids, rdist = orc.topk_reference(want, None, k)
                assert np.array_equal(rdist, odist)
        c.close()


def test_int8_extreme_values_bit_exact(pkg, orc):
    """all-255 / all-(-128) rows: the largest integer sums (768*255^2 > 2^24: one rounding at the end only)."""
    for vt in (dg.U8, dg.I8):
        dim = 768
        q, rows = dg.edge_rows(vt, dim, 77)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for metric in dg.ALL_METRICS:
            for qq in dg.edge_queries(vt, dim, 78):
                want = orc.scan_distances(orc.AVX2, metric, vt, qq, rows)
                got = c.scan_distances(metric, qq)
                assert dg.same_float_bits(got, want), (vt, metric)
        c.close()


def test_f32_nan_inf_rows_never_enter_topk(pkg, orc):
    dim = 35
    q, rows = dg.edge_rows(dg.F32, dim, 90)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for metric in dg.ALL_METRICS:
        got = c.scan_distances(metric, q)
        want = orc.scan_distances(orc.AVX2, metric, dg.F32, q, rows)
        assert np.array_equal(np.isnan(got), np.isnan(want)), metric
        assert np.array_equal(np.isposinf(got), np.isposinf(want)), metric
        ids, dist = c.scan_topk(metric, q, 64)
        oids, odist, _ = orc.topk_ordered(got, None, 64)
        assert ids.tolist() == oids.tolist()
        assert len(ids) == int(np.sum(got < np.inf))          # NaN and +Inf rows are absent, -Inf/finite present
    # exact copy of the query: L2 distance clamps to exactly +0.0 and wins
    ids, dist = c.scan_topk(dg.L2, q, 1)
    assert ids.tolist() == [1] and dist[0] == 0.0
    c.close()


def test_short_table_empty_table_and_k_edge_cases(pkg, orc):
    dim = 8
    rows = dg.corpus(dg.F32, 5, dim, 1)
    q = dg.query(dg.F32, dim, 2)
    c = pkg.Corpus(pkg.F32, dim)
    ids, dist = c.scan_topk(dg.L2, q, 10)
    assert len(ids) == 0                                       # empty corpus
    c.append(rows)
    ids, dist = c.scan_topk(dg.L2, q, 10)                      # fewer rows than k (sqlite-vector.c:1816-1817)
    d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, rows)
    oids, odist, _ = orc.topk_ordered(d, None, 10)
    assert ids.tolist() == oids.tolist() and len(ids) == 5
    ids, dist = c.scan_topk(dg.L2, q, 0)                       # k == 0 -> nothing (:1796)
    assert len(ids) == 0
    c.close()


def test_large_k_path(pkg, orc):
    dim, n = 48, 3000
    rows = dg.corpus(dg.I8, n, dim, 5, low_entropy=True)
    q = dg.query(dg.I8, dim, 6, low_entropy=True)
    c = pkg.Corpus(pkg.I8, dim)
    c.append(rows)
    want = orc.scan_distances(orc.AVX2, dg.L1, dg.I8, q, rows)
    for k in (65, 500, n, n + 10):
        ids, dist = c.scan_topk(dg.L1, q, k)
        oids, odist, _ = orc.topk_ordered(want, None, k)
        assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist)
    c.close()


def test_rowids_strides_and_reference_record_format(pkg, orc):
    """explicit rowids, a padded host stride, incremental appends, and the reference's [int64 rowid | dim bytes]
    preload records (stride 8+dim, unaligned vectors) all stage to the same corpus."""
    dim, n = 100, 1200
    rows = dg.corpus(dg.U8, n, dim, 9)
    q = dg.query(dg.U8, dim, 10)
    rowids = (np.arange(n, dtype=np.int64) * 7 + 3) * np.where(np.arange(n) % 2 == 0, 1, -1)
    want = orc.scan_distances(orc.AVX2, dg.COSINE, dg.U8, q, rows)
    oids, odist, _ = orc.topk_ordered(want, rowids, 20)

    c1 = pkg.Corpus(pkg.U8, dim)
    c1.append(rows[:500], rowids[:500])
    c1.append(rows[500:], rowids[500:])
    ids, dist = c1.scan_topk(dg.COSINE, q, 20)
    assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist)
    c1.close()

    # 112 is exactly the corpus' own padded HBM stride for 100 bytes: the gap must still be zeroed, not imported
    for stride in (112, 128, 101):
        padded = np.zeros((n, stride), dtype=np.uint8)
        padded[:, :dim] = rows
        padded[:, dim:] = 0xAB                                 # garbage in the stride gap must be ignored
        c2 = pkg.Corpus(pkg.U8, dim)
        c2.append_strided(padded, n, stride, rowids)
        ids, dist = c2.scan_topk(dg.COSINE, q, 20)
        assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist), stride
        assert dg.same_float_bits(c2.scan_distances(dg.L2, q), orc.scan_distances(orc.AVX2, dg.L2, dg.U8, q, rows))
        c2.close()

    rec = np.zeros((n, 8 + dim), dtype=np.uint8)
    rec[:, :8] = rowids.astype("<i8").view(np.uint8).reshape(n, 8)
    rec[:, 8:] = rows
    c3 = pkg.Corpus(pkg.U8, dim)
    c3.append_records(rec, n)
    ids, dist = c3.scan_topk(dg.COSINE, q, 20)
    assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist)
    got = c3.scan_distances(dg.COSINE, q)
    assert dg.same_float_bits(got, want)
    c3.close()


def test_logical_shards_merge_equals_single_shard(pkg, orc):
    """row-range sharding (the 8-GPU layout) on one device: per-shard device keys + vg_merge_keys == 1 shard."""
    import torch
    dim, n, k, G = 64, 5000, 20, 4
    rows = dg.corpus(dg.I8, n, dim, 21, low_entropy=True)      # heavy ties across shard borders
    q = dg.query(dg.I8, dim, 22, low_entropy=True)
    whole = pkg.Corpus(pkg.I8, dim)
    whole.append(rows)
    ids1, dist1 = whole.scan_topk(dg.SQUARED_L2, q, k)
    whole.close()
    bounds = [0, 1300, 2600, 2601, n]
    qd = torch.zeros(((dim + 15) // 16) * 16, dtype=torch.uint8, device="cuda")
    qd[:dim] = torch.from_numpy(q.view(np.uint8)).cuda()
    keys = torch.empty((G, 64), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    st = torch.cuda.Stream()                                    # non-null handle (0 = "corpus stream" to the C-ABI)
    shards = []
    for g in range(G):
        s = pkg.Corpus(pkg.I8, dim)
        s.append(rows[bounds[g]:bounds[g + 1]])
        s.scan_topk_device(dg.SQUARED_L2, qd.data_ptr(), k, keys[g].data_ptr(), st.cuda_stream)
        shards.append(s)
    st.synchronize()
    pos, dist = pkg.merge_keys(keys.cpu().numpy().view(np.uint64), bounds[:G], k)
    assert (pos + 1).tolist() == ids1.tolist()
    assert np.array_equal(dist, dist1)
    for s in shards:
        s.close()


def _assert_topk_close(ids, dist, all_dist, k, tol):
    """top-k validity under a floating-point tolerance: returned rows carry (within tol) their oracle distance, come
    back ascending, and no unreturned row is better than the k-th returned one by more than tol."""
    want_k = min(k, int(np.sum(all_dist < np.inf)))
    assert len(ids) == want_k
    assert np.all(np.diff(dist) >= 0)
    pos = np.asarray(ids) - 1
    assert len(set(pos.tolist())) == len(pos)
    assert np.all(np.abs(dist - all_dist[pos].astype(np.float64)) <= tol[pos])
    if want_k:
        rest = np.ones(all_dist.shape[0], dtype=bool)
        rest[pos] = False
        rest &= all_dist < np.inf
        assert np.all(all_dist[rest].astype(np.float64) >= dist[-1] - tol[rest])


@pytest.mark.parametrize("dim", (16, 100, 128, 384, 500, 512))
@pytest.mark.parametrize("metric", (dg.DOT, dg.COSINE, dg.L2, dg.SQUARED_L2))
def test_batch_mfma_path_vs_oracle(pkg, orc, dim, metric):
    """vg_scan_topk_batch on f32 dot / cosine runs Q x C^T on the matrix cores (v_mfma_f32_32x32x2_f32) with the
    fused per-query top-k; every query is checked against the reference arithmetic (<= 1e-5 relative to the
    magnitude of the summed terms)."""
    n = 4133                                      # not a multiple of the 32-row tile
    rows = dg.corpus(dg.F32, n, dim, 900 + dim)
    rows[17] = 0.0                                # a zero-norm row (cosine -> 1.0)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for nq, k in ((1, 20), (7, 1), (130, 20), (200, 32)):
        qs = dg.corpus(dg.F32, nq, dim, 901 + dim + nq)
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        for i in range(nq):
            want = orc.scan_distances(orc.AVX2, metric, dg.F32, qs[i], rows)
            scale = (np.abs(rows.astype(np.float64) * qs[i].astype(np.float64)).sum(axis=1) if metric == dg.DOT
                     else np.ones(n) if metric == dg.COSINE else np.zeros(n))          # L2: purely relative
            tol = REL_TOL * (np.abs(want.astype(np.float64)) + scale) + 1e-6
            _assert_topk_close(ids[i][:cnt[i]], dist[i][:cnt[i]], want, k, tol)
    c.close()


def test_batch_small_corpus_and_fallback_shapes(pkg, orc):
    dim = 96
    rows = dg.corpus(dg.F32, 10, dim, 31)          # fewer rows than one tile, fewer than k
    qs = dg.corpus(dg.F32, 3, dim, 32)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    ids, dist, cnt = c.scan_topk_batch(dg.DOT, qs, 20)
    for i in range(3):
        want = orc.scan_distances(orc.AVX2, dg.DOT, dg.F32, qs[i], rows)
        assert cnt[i] == 10
        assert sorted(ids[i][:10].tolist()) == list(range(1, 11))
        assert np.allclose(dist[i][:10], np.sort(want), rtol=1e-5, atol=1e-5)
    c.close()
    # shapes the matrix-core kernel does not serve fall back to per-query scans with identical results
    rows = dg.corpus(dg.F32, 2000, dim, 33)
    qs = dg.corpus(dg.F32, 5, dim, 34)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for metric, k in ((dg.L1, 10), (dg.DOT, 40), (dg.L2, 40)):
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        for i in range(5):
            one_ids, one_dist = c.scan_topk(metric, qs[i], k)       # the multi-query scan may sum in another order
            assert cnt[i] == k and ids[i].tolist() == one_ids.tolist() and np.allclose(dist[i], one_dist, rtol=1e-5, atol=1e-6)
    c.close()
    r8 = dg.corpus(dg.I8, 1500, 64, 35)
    q8 = dg.corpus(dg.I8, 4, 64, 36)
    c8 = pkg.Corpus(pkg.I8, 64)
    c8.append(r8)
    ids, dist, cnt = c8.scan_topk_batch(dg.DOT, q8, 10)
    for i in range(4):
        one_ids, one_dist = c8.scan_topk(dg.DOT, q8[i], 10)
        assert ids[i].tolist() == one_ids.tolist() and np.array_equal(dist[i], one_dist)
    c8.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.F16, dg.BF16, dg.I8, dg.U8])
def test_batch_multi_query_scan_vs_single_scans(pkg, orc, vt, monkeypatch):
    """Batches the matrix-core kernels do not serve (f16 / bf16, L1, k > 32, long rows) run vg_scan_multi_kernel: 4
    (2 for f16 / bf16) queries share every pass over the rows.  Same arithmetic per (query, row) as the single scan:
    int8 / uint8 bit-exact, floats <= 1e-5 relative (summation order), same total order of the results."""
    for dim, n in ((7, 300), (100, 5000), (384, 9001), (1000, 2500), (1536, 1200)):
        rows = dg.corpus(vt, n, dim, 7000 + dim)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for metric in (dg.L2, dg.SQUARED_L2, dg.L1, dg.COSINE, dg.DOT):
            for nq, k in ((2, 1), (5, 20), (9, 64), (3, 7000)):
                qs = dg.corpus(vt, nq, dim, 7100 + dim + nq)
                monkeypatch.setenv("VG_MULTI_SCAN", "1")
                monkeypatch.setenv("VG_BATCH_MFMA", "0")
                ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
                monkeypatch.setenv("VG_MULTI_SCAN", "0")
                ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)
                assert np.array_equal(cnt, cnt0)
                for i in range(nq):
                    m = cnt[i]
                    if vt in (dg.I8, dg.U8):
                        assert np.array_equal(ids[i][:m], ids0[i][:m]) and np.array_equal(dist[i][:m], dist0[i][:m])
                        continue
                    want = orc.scan_distances(orc.AVX2, metric, vt, qs[i], rows)
                    _check_float_distances(dist[i][:m].astype(np.float32), want[ids[i][:m] - 1], vt, metric, qs[i], rows[ids[i][:m] - 1])
                    assert np.allclose(dist[i][:m], dist0[i][:m], rtol=1e-5, atol=1e-6)
                    assert np.all(np.diff(dist[i][:m]) >= 0)
        c.close()


@pytest.mark.parametrize("shadow", ("int8", "bf16"))
@pytest.mark.parametrize("dim", (3, 33, 100, 384, 768, 1024, 1536))
@pytest.mark.parametrize("vt", (dg.F32, dg.F16, dg.BF16))
def test_filter_scan_is_bit_identical_to_the_plain_scan(pkg, orc, vt, dim, shadow, monkeypatch):
    """L2 / squared-L2 / dot / cosine (f16 / bf16: also L1) top-k scans of f32, f16 and bf16 corpora go through a lower-bound filter
    (shadow = int8: an int8 shadow copy with a residual-norm bound, a quarter / half of the bytes; shadow = bf16: f32 corpora through the bf16
    shadow copy, half the bytes; f16 / bf16: their own rows with f32 sums instead of the reference's f64 chain) and re-evaluate the candidates
    with the plain kernel's accumulator in its summation order (vg_scan_filter.h): rowids and distance BITS must equal the
    plain scan's (filter off), for ordinary rows, edge rows (NaN / Inf / huge / tiny / zero / subnormal) and edge queries."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")         # (by default only corpora >= 3 GB / 1 GB take the filter scan)
    monkeypatch.setenv("VG_SCAN_FILTER_NO_GUARD", "1")       # (edge queries make every row a candidate: keep the filter kernel under test)
    monkeypatch.setenv("VG_SCAN_FILTER_SHADOW", shadow)
    n = 60_007
    rows = dg.corpus(vt, n, dim, 9700 + dim)
    _, edge = dg.edge_rows(vt, dim, 9800 + dim)
    rows[500:500 + len(edge)] = edge
    rows[40000] = rows[17]                                   # an exact duplicate: a tie decided by position
    small = dg.storage_to_f64(vt, rows[30000:30050]) * 1e-4  # tiny rows (f16: down into the subnormal range)
    rows[30000:30050] = dg.to_storage(vt, small.astype(np.float32))
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    queries = [rows[17].copy()] + dg.edge_queries(vt, dim, 9900 + dim) + [dg.query(vt, dim, 9901 + i) for i in range(4)]
    queries.append(rows[30007].copy())                       # a tiny (subnormal) query
    tag = dg.TYPE_NAMES[vt]
    for metric in (dg.L2, dg.SQUARED_L2, dg.DOT, dg.COSINE) + ((dg.L1,) if vt != dg.F32 else ()):   # (f32 L1: plain scan only)
        c.set_scan_filter(1)
        assert c.kernel_name(metric).startswith("scan_filter_" + tag), c.kernel_name(metric)
        assert ("_q8_" in c.kernel_name(metric)) == (shadow == "int8" and metric != dg.L1), c.kernel_name(metric)
        for qi, q in enumerate(queries):
            for k in (1, 20, 64):
                c.set_scan_filter(1)
                ids1, d1 = c.scan_topk(metric, q, k)
                c.set_scan_filter(0)
                ids0, d0 = c.scan_topk(metric, q, k)
                assert ids1.tolist() == ids0.tolist(), (vt, metric, qi, k)
                assert dg.same_float_bits(d1, d0), (vt, metric, qi, k)
    c.set_scan_filter(1)
    more = dg.corpus(vt, 300, dim, 9950)                     # appended rows extend the shadow copy and the norms
    more[5] = queries[-2]
    c.append(more)
    ids1, d1 = c.scan_topk(dg.L2, queries[-2], 3)
    assert ids1[0] == n + 6 and d1[0] == 0.0
    c.set_scan_filter(-1)
    c.close()


@pytest.mark.parametrize("shadow", ("int8", "bf16"))
@pytest.mark.parametrize("vt,dim", ((dg.F16, 64), (dg.BF16, 40), (dg.F32, 48)))
def test_filter_scan_with_its_prepass_on_clustered_rows(pkg, orc, vt, dim, shadow, monkeypatch):
    """n >= 2^20: the filter scan starts from its plain pre-pass' threshold; clustered unit-norm rows + near-duplicates of
    the query; every metric the filter serves; compared with the plain scan and the oracle"""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    monkeypatch.setenv("VG_SCAN_FILTER_NO_GUARD", "1")
    monkeypatch.setenv("VG_SCAN_FILTER_SHADOW", shadow)
    n = (1 << 20) + 777
    rng = np.random.default_rng(8800 + dim)
    centres = rng.standard_normal((9, dim)).astype(np.float32)
    x = centres[rng.integers(0, 9, n)] + np.float32(0.08) * rng.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    rows = dg.to_storage(vt, x)
    twin = n - 4321
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for qi, q in enumerate((rows[twin].copy(), rows[100].copy(), dg.to_storage(vt, centres[2] / np.linalg.norm(centres[2])))):
        for metric in (dg.L2, dg.SQUARED_L2, dg.DOT, dg.COSINE) + ((dg.L1,) if vt != dg.F32 else ()):
            c.set_scan_filter(1)
            ids1, d1 = c.scan_topk(metric, q, 20)
            c.set_scan_filter(0)
            ids0, d0 = c.scan_topk(metric, q, 20)
            assert ids1.tolist() == ids0.tolist(), (vt, metric, qi)
            assert dg.same_float_bits(d1, d0), (vt, metric, qi)
            want = orc.scan_distances(orc.AVX2, metric, vt, q, rows[ids1 - 1])
            _check_float_distances(d1.astype(np.float32), want, vt, metric, q, rows[ids1 - 1])
    c.set_scan_filter(1)
    c.filter_exact_evals()
    c.scan_topk(dg.COSINE, rows[100].copy(), 20)
    assert 0 < c.filter_exact_evals() < n // 4              # the bound is selective on this data
    c.close()


@pytest.mark.parametrize("metric", (dg.DOT, dg.COSINE, dg.L2, dg.SQUARED_L2))
def test_batch_f32_long_rows_through_the_bf16_filter(pkg, orc, metric, monkeypatch):
    """f32 rows of 513 .. 1024 floats (768- / 1024-dimensional embeddings) have no f32 matrix-core kernel: the batch runs
    the half-precision kernel on a bf16 SHADOW copy as a filter and re-evaluates every candidate on the f32 rows with the
    single-query kernel's arithmetic - the lists are the single scans' lists (f32 summation order aside).  With
    VG_F32_FILTER=1 shorter rows take the same path."""
    for dim, force in ((520, "0"), (768, "0"), (1000, "0"), (1024, "0"), (384, "1"), (100, "1")):
        monkeypatch.setenv("VG_F32_FILTER", force)
        n = 4133
        rows = dg.corpus(dg.F32, n, dim, 9200 + dim)
        _, edge = dg.edge_rows(dg.F32, dim, 9300 + dim)                   # zeros, huge / tiny magnitudes, NaN, Inf rows
        rows[100:100 + len(edge)] = edge
        rows[2000] = rows[17]
        rows[3000:3010] *= np.float32(1e-3)
        c = pkg.Corpus(pkg.F32, dim)
        c.append(rows)
        for nq, k in ((1, 20), (7, 1), (150, 20), (40, 32 if dim <= 768 else 16)):
            qs = dg.corpus(dg.F32, nq, dim, 9400 + dim + nq)
            eq = dg.edge_queries(dg.F32, dim, 9500 + dim)
            for i, q in enumerate(eq[:min(len(eq), nq - 1)]):
                qs[1 + i] = q
            qs[0] = rows[17]
            monkeypatch.setenv("VG_BATCH_MFMA", "1")
            ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
            monkeypatch.setenv("VG_BATCH_MFMA", "0")
            monkeypatch.setenv("VG_MULTI_SCAN", "0")
            ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)           # nq single-query scans
            monkeypatch.delenv("VG_MULTI_SCAN")
            assert np.array_equal(cnt, cnt0), (dim, nq, k)
            for i in range(nq):
                m = cnt[i]
                if metric == dg.DOT:        # cancelling sums: the bar is relative to the magnitude of the summed terms
                    both = np.intersect1d(ids[i][:m], ids0[i][:m])
                    assert len(both) >= m - 2, (dim, i)
                    for r in both:
                        a, b = dist[i][:m][ids[i][:m] == r][0], dist0[i][:m][ids0[i][:m] == r][0]
                        scale = float(np.abs(rows[r - 1].astype(np.float64) * qs[i].astype(np.float64)).sum())
                        assert (np.isinf(a) and a == b) or abs(a - b) <= 1e-5 * (abs(b) + scale) + 1e-6, (dim, i, r, a, b)
                else:
                    _same_topk_up_to_ties(ids[i][:m], dist[i][:m], ids0[i][:m], dist0[i][:m], rtol=1e-5)
            want = orc.scan_distances(orc.AVX2, metric, dg.F32, qs[0], rows)
            m = cnt[0]
            _check_float_distances(dist[0][:m].astype(np.float32), want[ids[0][:m] - 1], dg.F32, metric, qs[0], rows[ids[0][:m] - 1])
        if dim == 768:                           # rows appended after a batch extend the shadow copy and the cached norms
            more = dg.corpus(dg.F32, 500, dim, 9600)
            more[7] = qs[20] if metric != dg.DOT else qs[20] * np.float32(50.0)    # a new best row for query 20
            c.append(more)
            monkeypatch.setenv("VG_BATCH_MFMA", "1")
            ids, dist, cnt = c.scan_topk_batch(metric, qs, 5)
            assert (n + 8) in ids[20].tolist(), (ids[20], dist[20])   # (dot: the Inf / 1e18 edge rows still rank before it)
            one_ids, one_dist = c.scan_topk(metric, qs[20], 5)
            if metric == dg.DOT:
                assert ids[20].tolist() == one_ids.tolist()
            else:        # (the 1e-3-scaled rows nearly tie: a swap at the k-th place is a summation-order effect)
                assert np.allclose(dist[20], one_dist, rtol=1e-5, atol=1e-7) and len(set(ids[20].tolist()) ^ set(one_ids.tolist())) <= 2
        c.close()


@pytest.mark.gpu
def test_batch_f32_filter_default_policy_and_selectivity_guard(pkg, monkeypatch):
    """f32 batches of a corpus the filter scan serves (switched on, large enough - here VG_SCAN_FILTER_MIN_MB=0) go through
    a matrix-core FILTER by default (round 6: the int8 filter at every batch size; below 2^20 rows - the copies-of-one-row corpus
    at the end - the bf16 filter): same lists as the f32 matrix-core kernel; scan_filter=0 / VG_F32_FILTER=0 keep the f32
    kernel; on rows the bound cannot separate (copies of one row) the guard sends the next batches back to the f32 kernel."""
    monkeypatch.delenv("VG_F32_FILTER", raising=False)
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    monkeypatch.setenv("VG_BATCH_MFMA", "1")
    # (large enough for the two-pass launch: with lists warming up from +Inf in 256 partitions a small corpus has most of its
    # pairs evaluated exactly - which is why the default needs the filter scan's size threshold)
    n, dim, nq, k = 2_200_003, 64, 70, 20
    rows = np.random.default_rng(9901).standard_normal((n, dim), dtype=np.float32)
    qs = dg.corpus(dg.F32, nq, dim, 9902)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for metric in (dg.DOT, dg.L2, dg.COSINE):
        c.set_scan_filter(-1)
        c.batch_filter_exact_evals()
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        evals = c.batch_filter_exact_evals()
        assert 0 < evals < nq * n // 256, (metric, evals)          # the filtered path ran and was selective
        c.set_scan_filter(0)                                       # the per-corpus switch: f32 matrix-core kernel
        ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)
        assert c.batch_filter_exact_evals() == 0
        monkeypatch.setenv("VG_F32_FILTER", "0")                   # the environment switch does the same
        c.set_scan_filter(-1)
        ids1, dist1, cnt1 = c.scan_topk_batch(metric, qs, k)
        assert c.batch_filter_exact_evals() == 0
        monkeypatch.delenv("VG_F32_FILTER")
        assert np.array_equal(cnt, cnt0) and np.array_equal(cnt, cnt1)
        for i in range(nq):
            _same_topk_up_to_ties(ids[i], dist[i], ids0[i], dist0[i], rtol=1e-5)
            _same_topk_up_to_ties(ids1[i], dist1[i], ids0[i], dist0[i], rtol=1e-5)
        one_ids, one_dist = c.scan_topk(metric, qs[3], k)          # the filtered batch carries the single scan's arithmetic
        _same_topk_up_to_ties(ids[3], dist[3], one_ids, one_dist, rtol=1e-6)
    # a handful of queries: single filter scans instead of one 256-query-wide matrix pass - the single scans' answers bit for bit
    c.set_scan_filter(-1)
    c.batch_filter_exact_evals(); c.filter_exact_evals()
    ids2, dist2, cnt2 = c.scan_topk_batch(dg.L2, qs[:3], k)
    assert c.batch_filter_exact_evals() == 0 and c.filter_exact_evals() > 0
    for i in range(3):
        one_ids, one_dist = c.scan_topk(dg.L2, qs[i], k)
        assert ids2[i].tolist() == one_ids.tolist() and dg.same_float_bits(dist2[i], one_dist)
    c.close()
    # copies of one row: every (query, row) pair has the same bound - nothing can be excluded
    del rows
    same = np.tile(dg.corpus(dg.F32, 1, dim, 9903), (20000, 1))
    c = pkg.Corpus(pkg.F32, dim)
    c.append(same)
    ids, dist, cnt = c.scan_topk_batch(dg.L2, qs, k)
    first = c.batch_filter_exact_evals()
    assert first > nq * 20000 // 256                               # ... the guard sees it ...
    ids2, dist2, cnt2 = c.scan_topk_batch(dg.L2, qs, k)
    assert c.batch_filter_exact_evals() == 0                       # ... and this batch took the f32 matrix-core kernel
    assert np.array_equal(ids, ids2) and np.allclose(dist, dist2, rtol=1e-5)
    assert ids[0].tolist() == list(range(1, k + 1))                # ties resolve by scan position on both paths
    c.close()


def _same_topk_up_to_ties(ids, dist, ids0, dist0, rtol=1e-6):
    """two result lists of one query agree: same distances (summation order only) and the same rows except where
    neighbouring distances tie within that tolerance"""
    assert len(ids) == len(ids0)
    assert np.allclose(dist, dist0, rtol=rtol, atol=1e-7), (dist[:5], dist0[:5])
    diff = np.nonzero(ids != ids0)[0]
    for j in diff:
        near = np.abs(dist0 - dist0[j]) <= rtol * np.abs(dist0[j]) + 1e-7
        assert ids[j] in ids0[near], (j, ids[j], ids0[near], dist0[near])


@pytest.mark.parametrize("vt", [dg.F16, dg.BF16])
@pytest.mark.parametrize("metric", (dg.DOT, dg.COSINE, dg.L2, dg.SQUARED_L2))
def test_batch_half_matrix_core_filter_vs_single_scans(pkg, orc, vt, metric, monkeypatch):
    """f16 / bf16 batches: the matrix cores (v_mfma_f32_32x32x16_f16 / _bf16) only FILTER; every (query, row) pair
    that can beat the current k-th best is re-evaluated with the single-query kernel's f64 arithmetic, so the results
    are the single scans' results - including rows / queries with Inf, NaN, zeros, huge and subnormal values."""
    for dim in (8, 100, 128, 384, 500, 512, 520, 768, 1000, 1024):       # > 512: the 4-wavefront kernels (rows up to 2 KiB)
        n = 4133
        rows = dg.corpus(vt, n, dim, 8200 + dim)
        _, edge = dg.edge_rows(vt, dim, 8300 + dim)                       # Inf / NaN / max / subnormal / zero rows
        rows[100:100 + len(edge)] = edge
        rows[2000] = rows[17]                                              # exact duplicates: ties
        rows[3000:3010] = dg.to_storage(vt, dg.storage_to_f64(vt, rows[3000:3010]).astype(np.float32) * np.float32(1e-3))
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for nq, k in ((1, 20), (7, 1), (300, 20), (40, 32)):
            qs = dg.corpus(vt, nq, dim, 8400 + dim + nq)
            eq = dg.edge_queries(vt, dim, 8500 + dim)
            for i, q in enumerate(eq[:min(len(eq), nq - 1)]):
                qs[1 + i] = q                                              # zero / Inf / NaN queries
            qs[0] = rows[17]                                               # a query that IS a row (distance 0, twice)
            monkeypatch.setenv("VG_BATCH_MFMA", "1")
            ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
            monkeypatch.setenv("VG_BATCH_MFMA", "0")
            ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)           # nq single-query scans
            assert np.array_equal(cnt, cnt0)
            for i in range(nq):
                m = cnt[i]
                _same_topk_up_to_ties(ids[i][:m], dist[i][:m], ids0[i][:m], dist0[i][:m])
            want = orc.scan_distances(orc.AVX2, metric, vt, qs[0], rows)
            m = cnt[0]
            _check_float_distances(dist[0][:m].astype(np.float32), want[ids[0][:m] - 1], vt, metric, qs[0], rows[ids[0][:m] - 1])
        c.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.F16, dg.BF16])
@pytest.mark.parametrize("metric", (dg.DOT, dg.COSINE, dg.L2, dg.SQUARED_L2))
def test_batch_long_rows_matrix_core_path(pkg, orc, vt, metric, monkeypatch):
    """rows of 1025 .. 3072 elements (1536- / 3072-dimensional embeddings): vg_batch_hl.hip splits the K dimension over the four
    wavefronts of a workgroup (each keeps a quarter of every query's row in registers), sums the partial scores in LDS, filters,
    and vg_batch_hx_kernel evaluates the candidates with the single scan's arithmetic - the lists are the single scans' lists,
    with Inf / NaN / zero / huge rows and queries, exact duplicates, an odd chunk count, rows appended after a batch."""
    for dim in (1032, 1536, 2000, 2048, 2056, 3072):              # 24 / 24 / 32 / 32 / 48 / 48 k-steps per wavefront; 2056: odd chunk count for halves
        n = 4133
        rows = dg.corpus(vt, n, dim, 7200 + dim)
        _, edge = dg.edge_rows(vt, dim, 7300 + dim)
        rows[100:100 + len(edge)] = edge
        rows[2000] = rows[17]
        if vt == dg.F32:
            rows[3000:3010] *= np.float32(1e-3)
        else:
            rows[3000:3010] = dg.to_storage(vt, dg.storage_to_f64(vt, rows[3000:3010]).astype(np.float32) * np.float32(1e-3))
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for nq, k in ((7, 1), (70, 20), (130, 32), (33, 5)):
            qs = dg.corpus(vt, nq, dim, 7400 + dim + nq)
            eq = dg.edge_queries(vt, dim, 7500 + dim)
            for i, q in enumerate(eq[:min(len(eq), nq - 1)]):
                qs[1 + i] = q                                              # zero / Inf / NaN queries: answered by single scans
            qs[0] = rows[17]
            monkeypatch.setenv("VG_BATCH_MFMA", "1")
            ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
            assert c.last_batch_path() == 4, (dim, nq, k, c.last_batch_path())
            monkeypatch.setenv("VG_BATCH_MFMA", "0")
            monkeypatch.setenv("VG_MULTI_SCAN", "0")
            ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)           # nq single-query scans
            assert c.last_batch_path() == 6
            monkeypatch.delenv("VG_MULTI_SCAN")
            assert np.array_equal(cnt, cnt0), (dim, nq, k)
            for i in range(nq):
                m = cnt[i]
                if metric == dg.DOT and vt == dg.F32:   # cancelling f32 sums: the bar is relative to the magnitude of the summed terms
                    both = np.intersect1d(ids[i][:m], ids0[i][:m])
                    assert len(both) >= m - 2, (dim, i)
                    for r in both:
                        a, b = dist[i][:m][ids[i][:m] == r][0], dist0[i][:m][ids0[i][:m] == r][0]
                        scale = float(np.abs(rows[r - 1].astype(np.float64) * qs[i].astype(np.float64)).sum())
                        assert (np.isinf(a) and a == b) or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-5 * (abs(b) + scale) + 1e-6, (dim, i, r, a, b)
                else:
                    _same_topk_up_to_ties(ids[i][:m], dist[i][:m], ids0[i][:m], dist0[i][:m], rtol=1e-5)
            want = orc.scan_distances(orc.AVX2, metric, vt, qs[0], rows)
            m = cnt[0]
            _check_float_distances(dist[0][:m].astype(np.float32), want[ids[0][:m] - 1], vt, metric, qs[0], rows[ids[0][:m] - 1])
        if dim == 1536:                          # rows appended after a batch extend the tile-major copy and the cached norms
            more = dg.corpus(vt, 500, dim, 7600)
            q20 = dg.storage_to_f64(vt, qs[20:21]).astype(np.float32)[0] if vt != dg.F32 else qs[20]
            more[7] = qs[20] if metric != dg.DOT else (dg.to_storage(vt, q20 * np.float32(50.0)) if vt != dg.F32 else q20 * np.float32(50.0))
            c.append(more)
            monkeypatch.setenv("VG_BATCH_MFMA", "1")
            ids, dist, cnt = c.scan_topk_batch(metric, qs, 5)
            assert c.last_batch_path() == 4
            one_ids, one_dist = c.scan_topk(metric, qs[20], 5)
            _same_topk_up_to_ties(ids[20][:cnt[20]], dist[20][:cnt[20]], one_ids, one_dist, rtol=1e-5)
            if metric != dg.DOT:                 # (dot: the Inf / huge edge rows rank before it)
                assert (n + 8) in ids[20].tolist(), (ids[20], dist[20])
        c.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.BF16])
def test_batch_long_rows_empty_vectors_and_unjudgeable_queries_stay_on_the_matrix_cores(pkg, vt):
    """a corpus where one row in fifty is all zeros (empty documents) and a batch that carries zero / NaN / Inf queries: the zero rows
    are judged by the filter like any other (their score is exactly 0; cosine gives them the reference's distance 1.0), the
    unjudgeable queries leave the matrix pass for scans of their own - neither fills the candidate regions: the batch stays on the
    long-row kernel and equals the single scans."""
    dim, n, nq, k = 1536, 300_000, 70, 10
    rows = dg.corpus(vt, n, dim, 8300)
    rows[::50] = 0
    qs = dg.corpus(vt, nq, dim, 8301)
    qs[4] = 0
    qf = dg.storage_to_f64(vt, qs[5:7]).astype(np.float32)
    qf[0, 10] = np.nan
    qf[1, 20] = np.inf
    qs[5:7] = qf if vt == dg.F32 else dg.to_storage(vt, qf)
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for metric in (dg.COSINE, dg.L2, dg.DOT):
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 4, (metric, c.last_batch_path())
        for i in list(range(0, nq, 9)) + [4, 5, 6]:
            one_ids, one_dist = c.scan_topk(metric, qs[i], k)
            assert cnt[i] == len(one_ids), (metric, i)
            m = cnt[i]
            if m:
                assert np.allclose(dist[i][:m], one_dist, rtol=1e-5, atol=1e-6, equal_nan=True), (metric, i)
                assert len(set(ids[i][:m].tolist()) ^ set(one_ids.tolist())) <= 2, (metric, i)
    # the negated query of a cosine batch: every non-empty row lies beyond distance 1, the empty ones AT 1.0 - they are its best rows
    neg = dg.to_storage(vt, -dg.storage_to_f64(vt, rows[1:2]).astype(np.float32)) if vt != dg.F32 else -rows[1:2]
    far = np.repeat(neg, 8, axis=0)
    ids, dist, cnt = c.scan_topk_batch(dg.COSINE, far, 3)
    one_ids, one_dist = c.scan_topk(dg.COSINE, far[0], 3)
    assert c.last_batch_path() == 4 and np.allclose(dist[0], one_dist, rtol=1e-5) and dist[0][0] <= 1.0 + 1e-5
    c.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.F16])
def test_batch_long_rows_tiny_corpora(pkg, vt):
    """fewer rows than one 32-row tile, fewer than k, one row more than a tile; more queries than one workgroup holds; k = 32 (the
    kernels' largest) and k = 33 (one more: the old path)"""
    dim = 1536
    for n in (1, 10, 31, 33, 70):
        rows = dg.corpus(vt, n, dim, 7800 + n)
        qs = dg.corpus(vt, 97, dim, 7900 + n)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for k in (1, 20, 32, 33):
            ids, dist, cnt = c.scan_topk_batch(dg.L2, qs, k)
            assert c.last_batch_path() == (4 if k <= 32 else (5 if vt == dg.F32 else 6)), (n, k, c.last_batch_path())
            assert np.all(cnt == min(k, n))
            for i in (0, 31, 32, 64, 96):
                one_ids, one_dist = c.scan_topk(dg.L2, qs[i], k)
                _same_topk_up_to_ties(ids[i][:cnt[i]], dist[i][:cnt[i]], one_ids, one_dist, rtol=1e-5)
        c.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.BF16])
def test_batch_long_rows_staged_passes_and_partitions(pkg, vt, monkeypatch):
    """a corpus large enough for the long-row kernel's bound pre-pass, several stages and more than one tile per partition, two
    query groups; a batch whose candidate pairs overflow a region (every row a duplicate of the query) is answered by one
    scan per query (rows this long have no multi-query scan) and the next batches stay there for a while."""
    dim, n, nq, k = 1536, 200_000, 100, 10
    rows = dg.corpus(vt, n, dim, 7700)
    qs = dg.corpus(vt, nq, dim, 7701)
    rows[150_000] = qs[3]                                                  # late best rows: they beat every earlier stage's threshold
    rows[199_999] = qs[64]
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for metric in (dg.COSINE, dg.L2, dg.DOT):
        monkeypatch.setenv("VG_BATCH_MFMA", "1")
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 4
        monkeypatch.setenv("VG_BATCH_MFMA", "0")
        ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)               # what such rows had before: the multi-query scan (f32) / one scan per query
        assert c.last_batch_path() == (5 if vt == dg.F32 else 6)
        assert np.array_equal(cnt, cnt0)
        for i in range(nq):
            _same_topk_up_to_ties(ids[i], dist[i], ids0[i], dist0[i], rtol=1e-5)
        if metric != dg.DOT:
            assert ids[3][0] == 150_001 and ids[64][0] == 200_000
    c.close()
    # unseparable data: 400k copies of one row - every pair passes the filter, the candidate regions overflow
    one = dg.corpus(vt, 1, dim, 7702)
    c = pkg.Corpus(vt, dim)
    for _ in range(4):
        c.append(np.repeat(one, 100_000, axis=0))
    monkeypatch.setenv("VG_BATCH_MFMA", "1")
    qs = np.repeat(one, 64, axis=0)
    ids, dist, cnt = c.scan_topk_batch(dg.L2, qs, 5)
    assert c.last_batch_path() in (5, 6) and ids[0].tolist() == [1, 2, 3, 4, 5] and np.all(dist == 0.0)
    ids, dist, cnt = c.scan_topk_batch(dg.L2, qs, 5)
    assert c.last_batch_path() in (5, 6)                                   # (cooling down)
    c.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.F16, dg.BF16])
def test_batch_half_workgroup_forms_agree(pkg, vt, monkeypatch):
    """the f16 / bf16 / f32-through-bf16 batch kernel comes as one 8-wavefront workgroup per CU or - rows up to 768 bytes, both
    workgroups' LDS fitting - as two of four (vg_batch_h_plan): the same keys out of both, bit for bit, over the row lengths that
    have both forms, a k where the second workgroup no longer fits (the plan must fall back by itself) and a batch too small for it."""
    monkeypatch.setenv("VG_F32_FILTER", "1")
    monkeypatch.setenv("VG_BATCH_MFMA", "1")
    rng = np.random.default_rng(17)
    for dim in ((8, 100, 192) if vt == dg.F32 else (8, 100, 256, 384)):
        n = 70_000
        rows = dg.corpus(vt, n, dim, 9100 + dim)
        rows[5000] = rows[17]
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for metric in (dg.DOT, dg.L2, dg.COSINE):
            for nq, k in ((300, 20), (513, 27), (300, 32), (100, 10)):
                qs = dg.corpus(vt, nq, dim, 9200 + dim + nq)
                qs[0] = rows[17]
                got = {}
                for form in ("8", "4", "", "split"):
                    monkeypatch.delenv("VG_BATCH_H_SPLIT", raising=False)
                    if form == "split":                            # the filter kernel + exact-evaluation kernel pair (vg_batch_h.hip, FILTER kind)
                        monkeypatch.delenv("VG_BATCH_H_WAVES", raising=False)
                        monkeypatch.setenv("VG_BATCH_H_SPLIT", "1")
                    elif form:
                        monkeypatch.setenv("VG_BATCH_H_WAVES", form)
                    else:
                        monkeypatch.delenv("VG_BATCH_H_WAVES", raising=False)
                    got[form] = c.scan_topk_batch(metric, qs, k)
                monkeypatch.delenv("VG_BATCH_H_SPLIT", raising=False)
                for form in ("4", "", "split"):
                    for a, b in zip(got["8"], got[form]):
                        assert np.array_equal(a, b), (dg.TYPE_NAMES[vt], dim, metric, nq, k, form)
        c.close()


@pytest.mark.parametrize("vt", [dg.F16, dg.BF16])
def test_batch_half_two_pass_launch_and_partitions(pkg, vt, monkeypatch):
    """enough rows for the two-pass launch (pre-pass thresholds) and many partitions; clustered data so that the
    filter sees near-duplicates of the queries"""
    dim, n, nq, k = 64, 2_200_000, 260, 10
    rng = np.random.default_rng(91)
    centers = rng.standard_normal((50, dim), dtype=np.float32)
    rows = dg.to_storage(vt, centers[rng.integers(0, 50, n)] + 0.05 * rng.standard_normal((n, dim), dtype=np.float32))
    qs = dg.to_storage(vt, centers[rng.integers(0, 50, nq)] + 0.05 * rng.standard_normal((nq, dim), dtype=np.float32))
    c = pkg.Corpus(vt, dim, capacity=n)
    c.append(rows)
    for metric in (dg.L2, dg.COSINE, dg.DOT):
        monkeypatch.setenv("VG_BATCH_MFMA", "1")
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert np.all(cnt == k)
        for i in range(0, nq, 13):
            one_ids, one_dist = c.scan_topk(metric, qs[i], k)
            _same_topk_up_to_ties(ids[i], dist[i], one_ids, one_dist)
    c.close()


# ------------------------------------------------------------------------------------------------- f16 / bf16

@pytest.mark.parametrize("vt", [dg.F16, dg.BF16])
@pytest.mark.parametrize("dim", (1, 5, 8, 13, 64, 100, 384, 768, 1024, 2048))
def test_half_types_finite_rows_vs_oracle(pkg, orc, vt, dim):
    """finite data: f64 accumulation like distance-avx2.c, only the summation order differs -> <= 1e-5 relative
    (in practice the float results are bit-identical except for rare last-ulp roundings)."""
    n = 2000
    rows = dg.corpus(vt, n, dim, 500 + dim)
    q = dg.query(vt, dim, 501 + dim)
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for metric in dg.ALL_METRICS:
        want = orc.scan_distances(orc.AVX2, metric, vt, q, rows)
        got = c.scan_distances(metric, q)
        _check_float_distances(got, want, vt, metric, q, rows)
        ulp_off = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert ulp_off.max() <= (2 if metric in (dg.L2, dg.SQUARED_L2, dg.L1) else 1 << 30)
        ids, dist = c.scan_topk(metric, q, 20)
        oids, odist, _ = orc.topk_ordered(got, None, 20)
        assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist)
    c.close()


@pytest.mark.parametrize("vt", [dg.F16, dg.BF16])
@pytest.mark.parametrize("dim", (5, 8, 13, 16, 35, 384))
def test_half_types_special_values_bit_exact(pkg, orc, vt, dim):
    """rows / queries holding Inf, NaN, max-magnitude and subnormal lanes (block and tail positions) take the exact
    slow path: bit-identical to the reference's distance-avx2.c results, quirks included."""
    _, rows = dg.edge_rows(vt, dim, 1000 + dim)
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    special = np.zeros(rows.shape[0], dtype=bool)
    expo = 0x7C00 if vt == dg.F16 else 0x7F80
    special |= ((rows & expo) == expo).any(axis=1)
    for qi, q in enumerate(dg.edge_queries(vt, dim, 2000 + dim)):
        qspecial = bool(((q & expo) == expo).any())
        for metric in dg.ALL_METRICS:
            want = orc.scan_distances(orc.AVX2, metric, vt, q, rows)
            got = c.scan_distances(metric, q)
            sel = np.ones_like(special) if qspecial else special
            assert dg.same_float_bits(got[sel], want[sel]), (dg.TYPE_NAMES[vt], dg.METRIC_NAMES[metric], dim, qi,
                                                             got[sel], want[sel])
            _check_float_distances(got[~sel], want[~sel], vt, metric, q, rows[~sel])
            ids, dist = c.scan_topk(metric, q, 64)
            oids, odist, _ = orc.topk_ordered(got, None, 64)
            assert ids.tolist() == oids.tolist()
    c.close()


# ------------------------------------------------------------------------------------------------- long rows

@pytest.mark.parametrize("vt,dim", [(dg.F32, 2304), (dg.F32, 5000), (dg.U8, 10000), (dg.I8, 8200), (dg.F16, 4000),
                                    (dg.BF16, 4100), (dg.F16, 3200), (dg.F32, 20000)])
def test_long_row_kernel_vs_oracle(pkg, orc, vt, dim):
    """rows with more 16-byte chunks than a register-resident shape covers go through vg_scan_long_kernel
    (query in LDS, one row per wavefront, sliced)."""
    n = 300
    rows = dg.corpus(vt, n, dim, 700 + dim)
    q = dg.query(vt, dim, 701 + dim)
    if vt in (dg.F16, dg.BF16):                       # a few special rows through the slow path as well
        rows[5, dim - 1] = dg.F16_INF if vt == dg.F16 else dg.BF_INF
        rows[9, 3] = dg.F16_NAN if vt == dg.F16 else dg.BF_NAN
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    assert "long" in c.kernel_name(dg.L2)
    for metric in dg.ALL_METRICS:
        want = orc.scan_distances(orc.AVX2, metric, vt, q, rows)
        got = c.scan_distances(metric, q)
        if vt in (dg.U8, dg.I8):
            assert dg.same_float_bits(got, want), (vt, metric, dim)
        else:
            sp = np.zeros(n, dtype=bool)
            if vt in (dg.F16, dg.BF16):
                sp[[5, 9]] = True
                assert dg.same_float_bits(got[sp], want[sp]), (vt, metric, got[sp], want[sp])
            _check_float_distances(got[~sp], want[~sp], vt, metric, q, rows[~sp])
        ids, dist = c.scan_topk(metric, q, 20)
        oids, odist, _ = orc.topk_ordered(got, None, 20)
        assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist)
    c.close()


def test_long_row_kernel_forced_on_ordinary_rows(pkg, orc, monkeypatch):
    """the long-row kernel must agree with the register-resident kernel on rows both can handle"""
    dim, n = 384, 3000
    rows = dg.corpus(dg.I8, n, dim, 41)
    q = dg.query(dg.I8, dim, 42)
    c = pkg.Corpus(pkg.I8, dim)
    c.append(rows)
    a = c.scan_distances(dg.COSINE, q)
    ida, da = c.scan_topk(dg.COSINE, q, 33)
    monkeypatch.setenv("VG_FORCE_LONG", "1")
    assert "long" in c.kernel_name(dg.COSINE)
    b = c.scan_distances(dg.COSINE, q)
    idb, db = c.scan_topk(dg.COSINE, q, 33)
    assert dg.same_float_bits(a, b) and ida.tolist() == idb.tolist() and np.array_equal(da, db)
    c.close()


# ------------------------------------------------------------------------------------------------- GPU quantization

@pytest.mark.parametrize("dim", [77, 200, 384, 700, 2100])
@pytest.mark.parametrize("vt", dg.ALL_TYPES)
def test_gpu_minmax_and_quantize_bit_exact(pkg, orc, vt, dim):
    """vector_quantize's two passes as kernels over the staged corpus: min / max / any-negative and every quantized
    byte must equal the pinned oracle (= the reference's host arithmetic, sqlite-vector.c:495-757, :1210-1268),
    including NaN / Inf / huge elements.  (The dims walk the kernels' chunks-per-lane instantiations, 1 .. 8, and rows longer than
    one round of 128 chunks; all of them end in a partly filled chunk for some type.)"""
    n = 1500 if dim == 77 else 1001
    rows = dg.corpus(vt, n, dim, 61)
    if vt == dg.F32:
        rows[3, 5] = np.float32(np.nan); rows[4, 6] = np.float32(3e9); rows[5, 7] = np.float32(-np.inf); rows[6, 8] = np.float32(-3e9)
    if vt in (dg.F16, dg.BF16):
        rows[3, 5] = dg.F16_NAN if vt == dg.F16 else dg.BF_NAN
        rows[4, 6] = dg.F16_INF if vt == dg.F16 else dg.BF_INF
        rows[5, 7] = dg.F16_NINF if vt == dg.F16 else dg.BF_NINF
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    finite = rows.copy()
    qt, scale, offset = orc.quant_params(vt, finite, 0)
    lo, hi, neg = c.minmax()
    f = dg.storage_to_f64(vt, rows)
    with np.errstate(invalid="ignore"):
        want_lo = np.float32(np.nanmin(np.where(np.isnan(f), np.inf, f)))
        want_hi = np.float32(np.nanmax(np.where(np.isnan(f), -np.inf, f)))
    assert np.float32(lo) == max(want_lo, np.float32(-3.4028235e38)) or (np.isinf(want_lo) and lo == want_lo)
    assert np.float32(hi) == min(want_hi, np.float32(3.4028235e38)) or (np.isinf(want_hi) and hi == want_hi)
    assert neg == bool((f < 0).any())
    for qtype, sc, off in ((pkg.QUANT_U8, 37.5, -1.25), (pkg.QUANT_S8, 21.0, 0.0), (qt, scale, offset)):
        if not np.isfinite(sc):
            continue
        got = c.quantize_rows(sc, off, qtype)
        want = np.stack([orc.quantize(vt, rows[i], off, sc, qtype).view(np.uint8) for i in range(n)])
        assert np.array_equal(got, want), (dg.TYPE_NAMES[vt], qtype, np.argwhere(got != want)[:5])
        part = c.quantize_rows(sc, off, qtype, row0=700, n_rows=300)
        assert np.array_equal(part, want[700:1000])
    c.close()


@pytest.mark.parametrize("dim", [5, 77, 390, 1030])
@pytest.mark.parametrize("vt", dg.ALL_TYPES)
def test_gpu_minmax_ignores_the_row_padding(pkg, vt, dim):
    """rows are padded with zero bytes to a multiple of 16 (and lanes without a chunk read zeros): none of that may become the
    minimum of a corpus whose elements are all >= 3, or the maximum of one whose elements are all <= -3"""
    rng = np.random.default_rng(dim)
    n = 257
    for sign in (1, -1):
        if vt == dg.U8 and sign < 0:
            continue
        if vt in (dg.F32, dg.F16, dg.BF16):
            rows = dg.to_storage(vt, (sign * (np.abs(rng.standard_normal((n, dim), dtype=np.float32)) + 3.0)).astype(np.float32))
        elif vt == dg.U8:
            rows = rng.integers(3, 256, (n, dim)).astype(np.uint8)
        else:
            rows = (sign * rng.integers(3, 128, (n, dim))).astype(np.int8)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        lo, hi, neg = c.minmax()
        f = dg.storage_to_f64(vt, rows)
        assert lo == float(f.min()) and hi == float(f.max()) and neg == (sign < 0), (dg.TYPE_NAMES[vt], dim, sign, lo, hi, f.min(), f.max())
        c.close()


def test_device_appends_and_cross_stream_scan(pkg, orc):
    """append_device with the corpus' own padded stride and garbage in the gap; then a scan on a caller stream that
    was launched straight after an (asynchronous) host append must see every appended row."""
    import torch
    dim, n = 100, 70_000
    rows = dg.corpus(dg.I8, n, dim, 77)
    q = dg.query(dg.I8, dim, 78)
    want = orc.scan_distances(orc.AVX2, dg.SQUARED_L2, dg.I8, q, rows)
    oids, odist, _ = orc.topk_ordered(want, np.arange(1, n + 1, dtype=np.int64), 10)

    padded = np.full((n, 112), 0x5A, dtype=np.uint8)
    padded[:, :dim] = rows.view(np.uint8)
    t = torch.from_numpy(padded).cuda()
    c = pkg.Corpus(pkg.I8, dim)
    c.append_device(t.data_ptr(), n, 112)
    ids, dist = c.scan_topk(dg.SQUARED_L2, q, 10)
    assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist)
    c.close()

    c = pkg.Corpus(pkg.I8, dim)
    qpad = np.zeros(112, dtype=np.uint8)
    qpad[:dim] = q.view(np.uint8)
    dq = torch.from_numpy(qpad).cuda()
    keys = torch.zeros(64, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    c.append(rows)                                              # host append: only enqueued on the corpus stream
    c.scan_topk_device(dg.SQUARED_L2, dq.data_ptr(), 10, keys.data_ptr(), st.cuda_stream)
    st.synchronize()
    k = keys.cpu().numpy().view(np.uint64)[:10]
    pos = (k & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert (pos + 1).tolist() == oids.tolist()
    c.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.BF16])
def test_in_process_shards_long_row_batches(pkg, vt):
    """a batch over 1536-dimensional rows dealt over three logical shards: every shard answers through the long-row matrix-core
    kernel (vg_batch_hl.hip) and the merged lists are the single corpus' lists - also in the reference's result order (bf16: the
    batch keys are the single scan's floats, one more slot and a look for ties; f32: query by query)."""
    dim, n, nq, k = 1536, 9_000, 40, 10
    rows = dg.corpus(vt, n, dim, 63)
    rows[7000] = rows[11]                                   # an exact duplicate in another shard: a tie at distance 0
    qs = dg.corpus(vt, nq, dim, 64)
    qs[3] = rows[11]
    one = pkg.Corpus(vt, dim)
    one.append(rows)
    sh = pkg.Shards(vt, dim, [0, 0, 0], block_rows=1000)
    sh.append(rows)
    for metric in (dg.COSINE, dg.L2):
        a = one.scan_topk_batch(metric, qs, k)
        assert one.last_batch_path() == 4
        b = sh.scan_topk_batch(metric, qs, k)
        assert np.array_equal(a[2], b[2])
        for i in range(nq):
            _same_topk_up_to_ties(b[0][i], b[1][i], a[0][i], a[1][i], rtol=1e-5)
        assert b[0][3][:2].tolist() == [12, 7001]
    for c in (one, sh):
        c.set_tie_order(pkg.TIE_REFERENCE)
    a = one.scan_topk_batch(dg.L2, qs, k)
    b = sh.scan_topk_batch(dg.L2, qs, k)
    for i in range(nq):
        one_ids, one_d = one.scan_topk(dg.L2, qs[i], k)
        assert a[0][i].tolist() == one_ids.tolist() and b[0][i].tolist() == one_ids.tolist(), i
        assert np.array_equal(a[1][i], one_d) and np.array_equal(b[1][i], one_d), i
    one.close(); sh.close()


@pytest.mark.parametrize("threads", (0, 1))
@pytest.mark.parametrize("vt,metric", [(dg.F32, dg.L2), (dg.U8, dg.COSINE), (dg.I8, dg.L1), (dg.F32, dg.DOT)])
def test_in_process_shards_equal_a_single_corpus(pkg, orc, vt, metric, threads, monkeypatch):
    """vg_shards (one logical corpus dealt block-cyclically over several devices of one process - here three logical
    shards on device 0): every call returns bit-for-bit what one corpus holding all rows returns.  threads = 1: each query's
    per-shard enqueue + collect runs on the handle's persistent host threads, the way it does by default on several devices."""
    monkeypatch.setenv("VECTORGPU_SHARD_THREADS", str(threads))
    dim, n, block = 72, 20_011, 1000                       # ragged last block; low-entropy rows force distance ties
    rows = dg.corpus(vt, n, dim, 61, low_entropy=vt != dg.F32)
    q = dg.query(vt, dim, 62, low_entropy=vt != dg.F32)
    rowids = np.arange(n, dtype=np.int64) * 3 - 1000
    one = pkg.Corpus(vt, dim)
    one.append(rows, rowids)
    sh = pkg.Shards(vt, dim, [0, 0, 0], block_rows=block)
    assert sh.threaded == bool(threads)
    sh.reserve(n)
    for r0 in range(0, n, 3333):                            # appends that straddle block boundaries
        sh.append(rows[r0:r0 + 3333], rowids[r0:r0 + 3333])
    assert sh.rows == n and sum(sh.shard_rows()) == n and min(sh.shard_rows()) > 6000
    assert [sh.rowid_at(p) for p in (0, 999, 1000, 2999, 3000, n - 1)] == [int(rowids[p]) for p in (0, 999, 1000, 2999, 3000, n - 1)]
    for k in (1, 20, 64, 200):
        a_ids, a_d = one.scan_topk(metric, q, k)
        b_ids, b_d = sh.scan_topk(metric, q, k)
        assert a_ids.tolist() == b_ids.tolist() and np.array_equal(a_d, b_d), k
    assert dg.same_float_bits(one.scan_distances(metric, q), sh.scan_distances(metric, q))
    qs = np.stack([dg.query(vt, dim, 70 + i, low_entropy=vt != dg.F32) for i in range(5)])
    a = one.scan_topk_batch(metric, qs, 10)
    b = sh.scan_topk_batch(metric, qs, 10)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if vt == dg.F32:
        assert one.minmax() == sh.minmax()
        lo, hi, _ = one.minmax()
        scale = 255.0 / (hi - lo)
        assert np.array_equal(one.quantize_rows(scale, lo, pkg.QUANT_U8, 500, 4100), sh.quantize_rows(scale, lo, pkg.QUANT_U8, 500, 4100))
    # implicit rowids follow the global position; clear + refill works
    sh.clear()
    sh.append(rows[:2500])
    ids, _ = sh.scan_topk(metric, q, 5)
    one2 = pkg.Corpus(vt, dim)
    one2.append(rows[:2500])
    assert ids.tolist() == one2.scan_topk(metric, q, 5)[0].tolist()
    # the reference's persisted record format goes through the same dealing
    if vt in (dg.U8, dg.I8):
        rec = np.zeros((n, 8 + dim), dtype=np.uint8)
        rec[:, :8] = rowids.astype("<i8").view(np.uint8).reshape(n, 8)
        rec[:, 8:] = rows.view(np.uint8)
        sh.clear()
        sh.append_records(rec, n)
        b_ids, b_d = sh.scan_topk(metric, q, 20)
        a_ids, a_d = one.scan_topk(metric, q, 20)
        assert a_ids.tolist() == b_ids.tolist() and np.array_equal(a_d, b_d)
    one.close(); one2.close(); sh.close()


@pytest.mark.parametrize("metric", (dg.DOT, dg.COSINE, dg.L2))
def test_batch_mfma_rows_with_nan_inf_at_the_start_of_a_partition(pkg, orc, metric):
    """rows whose score is NaN / +Inf never enter a list; when they sit in the first tile of a partition the list is
    not full after that tile and must keep accepting every finite row (threshold stays +Inf, not NaN)."""
    dim, n = 64, 40_000
    rows = dg.corpus(dg.F32, n, dim, 91)
    rows[0:30, 3] = np.nan                      # first tile of the first partition: only 2 finite rows
    rows[64:90, 5] = np.inf
    rows[5000:5031, 7] = np.nan
    qs = dg.corpus(dg.F32, 6, dim, 92)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    ids, dist, cnt = c.scan_topk_batch(metric, qs, 20)
    for i in range(6):
        one_ids, one_dist = c.scan_topk(metric, qs[i], 20)        # per-query kernel (oracle-checked elsewhere)
        assert cnt[i] == 20 and ids[i].tolist() == one_ids.tolist()
        assert np.allclose(dist[i], one_dist, rtol=1e-5, atol=1e-5)
    c.close()


@pytest.mark.parametrize("metric", (dg.L2, dg.SQUARED_L2))
def test_batch_l2_near_duplicates_keep_the_relative_bar(pkg, orc, metric):
    """batched L2 filters with |q|^2 + |x|^2 - 2<q,x> on the matrix cores, which cancels catastrophically when a row
    almost equals the query; survivors are re-evaluated with the reference's sum (q-x)^2, so even distances 1e-6 of the
    norms away are within 1e-5 RELATIVE of the reference (an identity-only kernel is off by orders of magnitude)."""
    dim, n, nq, k = 384, 6000, 40, 10
    rng = np.random.default_rng(77)
    rows = dg.corpus(dg.F32, n, dim, 71)
    qs = dg.corpus(dg.F32, nq, dim, 72)
    for i in range(nq):                                  # plant near-duplicates of every query at several scales
        for j, eps in enumerate((1e-1, 1e-2, 1e-3, 1e-4, 0.0)):
            rows[(i * 131 + j * 1009) % n] = qs[i] + eps * rng.standard_normal(dim).astype(np.float32)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
    for i in range(nq):
        want = orc.scan_distances(orc.AVX2, metric, dg.F32, qs[i], rows)
        tol = REL_TOL * np.abs(want.astype(np.float64)) + 8 * np.finfo(np.float32).eps * 1.01     # clamp straddle only
        _assert_topk_close(ids[i][:cnt[i]], dist[i][:cnt[i]], want, k, tol)
        assert dist[i][0] <= 1e-3 * float(np.sqrt(dim))                  # the planted rows are what was found
    c.close()


def test_maximum_row_size_and_the_limit_beyond_it(pkg, orc):
    """the largest supported row is 128 KiB (the query must fit the CU's LDS next to the merge scratch): f32 dim
    32768 and u8 dim 131072 scan correctly; one element more is refused with a message, not a crash."""
    for vt, dim, metric in ((dg.F32, 32768, dg.L2), (dg.U8, 131072, dg.COSINE), (dg.BF16, 65536, dg.DOT)):
        rows = dg.corpus(vt, 70, dim, 5)
        q = dg.query(vt, dim, 6)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        got = c.scan_distances(metric, q)
        want = orc.scan_distances(orc.AVX2, metric, vt, q, rows)
        if vt == dg.U8:
            assert dg.same_float_bits(got, want)
        else:
            _check_float_distances(got, want, vt, metric, q, rows)
        ids, dist = c.scan_topk(metric, q, 5)
        order = np.lexsort((np.arange(70), got))[:5]
        assert ids.tolist() == (order + 1).tolist()
        c.close()
    with pytest.raises(pkg.VectorGpuError, match="128 KiB"):
        pkg.Corpus(pkg.F32, 32769)


def test_large_k_radix_select_equals_full_sort_and_oracle(pkg, orc, monkeypatch):
    """k > 64: radix select (3 histogram passes + gather + small sort) must return exactly what the full device sort
    and the oracle's ordered top-k return - with NaN / +-Inf distances, negative distances (dot), massive ties
    (every distance equal) and k beyond the number of finite rows."""
    dim, n = 32, 50_000
    rows = dg.corpus(dg.F32, n, dim, 81)
    rows[100:140, 3] = np.nan
    rows[200:230, 5] = np.inf
    rows[300:310, 7] = -np.inf
    q = dg.query(dg.F32, dim, 82)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for metric in (dg.L2, dg.DOT):
        alld = c.scan_distances(metric, q)
        for k in (65, 100, 1000, 20_000, n):
            oids, odist, _ = orc.topk_ordered(alld, None, k)
            monkeypatch.setenv("VG_RADIX_SELECT", "1")
            ids, dist = c.scan_topk(metric, q, k)
            monkeypatch.setenv("VG_RADIX_SELECT", "0")
            ids0, dist0 = c.scan_topk(metric, q, k)
            assert ids.tolist() == ids0.tolist() == oids.tolist() and np.array_equal(dist, dist0) and np.array_equal(dist, odist), (metric, k)
    c.close()
    monkeypatch.setenv("VG_RADIX_SELECT", "1")
    same = np.tile(dg.corpus(dg.U8, 1, 16, 83), (30_000, 1))            # every row identical: one giant tie class
    c = pkg.Corpus(pkg.U8, 16)
    c.append(same)
    ids, dist = c.scan_topk(dg.L2, dg.query(dg.U8, 16, 84), 500)
    assert ids.tolist() == list(range(1, 501)) and len(set(dist.tolist())) == 1
    c.close()


@pytest.mark.parametrize("vt", (dg.U8, dg.I8))
@pytest.mark.parametrize("dim", (16, 100, 384, 768, 1000, 1024, 1030, 1536, 2048))     # > 1024: the 4-wavefront kernels
def test_quantized_batch_on_integer_matrix_cores_is_bit_exact(pkg, orc, vt, dim):
    """uint8 / int8 batches run Q x C^T on the integer matrix cores (v_mfma_i32_32x32x32_i8; uint8 through the
    x - 128 identity).  Integer sums are exact, so every query's list must equal the single-query scan - itself
    bit-exact with distance-avx2.c - rowid for rowid and bit for bit, ties (low-entropy rows) included."""
    n = 4133
    for low in (False, True):
        rows = dg.corpus(vt, n, dim, 700 + dim, low_entropy=low)
        rows[17] = 0                                  # zero-norm row (cosine -> 1.0)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for metric in (dg.DOT, dg.COSINE, dg.L2, dg.SQUARED_L2):
            for nq, k in ((3, 20), (130, 1), (70, 32)):
                qs = dg.corpus(vt, nq, dim, 701 + dim + nq, low_entropy=low)
                ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
                for i in (0, 1, nq // 2, nq - 1):
                    one_ids, one_dist = c.scan_topk(metric, qs[i], k)
                    assert cnt[i] == len(one_ids) and ids[i][:cnt[i]].tolist() == one_ids.tolist(), (metric, low, nq, k, i)
                    assert np.array_equal(dist[i][:cnt[i]], one_dist), (metric, low, nq, k, i)
        # against the oracle directly for one query / metric
        want = orc.scan_distances(orc.AVX2, dg.COSINE, vt, qs[0], rows)
        oids, odist, _ = orc.topk_ordered(want, None, 20)
        ids, dist, cnt = c.scan_topk_batch(dg.COSINE, qs[:1], 20)
        assert ids[0].tolist() == oids.tolist() and np.array_equal(dist[0], odist)
        c.close()


@pytest.mark.parametrize("vt", (dg.F32, dg.U8))
def test_batch_with_zero_queries_and_exact_duplicate_rows(pkg, orc, vt):
    """an all-zero query makes EVERY row tie (cosine 1.0, dot 0): the answer is the first k rows in scan order, also
    across partitions and through the pre-pass rule "a tie with the k-th best never wins"; duplicated rows tie too."""
    dim, n, k = 64, 9000, 20
    rows = dg.corpus(vt, n, dim, 95)
    rows[4000:4040] = rows[10]                      # 40 exact copies of one row, far from the original
    qs = dg.corpus(vt, 5, dim, 96)
    qs[1] = 0
    qs[3] = rows[10]                                # its best hits are the 41 identical rows: ties by position
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for metric in (dg.COSINE, dg.DOT, dg.L2):
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        for i in range(5):
            one_ids, one_dist = c.scan_topk(metric, qs[i], k)
            assert cnt[i] == k and ids[i].tolist() == one_ids.tolist(), (metric, i)
            if vt == dg.U8:
                assert np.array_equal(dist[i], one_dist)
            else:
                assert np.allclose(dist[i], one_dist, rtol=1e-5, atol=1e-5)
    c.close()


def test_batch_larger_than_one_slice(pkg, monkeypatch):
    """batches beyond VG_BATCH_SLICE queries run slice by slice; the result does not depend on the slicing"""
    dim, n, k, nq = 32, 3000, 5, 700
    rows = dg.corpus(dg.F32, n, dim, 97)
    qs = dg.corpus(dg.F32, nq, dim, 98)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    a = c.scan_topk_batch(dg.DOT, qs, k)
    monkeypatch.setenv("VG_BATCH_SLICE", "256")
    b = c.scan_topk_batch(dg.DOT, qs, k)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    c.close()


@pytest.mark.parametrize("vt", (dg.U8, dg.F16))
def test_batch_staged_real_passes_with_ties_across_stage_boundaries(pkg, vt, monkeypatch):
    """>= 2^16 tiles: the real pass runs in stages over growing row ranges, each one starting from (and partition 0 carrying
    on) the merged lists of the rows before it (vg_batch_common.h).  Low-entropy rows make every list a chain of ties that
    straddles the stage boundaries; copies of one query's best row sit in the pre-pass range, in the first stage, exactly at
    stage boundaries and in the last stage.  Every growth setting and the single real pass must return the lists of the
    single-query scans - rowid for rowid, bit for bit for uint8."""
    dim, n, k, nq = 32, 2_300_000, 20, 40
    rng = np.random.default_rng(123)
    if vt == dg.U8:
        rows = rng.integers(0, 3, (n, dim)).astype(np.uint8)           # 3 levels: huge tie classes
        qs = rng.integers(0, 3, (nq, dim)).astype(np.uint8)
        qs[5] = rng.integers(0, 256, dim).astype(np.uint8)
        special = (rng.integers(0, 256, dim).astype(np.uint8) // 2 + 64)
    else:
        rows = dg.to_storage(vt, rng.integers(-1, 2, (n, dim)).astype(np.float32))
        qs = dg.to_storage(vt, rng.integers(-1, 2, (nq, dim)).astype(np.float32))
        special = dg.to_storage(vt, rng.standard_normal((1, dim), dtype=np.float32) * 3)[0]
    ntiles = (n + 31) // 32
    pre = ntiles // 32
    spots = [3, 32 * pre - 1, 32 * pre, 64 * pre - 1, 64 * pre, 64 * pre + 1, 128 * pre - 1, 128 * pre, 256 * pre, 512 * pre - 1,
             512 * pre, n - 2]
    spots += [int(x) for x in rng.integers(0, n, 14)]                   # 26 copies > k: the tie class itself is cut by position
    for s in spots:
        rows[min(s, n - 1)] = special
    qs[7] = special
    c = pkg.Corpus(vt, dim, capacity=n)
    c.append(rows)
    for metric in (dg.DOT, dg.COSINE, dg.L2):
        singles = [c.scan_topk(metric, qs[i], k) for i in range(nq)]
        for growth in ("200", "0", "130", "400"):
            monkeypatch.setenv("VG_BATCH_STAGES", growth)
            ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
            for i in range(nq):
                one_ids, one_dist = singles[i]
                assert cnt[i] == k, (metric, growth, i)
                if vt == dg.U8:
                    assert ids[i].tolist() == one_ids.tolist(), (metric, growth, i)
                    assert np.array_equal(dist[i], one_dist), (metric, growth, i)
                else:
                    _same_topk_up_to_ties(ids[i], dist[i], one_ids, one_dist)
                    if metric != dg.COSINE:                              # small integers: both paths are exact, ties go by position
                        assert ids[i].tolist() == one_ids.tolist(), (metric, growth, i)
    c.close()


@pytest.mark.parametrize("vt", [dg.F16, dg.BF16])
def test_batch_half_tile_major_copy_equals_row_major_gather(pkg, vt, monkeypatch):
    """f16 / bf16 batches stream a tile-major copy of the corpus (contiguous 1 KiB LDS-DMA pieces); VG_BATCH_TILE_MAJOR=0
    keeps the row-major gather.  Same lists either way, also after rows are appended behind a ragged last tile."""
    k, nq = 10, 70
    for dim in (24, 384, 500, 768):
        rows = dg.corpus(vt, 5000, dim, 4400 + dim)
        qs = dg.corpus(vt, nq, dim, 4401 + dim)
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("VG_BATCH_TILE_MAJOR", mode)
            c = pkg.Corpus(vt, dim)
            c.append(rows[:3333])                                # (not a multiple of 32)
            first = c.scan_topk_batch(dg.DOT, qs, k)
            c.append(rows[3333:])
            out[mode] = (first, [c.scan_topk_batch(m, qs, k) for m in (dg.DOT, dg.COSINE, dg.L2)])
            c.close()
        for a, b in zip([out["1"][0]] + out["1"][1], [out["0"][0]] + out["0"][1]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), dim


@pytest.mark.parametrize("vt,dim", ((dg.F32, 100), (dg.U8, 768), (dg.F16, 384), (dg.I8, 33)))
def test_patch_and_delete_rows_equal_a_fresh_corpus(pkg, orc, vt, dim, monkeypatch):
    """vg_corpus_patch_rows / vg_corpus_delete_rows / vg_corpus_find_rowid (what the extension applies after UPDATE / DELETE):
    after any mix of them the corpus answers like one staged from scratch with the surviving rows - single scans (plain and
    through the filter with its shadow copies and norms), batches (tile-major copies, row sums), rowids, scan-position ties."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    n = 50_003
    rows = dg.corpus(vt, n, dim, 7300 + dim)
    ids = np.arange(10, 10 + 3 * n, 3, dtype=np.int64)                 # ascending, not contiguous
    c = pkg.Corpus(vt, dim)
    c.append(rows[:30000], ids[:30000])
    c.append(rows[30000:], ids[30000:])
    q = dg.query(vt, dim, 7400 + dim)
    qs = dg.corpus(vt, 40, dim, 7500 + dim)
    metric = dg.L2
    c.scan_topk(metric, q, 20)                                         # derived data exists before the edits
    c.scan_topk_batch(dg.DOT, qs, 20)
    assert c.find_rowid(int(ids[12345])) == 12345 and c.find_rowid(11) == -1 and c.find_rowid(5) == -1
    rng = np.random.default_rng(7600 + dim)
    keep = np.ones(n, bool)
    cur_rows, cur_ids = rows.copy(), ids.copy()
    for step in range(4):
        # patches: some rows become copies of the query (new best rows, ties among themselves) or fresh random rows
        pos = rng.permutation(len(cur_ids))[:300]
        new = dg.corpus(vt, 300, dim, 7700 + dim + step)
        new[:5] = q
        c.patch_rows(pos, new)
        cur_rows[pos] = new
        # deletions: a run at the front, singles, a run at the very end
        m = len(cur_ids)
        dele = np.unique(np.concatenate([np.arange(0, 7), rng.permutation(m)[:200], np.arange(m - 3, m)])).astype(np.int64)
        c.delete_rows(dele)
        mask = np.ones(m, bool)
        mask[dele] = False
        cur_rows, cur_ids = cur_rows[mask], cur_ids[mask]
        assert c.rows == len(cur_ids)
        fresh = pkg.Corpus(vt, dim)
        fresh.append(cur_rows, cur_ids)
        for mt in (dg.L2, dg.COSINE, dg.DOT):
            for filt in (1, 0):
                c.set_scan_filter(filt); fresh.set_scan_filter(filt)
                a_ids, a_d = c.scan_topk(mt, q, 20)
                b_ids, b_d = fresh.scan_topk(mt, q, 20)
                assert a_ids.tolist() == b_ids.tolist() and dg.same_float_bits(a_d, b_d), (vt, step, mt, filt)
            ai, ad, ac = c.scan_topk_batch(mt, qs, 20)
            bi, bd, bc = fresh.scan_topk_batch(mt, qs, 20)
            assert np.array_equal(ai, bi) and np.array_equal(ac, bc) and dg.same_float_bits(ad, bd), (vt, step, mt)
        assert c.find_rowid(int(cur_ids[len(cur_ids) // 2])) == len(cur_ids) // 2
        fresh.close()
    more = dg.corpus(vt, 100, dim, 7900 + dim)                         # appends keep working behind the edits
    c.append(more, np.arange(10**7, 10**7 + 100, dtype=np.int64))
    a_ids, _ = c.scan_topk(metric, more[7].copy(), 1)
    assert a_ids[0] == 10**7 + 7
    c.close()


def test_shards_rccl_candidate_gather_equals_the_host_gather(pkg, orc):
    """vg_shards' two forms of the candidate exchange (include/vectorgpu.h vg_shards_set_gather): the host gather and ONE grouped
    ncclAllGather over RCCL on the scan streams.  Same keys, same merge: results bit-identical.  This box has one device, so the
    RCCL communicator has one rank (devices must be distinct) - the multi-rank form of the same call runs when a handle spans
    several devices; logical shards on one device silently keep the host gather."""
    n, dim, k = 300_000, 64, 20
    rows = dg.corpus(dg.U8, n, dim, 4242)
    qs = [dg.query(dg.U8, dim, 4300 + i) for i in range(4)]
    one = pkg.Shards(dg.U8, dim, [0])
    one.append(rows)
    want = [one.scan_topk(dg.COSINE, q, k) for q in qs]
    assert one.gather_stats() == {"host": 0, "rccl": 0, "rccl_serving": False}          # (a single shard forwards to its corpus)
    one.set_gather("rccl")
    for q, (w_ids, w_d) in zip(qs, want):
        ids, d = one.scan_topk(dg.COSINE, q, k)
        assert ids.tolist() == w_ids.tolist() and np.array_equal(d, w_d)
    st = one.gather_stats()
    assert st["rccl"] == len(qs) and st["rccl_serving"], st
    # tie_order = reference rides on the same gather (one more key per shard)
    one.set_tie_order(pkg.TIE_REFERENCE)
    ids, d = one.scan_topk(dg.L2, qs[0], k)
    r_ids, r_d = orc.topk_reference(orc.scan_distances(orc.AVX2, dg.L2, dg.U8, qs[0], rows), None, k)
    assert ids.tolist() == r_ids.tolist() and np.array_equal(d, r_d)
    one.close()
    # three logical shards on ONE device: RCCL cannot serve (ranks must sit on distinct devices) - the handle keeps the host gather
    three = pkg.Shards(dg.U8, dim, [0, 0, 0], block_rows=4099)
    three.append(rows)
    three.set_gather("rccl")
    for q, (w_ids, w_d) in zip(qs, want):
        ids, d = three.scan_topk(dg.COSINE, q, k)
        assert ids.tolist() == w_ids.tolist() and np.array_equal(d, w_d)
    st = three.gather_stats()
    assert st["host"] == len(qs) and st["rccl"] == 0 and not st["rccl_serving"], st
    three.close()
