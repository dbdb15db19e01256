"""The int8 matrix-core filter for f32 batches (vg_batch_q8.hip): Q x C^T on v_mfma_i32_32x32x32_i8 over the int8 shadow copy decides
which (query, row) pairs CAN beat a query's k-th best; the pairs that pass carry the single scan's f32 arithmetic.  A sound filter
changes nothing in the answer, so the bars are

  * against the bf16-filter path (vg_batch_h.hip, same exact-evaluation kernel behind another filter): rowids, distance BITS and counts
    identical for every query the filters judge;
  * against the oracle's distances (distance-avx2.c order restated in oracle/oracle.c): within the f32 bar;
  * adversarial rows for the bound (exact duplicates of a query, rows scaled to the edge of the int8 grid, one huge element, constant
    rows, zero rows, Inf / NaN rows, huge / tiny norms) must not be lost.

The reference has no batched entry point: a batch's oracle is Q independent vFullScanRun calls (sqlite-vector.c:2071-2113)."""
import numpy as np
import pytest

import datagen as dg
from test_gpu_scan import _check_float_distances, _same_topk_up_to_ties, pkg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _adversarial(rows, qs, rng):
    """plant rows the int8 bound has to be honest about"""
    n, dim = rows.shape
    rows[11] = qs[0]                                           # exact duplicates of queries (distance 0 / the best dot)
    rows[n - 1] = qs[1]                                        # ... in the last (ragged) tile
    rows[5000] = qs[2] * np.float32(3.0)
    rows[5001] = -qs[2]
    rows[6000] = 0.0                                           # zero row: cosine 1.0, dot 0
    rows[6001] = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)      # constant row whose elements all round the same way
    rows[6002] = 0.0
    rows[6002, dim // 2] = np.float32(1000.0)                  # one huge element: every other one quantizes to 0
    rows[6003] = qs[3]
    rows[6003, 0] = np.float32(400.0)                          # a good match whose scale is ruined by one element
    rows[6004] = rng.standard_normal(dim).astype(np.float32) * np.float32(1e-12)    # tiny norm (below the judged range squared)
    rows[6005] = rng.standard_normal(dim).astype(np.float32) * np.float32(1e12)     # huge norm
    rows[6006, 3] = np.float32(np.nan)
    rows[6007, dim - 1] = np.float32(np.inf)
    rows[6008] = np.float32(-np.inf)
    rows[7000:7040] = qs[4] + rng.standard_normal((40, dim)).astype(np.float32) * np.float32(1e-3)   # a cluster of near-duplicates
    return rows


@pytest.mark.parametrize("dim", (33, 100, 128, 384, 400, 512))
def test_int8_filter_batch_equals_the_bf16_filter_batch(pkg, orc, dim, monkeypatch):
    rng = np.random.default_rng(7100 + dim)
    n = 70_001 if dim > 128 else 150_003
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    nq = 513                                                   # two query groups, the second one nearly all padding
    qs = rng.standard_normal((nq, dim), dtype=np.float32)
    qs[5] = 0.0                                                # queries the filter cannot judge: a single scan each
    qs[6, 1] = np.float32(np.nan)
    qs[7] *= np.float32(1e25)
    qs[8] *= np.float32(1e-25)
    qs[9, 0] = np.float32(50.0)                                # one large element: a coarse int8 image of the rest
    rows = _adversarial(rows, qs, rng)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for metric in (dg.DOT, dg.COSINE, dg.L2, dg.SQUARED_L2):
        for k in ((20, 1, 32) if metric == dg.DOT else (20,)):
            monkeypatch.setenv("VG_BATCH_Q8", "1")
            ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
            assert c.last_batch_path() == 7, (dim, metric, k, c.last_batch_path(), c.batch_q8_status())
            monkeypatch.setenv("VG_BATCH_Q8", "0")
            monkeypatch.setenv("VG_F32_FILTER", "1")
            ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)
            assert c.last_batch_path() == 3
            monkeypatch.delenv("VG_F32_FILTER")
            assert np.array_equal(cnt, cnt0), (dim, metric, k)
            # (L2 batches share one int8 scale: query 9, one element of 50, does not fit it and is answered by a single scan too)
            unjudged = (5, 6, 7, 8) + ((9,) if metric in (dg.L2, dg.SQUARED_L2) else ())
            judged = [i for i in range(nq) if i not in unjudged]
            for i in judged:
                m = cnt[i]
                assert ids[i][:m].tolist() == ids0[i][:m].tolist() and dg.same_float_bits(dist[i][:m], dist0[i][:m]), (dim, metric, k, i)
            for i in unjudged:                                 # single scans here, in-kernel evaluation there: another f32 summation order
                m = cnt[i]
                if m:
                    fin = np.isfinite(dist0[i][:m])
                    assert np.array_equal(np.isfinite(dist[i][:m]), fin)
                    assert np.allclose(dist[i][:m][fin], dist0[i][:m][fin], rtol=1e-5, atol=1e-30), (dim, metric, k, i)
            # planted rows are found: the duplicate of query 0 / 1 is its best row under every metric but dot
            if metric != dg.DOT:
                assert ids[0][0] == 12 and ids[1][0] == n, (ids[0][:3], ids[1][:3])
            for i in (0, 3, 9, 40):                            # against the oracle's arithmetic
                m = cnt[i]
                want = orc.scan_distances(orc.AVX2, metric, dg.F32, qs[i], rows)
                _check_float_distances(dist[i][:m].astype(np.float32), want[ids[i][:m] - 1], dg.F32, metric, qs[i], rows[ids[i][:m] - 1])
                # and nothing better was left behind: the k-th distance returned bounds every other finite distance from below
                rest = np.delete(want, ids[i][:m] - 1)
                rest = rest[np.isfinite(rest)]
                tol = 1e-5 * (abs(float(dist[i][m - 1])) + (float(np.abs(qs[i]).sum()) * 4.0 if metric == dg.DOT else (1.0 if metric == dg.COSINE else 0.0)))
                assert m == k and (len(rest) == 0 or rest.min() >= dist[i][m - 1] - tol), (dim, metric, i)
    monkeypatch.delenv("VG_BATCH_Q8")
    c.close()


def test_int8_filter_default_policy_appends_and_unselective_rows(pkg, monkeypatch):
    """default policy: EVERY batch (round 6; round 5: more than 256 queries) over a corpus the filter scans' policy covers (>= 2^20 rows)
    takes the int8 filter - a ragged 300-query batch (212 padding slots), 200 queries, 5 queries; VG_BATCH_Q8=0 keeps the bf16 filter;
    rows appended afterwards extend the tile-major copy; a corpus the bound cannot separate (copies of one row) overflows the pair
    regions, is answered by the other paths, and cools the int8 path down"""
    monkeypatch.delenv("VG_F32_FILTER", raising=False)
    monkeypatch.delenv("VG_BATCH_Q8", raising=False)
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    n, dim, k = 1_100_003, 64, 20
    rng = np.random.default_rng(7201)
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    qs = rng.standard_normal((300, dim), dtype=np.float32)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for metric in (dg.L2, dg.DOT, dg.COSINE):
        c.batch_filter_exact_evals()
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 7, (metric, c.last_batch_path(), c.batch_q8_status())
        evals = c.batch_filter_exact_evals()
        assert 0 < evals < 300 * n // 256, (metric, evals)          # the int8 bound is selective on this data
        # smaller batches take the same path (another padding, another sort order of the slots): the same bits for the same queries
        for sub in (200, 5):
            ids2, dist2, cnt2 = c.scan_topk_batch(metric, qs[:sub], k)
            assert c.last_batch_path() == 7, (metric, sub, c.last_batch_path())
            assert np.array_equal(ids[:sub], ids2) and dg.same_float_bits(dist[:sub], dist2), (metric, sub)
        # VG_BATCH_Q8=0: the bf16 filter (same exact-evaluation arithmetic: the same bits) - or, once ITS selectivity guard has seen this
        # small corpus' warm-up (lists starting at +Inf: one evaluation per 256 pairs is exceeded), the f32 matrix-core kernel
        monkeypatch.setenv("VG_BATCH_Q8", "0")
        ids2, dist2, cnt2 = c.scan_topk_batch(metric, qs[:200], k)
        monkeypatch.delenv("VG_BATCH_Q8")
        assert c.last_batch_path() in (1, 3)
        if c.last_batch_path() == 3:
            assert np.array_equal(ids[:200], ids2) and dg.same_float_bits(dist[:200], dist2)
        else:
            for i in range(200):
                _same_topk_up_to_ties(ids[i], dist[i], ids2[i], dist2[i], rtol=1e-5)
        one_ids, one_dist = c.scan_topk(metric, qs[7], k)
        _same_topk_up_to_ties(ids[7], dist[7], one_ids, one_dist, rtol=1e-6)
    more = rng.standard_normal((777, dim), dtype=np.float32)
    more[5] = qs[20]                                                 # a new best row for query 20
    c.append(more)
    ids, dist, cnt = c.scan_topk_batch(dg.L2, qs, k)
    assert c.last_batch_path() == 7 and ids[20][0] == n + 6 and dist[20][0] == 0.0
    c.close()
    del rows
    same = np.tile(rng.standard_normal((1, dim), dtype=np.float32), (70_000, 1))
    c = pkg.Corpus(pkg.F32, dim)
    c.append(same)
    monkeypatch.setenv("VG_BATCH_Q8", "1")
    ids, dist, cnt = c.scan_topk_batch(dg.L2, qs, k)
    assert c.last_batch_path() != 7                                  # every pair passes: the regions overflow, another path answers
    assert ids[0].tolist() == list(range(1, k + 1))                  # ties resolve by scan position
    ids2, dist2, cnt2 = c.scan_topk_batch(dg.L2, qs, k)
    assert c.last_batch_path() != 7 and np.array_equal(ids, ids2)    # (cooling down)
    c.close()


def test_int8_filter_after_patched_and_deleted_rows(pkg, monkeypatch):
    """row maintenance pulls the tile-major copy's watermark back (vg_corpus.hip: invalidate_derived_from): the next batch must see
    the patched rows and must not see the deleted ones"""
    monkeypatch.setenv("VG_BATCH_Q8", "1")
    n, dim, k = 80_000, 96, 10
    rng = np.random.default_rng(7301)
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    qs = rng.standard_normal((300, dim), dtype=np.float32)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    ids, dist, cnt = c.scan_topk_batch(dg.L2, qs, k)
    assert c.last_batch_path() == 7
    best = int(ids[0][0])
    c.patch_rows(np.array([100], dtype=np.int64), qs[1][None, :].copy())          # position 100 becomes query 1 itself
    c.delete_rows(np.array([best - 1], dtype=np.int64))                          # query 0's best row goes away
    ids2, dist2, cnt2 = c.scan_topk_batch(dg.L2, qs, k)
    assert c.last_batch_path() == 7
    assert dist2[1][0] == 0.0 and best not in ids2[0].tolist()
    fresh_rows = np.delete(np.concatenate([rows[:100], qs[1][None, :], rows[101:]]), best - 1, axis=0)
    f = pkg.Corpus(pkg.F32, dim)
    f.append(fresh_rows)
    ids3, dist3, cnt3 = f.scan_topk_batch(dg.L2, qs, k)
    assert dg.same_float_bits(dist2, dist3)
    c.close(); f.close()


@pytest.mark.parametrize("vt", (dg.F16, dg.BF16))
@pytest.mark.parametrize("dim", (100, 384))
def test_int8_filter_batch_over_half_precision_corpora(pkg, orc, vt, dim, monkeypatch):
    """f16 / bf16 corpora through the same int8 filter (the image is taken of the widened elements; the pairs that pass carry the single
    scan's arithmetic for the type, vg_batch_hx_kernel): rowids, distance bits and counts of the type's own matrix-core filter path."""
    rng = np.random.default_rng(7400 + dim + vt)
    n, nq, k = 90_001, 300, 20
    rows32 = rng.standard_normal((n, dim), dtype=np.float32)
    qs32 = rng.standard_normal((nq, dim), dtype=np.float32)
    qs32[5] = 0.0
    qs32[7] *= np.float32(300.0)
    qs32[9, 0] = np.float32(50.0)
    rows32 = _adversarial(rows32, qs32, rng)
    rows32[6005] = rng.standard_normal(dim).astype(np.float32) * np.float32(6e4 if vt == dg.F16 else 1e12)      # (f16: the largest norms it can hold)
    rows32[6004] = rng.standard_normal(dim).astype(np.float32) * np.float32(1e-4 if vt == dg.F16 else 1e-12)
    rows, qs = dg.to_storage(vt, rows32), dg.to_storage(vt, qs32)
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for metric in (dg.DOT, dg.COSINE, dg.L2):
        monkeypatch.setenv("VG_BATCH_Q8", "1")
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 7, (vt, dim, metric, c.last_batch_path(), c.batch_q8_status())
        monkeypatch.setenv("VG_BATCH_Q8", "0")
        ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 3
        assert np.array_equal(cnt, cnt0), (vt, dim, metric)
        for i in range(nq):
            m = cnt[i]
            assert ids[i][:m].tolist() == ids0[i][:m].tolist() and dg.same_float_bits(dist[i][:m], dist0[i][:m]), (vt, dim, metric, i)
        # ... which is the single scan's answer
        for i in (0, 3, 9, 40):
            one_ids, one_d = c.scan_topk(metric, qs[i], k)
            assert ids[i][:cnt[i]].tolist() == one_ids.tolist() and dg.same_float_bits(dist[i][:cnt[i]], one_d), (vt, dim, metric, i)
    monkeypatch.delenv("VG_BATCH_Q8")
    c.close()


@pytest.mark.parametrize("vt,dim", [(dg.F32, 600), (dg.F32, 1024), (dg.F32, 1400), (dg.F32, 1536), (dg.F16, 1536), (dg.BF16, 1040)])
def test_int8_filter_batch_over_long_rows(pkg, orc, vt, dim, monkeypatch):
    """rows of 513 .. 1536 elements: one query set per wavefront, eight wavefronts, the ring in K-parts of a tile (vg_batch_q8.hip, KS = 2 / 3)
    - the answers of the bf16 / f16 matrix-core paths for such rows (vg_batch_h.hip up to 1024 elements, the K-split vg_batch_hl.hip
    beyond), bit for bit, which are the single scan's"""
    rng = np.random.default_rng(7500 + dim + vt)
    n, nq, k = 66_003, 300, 20
    rows32 = rng.standard_normal((n, dim), dtype=np.float32)
    qs32 = rng.standard_normal((nq, dim), dtype=np.float32)
    qs32[5] = 0.0
    qs32[9, 0] = np.float32(50.0)
    rows32 = _adversarial(rows32, qs32, rng)
    if vt == dg.F16:
        rows32[6005] = rng.standard_normal(dim).astype(np.float32) * np.float32(1e3)
        rows32[6004] = rng.standard_normal(dim).astype(np.float32) * np.float32(1e-4)
    rows, qs = dg.to_storage(vt, rows32), dg.to_storage(vt, qs32)
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for metric in (dg.DOT, dg.COSINE, dg.L2):
        monkeypatch.setenv("VG_BATCH_Q8", "1")
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 7, (vt, dim, metric, c.last_batch_path(), c.batch_q8_status())
        monkeypatch.setenv("VG_BATCH_Q8", "0")
        monkeypatch.setenv("VG_F32_FILTER", "1")
        ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() in (3, 4), c.last_batch_path()
        monkeypatch.delenv("VG_F32_FILTER")
        assert np.array_equal(cnt, cnt0), (vt, dim, metric)
        unjudged = (5,) + ((9,) if metric == dg.L2 else ())
        for i in range(nq):
            m = cnt[i]
            if i in unjudged and vt == dg.F32:                 # (a single scan here, in-kernel evaluation there: another f32 summation order)
                assert np.allclose(dist[i][:m], dist0[i][:m], rtol=1e-5, atol=1e-30), (vt, dim, metric, i)
                continue
            assert ids[i][:m].tolist() == ids0[i][:m].tolist() and dg.same_float_bits(dist[i][:m], dist0[i][:m]), (vt, dim, metric, i)
        for i in (0, 3, 40):
            one_ids, one_d = c.scan_topk(metric, qs[i], k)
            assert ids[i][:cnt[i]].tolist() == one_ids.tolist(), (vt, dim, metric, i)
            if vt != dg.F32:
                assert dg.same_float_bits(dist[i][:cnt[i]], one_d)
            else:
                assert np.allclose(dist[i][:cnt[i]], one_d, rtol=1e-5, atol=1e-30)
    monkeypatch.delenv("VG_BATCH_Q8")
    c.close()


@pytest.mark.parametrize("k", (33, 64))
def test_int8_filter_lists_of_up_to_64(pkg, orc, k, monkeypatch):
    """lists of 33 .. 64 entries on the matrix cores (round 6: the int8 filter's exact-evaluation lists are as long as the single scans' -
    vector_full_scan_batch(..., 64)): every query against the ORACLE's distances over all rows (adversarial rows included), and against a
    single scan of the same query up to ties"""
    dim, n, nq = 384, 70_001, 513
    rng = np.random.default_rng(9100 + k)
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    qs = rng.standard_normal((nq, dim), dtype=np.float32)
    rows = _adversarial(rows, qs, rng)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    monkeypatch.setenv("VG_BATCH_Q8", "1")
    for metric in (dg.DOT, dg.COSINE, dg.L2):
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 7, (metric, k, c.last_batch_path(), c.batch_q8_status())
        assert np.all(cnt == k)
        for i in range(0, nq, 8):
            want = orc.scan_distances(orc.AVX2, metric, dg.F32, qs[i], rows)
            _check_float_distances(dist[i].astype(np.float32), want[ids[i] - 1], dg.F32, metric, qs[i], rows[ids[i] - 1])
            rest = np.delete(want, ids[i] - 1)
            rest = rest[np.isfinite(rest)]
            tol = 1e-5 * (abs(float(dist[i][k - 1])) + (float(np.abs(qs[i]).sum()) * 4.0 if metric == dg.DOT else (1.0 if metric == dg.COSINE else 0.0)))
            assert rest.min() >= dist[i][k - 1] - tol, (metric, k, i)          # nothing better was left behind
        for i in (0, 1, 2, 3, 4, 100):
            one_ids, one_dist = c.scan_topk(metric, qs[i], k)
            _same_topk_up_to_ties(ids[i], dist[i], one_ids, one_dist, rtol=1e-5)
    monkeypatch.delenv("VG_BATCH_Q8")
    c.close()


def test_int8_filter_adversarial_rows_against_the_oracle(pkg, orc, monkeypatch):
    """oracle-anchored, not GPU-vs-GPU: duplicates of queries, one huge element, constant / zero rows, Inf / NaN rows, 1e+-12
    norms, a cluster of near-duplicates - every 8th query's list of the int8-filter batch against orc.scan_distances over all 70 001 rows"""
    dim, n, nq, k = 384, 70_001, 513, 20
    rng = np.random.default_rng(9200)
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    qs = rng.standard_normal((nq, dim), dtype=np.float32)
    qs[9, 0] = np.float32(50.0)
    rows = _adversarial(rows, qs, rng)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    monkeypatch.setenv("VG_BATCH_Q8", "1")
    for metric in (dg.DOT, dg.COSINE, dg.L2, dg.SQUARED_L2):
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        assert c.last_batch_path() == 7
        for i in list(range(0, nq, 8)) + [1, 2, 3, 4, 9]:
            m = cnt[i]
            assert m == k, (metric, i, m)
            want = orc.scan_distances(orc.AVX2, metric, dg.F32, qs[i], rows)
            _check_float_distances(dist[i][:m].astype(np.float32), want[ids[i][:m] - 1], dg.F32, metric, qs[i], rows[ids[i][:m] - 1])
            rest = np.delete(want, ids[i][:m] - 1)
            rest = rest[np.isfinite(rest)]
            tol = 1e-5 * (abs(float(dist[i][m - 1])) + (float(np.abs(qs[i]).sum()) * 4.0 if metric == dg.DOT else (1.0 if metric == dg.COSINE else 0.0)))
            assert len(rest) == 0 or rest.min() >= dist[i][m - 1] - tol, (metric, i)
        if metric != dg.DOT:
            assert ids[0][0] == 12 and ids[1][0] == n, (ids[0][:3], ids[1][:3])    # the planted duplicates are every such query's best row
    monkeypatch.delenv("VG_BATCH_Q8")
    c.close()

@pytest.mark.parametrize("nq", (3, 64, 128))
def test_int8_filter_128_slot_form(pkg, nq, monkeypatch):
    """batches of up to 128 queries over short rows take the 128-slot form (four wavefronts x one query set, vg_batch_q8.hip): rowids, distance
    bits and counts of the bf16 filter batch, three metrics, two row lengths; 129 queries are back on the 256-slot form with the same answers"""
    k = 20
    for dim in (100, 384):
        rng = np.random.default_rng(9100 + dim + nq)
        n = 90_001
        rows = rng.standard_normal((n, dim), dtype=np.float32)
        rows[77] = rows[5]                                          # a duplicate: ties by position
        c = pkg.Corpus(pkg.F32, dim)
        c.append(rows)
        for nqq in (nq, nq + 1 if nq == 128 else nq):
            qs = rng.standard_normal((nqq, dim), dtype=np.float32)
            qs[0] = rows[5]
            for metric in (dg.DOT, dg.COSINE, dg.L2):
                monkeypatch.setenv("VG_BATCH_Q8", "1")
                ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
                assert c.last_batch_path() == 7, (dim, nqq, metric, c.last_batch_path(), c.batch_q8_status())
                monkeypatch.setenv("VG_BATCH_Q8", "0")
                monkeypatch.setenv("VG_F32_FILTER", "1")
                ids0, dist0, cnt0 = c.scan_topk_batch(metric, qs, k)
                assert c.last_batch_path() == 3
                monkeypatch.delenv("VG_F32_FILTER")
                assert np.array_equal(cnt, cnt0), (dim, nqq, metric)
                for i in range(nqq):
                    m = cnt[i]
                    assert m == k and ids[i][:m].tolist() == ids0[i][:m].tolist() and dg.same_float_bits(dist[i][:m], dist0[i][:m]), (dim, nqq, metric, i)
                if metric != dg.DOT:
                    assert ids[0][:2].tolist() == [6, 78], ids[0][:3]       # the row itself, then its duplicate (the later position loses the tie)
        monkeypatch.delenv("VG_BATCH_Q8")
        c.close()
