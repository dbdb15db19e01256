"""Out-of-core scans (vg_slabscan.hip: vg_slab_scan_*): a table handed over slab by slab - two slabs on the device, one filling while the
other is scanned - must answer exactly like ONE corpus holding all rows: rowids, distance bits and order, for tie_order = position
(ordered by (distance, scan position)) and tie_order = reference (the reference's slot algorithm, sqlite-vector.c:2022-2069 /
:2102-2106, whose state is carried from slab to slab the way the reference carries it from row to row), for k = 0 (every distance:
the *_stream functions) and for the persisted record format of the quantized scans (sqlite-vector.c:1296-1309)."""
import numpy as np
import pytest

import datagen as dg
from test_gpu_scan import pkg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("vt", (dg.F32, dg.U8, dg.I8, dg.F16))
@pytest.mark.parametrize("tie", (0, 1))
def test_slab_scan_equals_one_corpus(pkg, orc, vt, tie):
    dim, n = 40, 50_017
    rows = dg.corpus(vt, n, dim, 810 + vt, low_entropy=vt in (dg.U8, dg.I8))       # int types: distances tie constantly
    if vt == dg.F32:
        rows[::7] = np.rint(rows[::7])
        rows[20_000:20_040] = rows[3]                                               # exact duplicates across slabs
    rowids = np.arange(n, dtype=np.int64) * 5 - 77
    one = pkg.Corpus(vt, dim)
    one.append(rows, rowids)
    one.set_tie_order(tie)
    rng = np.random.default_rng(811)
    for metric in ((dg.L2, dg.COSINE, dg.L1) if vt != dg.F32 else (dg.L2, dg.DOT, dg.COSINE)):
        q = rows[int(rng.integers(0, n))].copy() if metric != dg.DOT else dg.query(vt, dim, 812)
        for k, slab in ((20, 7_000), (1, 50_017), (64, 49_999), (100, 3_333), (20, 60_000), (5, 1)):
            if slab == 1 and (metric != dg.L2 or tie):
                continue
            nn = n if slab > 1 else 300                                             # (one row per slab: a short table)
            if nn == n:
                want_ids, want_d = one.scan_topk(metric, q, k)
            else:
                few = pkg.Corpus(vt, dim)
                few.append(rows[:nn], rowids[:nn])
                few.set_tie_order(tie)
                want_ids, want_d = few.scan_topk(metric, q, k)
                few.close()
            s = pkg.SlabScan(vt, dim, metric, q, k, slab, tie_order=tie)
            for r0 in range(0, nn, 4_097):                                          # hand-overs that straddle slab boundaries
                s.rows(rows[r0:min(nn, r0 + 4_097)], rowids[r0:min(nn, r0 + 4_097)])
            ids, d = s.finish()
            s.close()
            assert ids.tolist() == want_ids.tolist(), (vt, tie, metric, k, slab)
            assert dg.same_float_bits(d.astype(np.float32), want_d.astype(np.float32)), (vt, tie, metric, k, slab)
        # every distance (k = 0): the *_stream functions
        s = pkg.SlabScan(vt, dim, metric, q, 0, 9_000)
        s.rows(rows, rowids)
        s.finish()
        d, ids = s.all()
        s.close()
        assert np.array_equal(ids, rowids) and dg.same_float_bits(d, one.scan_distances(metric, q)), (vt, metric)
    one.close()


def test_slab_scan_implicit_rowids_records_and_errors(pkg):
    dim, n = 24, 30_000
    rows = dg.corpus(dg.U8, n, dim, 820, low_entropy=True)
    q = rows[17].copy()
    one = pkg.Corpus(dg.U8, dim)
    one.append(rows)                                                                # implicit rowids 1 ..
    one.set_tie_order(1)
    want_ids, want_d = one.scan_topk(dg.L2, q, 30)
    s = pkg.SlabScan(dg.U8, dim, dg.L2, q, 30, 4_000, tie_order=1)
    s.rows(rows)
    ids, d = s.finish()
    assert ids.tolist() == want_ids.tolist() and np.array_equal(d, want_d)
    with pytest.raises(pkg.VectorGpuError):
        s.rows(rows[:10])                                                           # rows after finish
    s.close()
    # the persisted record format: [int64 LE rowid | dim bytes]
    rowids = np.arange(n, dtype=np.int64) * 2 + 9
    rec = np.zeros((n, 8 + dim), dtype=np.uint8)
    rec[:, :8] = rowids.astype("<i8").view(np.uint8).reshape(n, 8)
    rec[:, 8:] = rows
    s = pkg.SlabScan(dg.U8, dim, dg.L2, q, 30, 7_777, tie_order=1)
    s.records(rec[:12_345], 12_345)
    s.records(rec[12_345:], n - 12_345)
    ids, d = s.finish()
    s.close()
    assert ids.tolist() == (want_ids * 2 + 7).tolist() and np.array_equal(d, want_d)
    # an empty table
    s = pkg.SlabScan(dg.U8, dim, dg.L2, q, 5, 100)
    ids, d = s.finish()
    s.close()
    assert len(ids) == 0
    free, total = pkg.device_memory(0)
    assert 0 < free <= total
    one.close()
