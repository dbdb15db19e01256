"""tie_order = reference: the reference's slot-history-dependent result among EQUAL distances, rowid for rowid.

The reference keeps k unsorted slots, inserts with a strict '<' into the FIRST slot holding the maximum and exchange-sorts
at the end (sqlite-vector.c:2022-2069, 2102-2106); the oracle restates that (orc_topk_reference, pinned to the reference
build by tests/test_oracle_vs_reference.py).  The product replays it on the host over the rows that can enter at all
(csrc/vg_refslots.h) - CPU tests drive that replay directly, GPU tests the full device path."""
import os

import numpy as np
import pytest

import datagen as dg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg_cpu():
    import __graft_entry__ as g
    g._load_build().build_gpu_library()
    return g.load_package()


def _streams(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 5, 50, 700, 5000, 40000, 300000]))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        d = rng.standard_normal(n).astype(np.float32)
    elif kind == 1:
        d = rng.integers(0, 4, n).astype(np.float32)                      # heavy ties
    elif kind == 2:
        d = rng.integers(0, 200, n).astype(np.float32)
    elif kind == 3:
        d = np.sort(rng.integers(0, 50, n)).astype(np.float32)[::-1].copy()   # descending: every row enters
    elif kind == 4:
        d = np.full(n, 7.0, np.float32)
    else:
        d = rng.integers(0, 30, n).astype(np.float32)
        d[rng.random(n) < 0.2] = np.float32(np.inf)                       # never enter
        d[rng.random(n) < 0.1] = np.float32(np.nan)
        d[rng.random(n) < 0.01] = np.float32(-np.inf)
    k = int(rng.choice([1, 2, 3, 20, 33, 64, 100, 1000]))
    return d, k


@pytest.mark.parametrize("chunk", range(8))
def test_host_replay_equals_the_reference_slot_algorithm(pkg_cpu, orc, chunk):
    """the replay driver (prefix on the host, then only rows below the bound, then - when those overflow the candidate
    buffer - stretches replayed on the host) must end in the reference's slots, ties included"""
    for i in range(40):
        d, k = _streams(9000 + 100 * chunk + i)
        want_ids, want_d = orc.topk_reference(d, None, k)                  # rowids = position + 1
        for cap in (0, 7, 1):                                              # tiny caps force the overflow path
            pos, got_d = pkg_cpu.reference_topk_replay(d, k, cap)
            assert (pos + 1).tolist() == want_ids.tolist(), (chunk, i, cap, k, len(d))
            assert np.array_equal(got_d, want_d, equal_nan=True)


@pytest.mark.parametrize("chunk", range(4))
def test_host_replay_slab_by_slab_equals_the_reference_slot_algorithm(pkg_cpu, orc, chunk):
    """the out-of-core scan's form of the replay (vg_slabscan.hip / vg_ref_replay_more): the stream arrives in slabs, the slot state -
    distances, positions, the index of the current maximum - is carried from one slab to the next; any slab size, the overflow path
    included, must end in the reference's slots"""
    rng = np.random.default_rng(9900 + chunk)
    for i in range(30):
        d, k = _streams(9500 + 100 * chunk + i)
        want_ids, want_d = orc.topk_reference(d, None, k)
        for slab in (1, int(rng.integers(2, 50)), int(rng.integers(50, 5000)), len(d) + 5):
            for cap in (0, 3):
                pos, got_d = pkg_cpu.reference_topk_replay_slabs(d, k, slab, cap)
                assert (pos + 1).tolist() == want_ids.tolist(), (chunk, i, slab, cap, k, len(d))
                assert np.array_equal(got_d, want_d, equal_nan=True)


def test_reference_examples_from_the_survey(pkg_cpu):
    """SURVEY section 7 probes: [5,5,3] k=2 -> rowids {3,2}; [3,5,5,5,1] k=3 -> {5,1,3}"""
    pos, d = pkg_cpu.reference_topk_replay(np.array([5, 5, 3], np.float32), 2)
    assert (pos + 1).tolist() == [3, 2] and d.tolist() == [3.0, 5.0]
    pos, d = pkg_cpu.reference_topk_replay(np.array([3, 5, 5, 5, 1], np.float32), 3)
    assert (pos + 1).tolist() == [5, 1, 3] and d.tolist() == [1.0, 3.0, 5.0]


# ------------------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def pkg():
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


@pytest.mark.gpu
@pytest.mark.parametrize("vt", (dg.U8, dg.I8, dg.F32))
def test_reference_tie_order_on_device_equals_the_reference(pkg, orc, vt, monkeypatch):
    """low-entropy rows (distances tie constantly): rowids AND order must be the reference's for every metric; one corpus
    and 3 logical shards with ragged blocks (their per-query work on the handle's host threads, as on several devices); k below /
    at / above the fused limit; n below and above the prefix."""
    monkeypatch.setenv("VECTORGPU_SHARD_THREADS", "1")
    rng = np.random.default_rng(31 + vt)
    for dim, n in ((16, 500), (48, 70_001), (8, 1_200_000)):
        rows = dg.corpus(vt, n, dim, 600 + dim, low_entropy=True)
        if vt == dg.F32:
            rows = np.rint(rows).astype(np.float32)
        q = rows[int(rng.integers(0, n))].copy()
        ids = (np.arange(n, dtype=np.int64) * 3 + 11)                      # explicit rowids
        c = pkg.Corpus(vt, dim)
        c.append(rows, ids)
        sh = pkg.Shards(vt, dim, [0, 0, 0], block_rows=4099)
        assert sh.threaded
        sh.append(rows, ids)
        c.set_tie_order(pkg.TIE_REFERENCE)
        sh.set_tie_order(pkg.TIE_REFERENCE)
        for metric in dg.ALL_METRICS:
            dist = c.scan_distances(metric, q)                             # the GPU's own floats (int types: bit-exact with the oracle)
            if vt != dg.F32:
                assert dg.same_float_bits(dist, orc.scan_distances(orc.AVX2, metric, vt, q, rows))
            for k in (1, 20, 64, 65, 300):
                want_ids, want_d = orc.topk_reference(dist, ids, k)
                got_ids, got_d = c.scan_topk(metric, q, k)
                assert got_ids.tolist() == want_ids.tolist(), (vt, dim, n, metric, k)
                assert np.array_equal(got_d, want_d)
                s_ids, s_d = sh.scan_topk(metric, q, k)
                assert s_ids.tolist() == want_ids.tolist(), ("shards", vt, dim, n, metric, k)
                assert np.array_equal(s_d, want_d)
        # only k > 64 (no fused list that long) may take the store-mode replay - one corpus and three shards alike, small corpora
        # (the whole corpus is the replay's prefix) and k = 64 (always replayed from what the scan emitted) included
        for st in (c.tie_stats(), sh.tie_stats()):
            assert st["store_mode_replays"] == 2 * len(dg.ALL_METRICS), (vt, dim, n, st)
            assert st["fused_replays"] >= len(dg.ALL_METRICS), (vt, dim, n, st)      # (k = 64 alone replays once per metric)
        # batches in this mode are one replayed scan per query
        qs = rows[rng.integers(0, n, 5)].copy()
        bi, bd, bc = c.scan_topk_batch(dg.L2, qs, 20)
        si, sd, sc = sh.scan_topk_batch(dg.L2, qs, 20)
        for j in range(5):
            w_ids, w_d = orc.topk_reference(c.scan_distances(dg.L2, qs[j]), ids, 20)
            assert bi[j][:bc[j]].tolist() == w_ids.tolist() and si[j][:sc[j]].tolist() == w_ids.tolist()
        c.set_tie_order(pkg.TIE_POSITION)                                  # and back: (distance, position) order
        got_ids, got_d = c.scan_topk(dg.L1, q, 20)
        o_ids, o_d, _ = orc.topk_ordered(c.scan_distances(dg.L1, q), ids, 20)
        assert got_ids.tolist() == o_ids.tolist()
        c.close()
        sh.close()


@pytest.mark.gpu
def test_reference_tie_order_when_candidates_overflow_the_device_buffer(pkg, orc):
    """descending distances: every row beats the bound reached so far, the candidate buffer overflows and the driver falls
    back to replaying stretches on the host - still the reference's answer"""
    n, dim = 400_000, 4
    rows = np.zeros((n, dim), np.float32)
    rows[:, 0] = np.arange(n, 0, -1, dtype=np.float32) // 3                # distance to the zero query falls along the scan, with ties
    q = np.zeros(dim, np.float32)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    c.set_tie_order(pkg.TIE_REFERENCE)
    for k in (5, 64, 200):
        got_ids, got_d = c.scan_topk(dg.L1, q, k)
        want_ids, want_d = orc.topk_reference(c.scan_distances(dg.L1, q), None, k)
        assert got_ids.tolist() == want_ids.tolist() and np.array_equal(got_d, want_d)
    c.close()


@pytest.mark.gpu
def test_fused_reference_order_costs_a_replay_only_when_the_k_plus_1_best_hold_a_tie(pkg, orc):
    """the round-3 form of tie_order = reference: the ordinary top-k scan with one more list slot.  Distinct distances among the
    k + 1 best -> no replay at all (counters), and the answer is the (distance, position) answer = the reference's.  A tie -> the
    fused replay (prefix pass + the candidates the scan emitted), not a second pass over the corpus."""
    n, dim, k = 300_000, 32, 20
    rows = dg.corpus(dg.U8, n, dim, 901)
    q = dg.query(dg.U8, dim, 902)
    c = pkg.Corpus(dg.U8, dim)
    c.append(rows)
    c.set_tie_order(pkg.TIE_REFERENCE)
    dist = c.scan_distances(dg.COSINE, q)
    top = np.sort(dist)[:k + 1]
    assert len(np.unique(top)) == k + 1                                     # (random bytes, cosine: no tie up there)
    before = c.tie_stats()
    ids, d = c.scan_topk(dg.COSINE, q, k)
    after = c.tie_stats()
    want_ids, want_d = orc.topk_reference(dist, None, k)
    assert ids.tolist() == want_ids.tolist() and np.array_equal(d, want_d)
    assert after["scans"] == before["scans"] + 1 and after["with_a_tie_among_the_k_plus_1_best"] == before["with_a_tie_among_the_k_plus_1_best"]
    # plant exact copies of the best rows: ties among the k + 1 best, resolved by the fused replay
    rows2 = rows.copy()
    best = (want_ids[:3] - 1).tolist()
    rows2[[1000, 150_000, 299_999]] = rows[best]
    rows2[[7, 8]] = rows[best[0]]
    c2 = pkg.Corpus(dg.U8, dim)
    c2.append(rows2)
    c2.set_tie_order(pkg.TIE_REFERENCE)
    for metric in (dg.COSINE, dg.L2, dg.DOT, dg.L1):
        dist2 = c2.scan_distances(metric, q)
        for kk in (1, 5, 20, 63):
            ids2, d2 = c2.scan_topk(metric, q, kk)
            w_ids, w_d = orc.topk_reference(dist2, None, kk)
            assert ids2.tolist() == w_ids.tolist() and np.array_equal(d2, w_d), (metric, kk)
    st = c2.tie_stats()
    assert st["fused_replays"] >= 4 and st["store_mode_replays"] == 0, st
    c.close()
    c2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("vt", (dg.F32, dg.F16, dg.U8))
def test_reference_order_through_the_filter_scans(pkg, orc, vt, monkeypatch):
    """the filter kernels (int8 shadow copy for f32 / f16, high nibbles for uint8) emit the replay's candidates themselves and
    their pre-pass doubles as its prefix pass: tie-heavy corpora answered in the reference's order, no store-mode scan"""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    n, dim, k = 400_000, 64, 20
    rng = np.random.default_rng(77 + vt)
    rows = dg.corpus(vt, n, dim, 910 + vt)
    q = rows[12345].copy()
    # duplicates of the query's row and of a near row, spread over the scan: ties at distance 0 and at the next distance
    near = rows[54321].copy()
    rows[[5, 99_999, 250_000, 399_998]] = q
    rows[[6, 100_001, 300_000]] = near
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    c.set_scan_filter(1)
    c.set_tie_order(pkg.TIE_REFERENCE)
    for metric in (dg.L2, dg.COSINE, dg.DOT):
        ids0, _ = c.scan_topk(metric, q, k)                                 # (builds the shadow copy)
        assert c.kernel_name(metric).startswith("scan_filter"), c.kernel_name(metric)
        dist = c.scan_distances(metric, q)
        for kk in (3, 20, 40):
            ids, d = c.scan_topk(metric, q, kk)
            w_ids, w_d = orc.topk_reference(dist, None, kk)
            assert ids.tolist() == w_ids.tolist() and np.array_equal(d, w_d), (vt, metric, kk)
    st = c.tie_stats()
    assert st["fused_replays"] >= 3 and st["store_mode_replays"] == 0, st
    # batches: the matrix-core pass with one more slot, only tie queries are answered again one by one
    qs = np.stack([q, rows[777], near, rows[31]])
    bi, bd, bc = c.scan_topk_batch(dg.L2, qs, k)
    for j in range(len(qs)):
        w_ids, w_d = orc.topk_reference(c.scan_distances(dg.L2, qs[j]), None, k)
        assert bi[j][:bc[j]].tolist() == w_ids.tolist(), (vt, j)
        if vt == dg.U8:
            assert np.array_equal(bd[j][:bc[j]], w_d), (vt, j)
        else:                     # (a query without a tie keeps the matrix-core kernel's distances: another summation order)
            assert np.allclose(bd[j][:bc[j]], w_d, rtol=1e-5, atol=1e-6), (vt, j)
    c.close()
