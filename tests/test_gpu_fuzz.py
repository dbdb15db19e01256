"""Deterministic differential sweep: many random (type, metric, dim, rows, k) shapes, the HIP scan through the C-ABI
against the pinned oracle.  Catches shape-dependent bugs the hand-picked cases miss (launch-shape selection, ragged
tails, list merges with few rows, large k, store mode).  Same bars as test_gpu_scan.py."""
import numpy as np
import pytest

import datagen as dg
from test_gpu_scan import _check_float_distances, pkg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        vt = int(rng.choice(dg.ALL_TYPES))
        metric = int(rng.choice(dg.ALL_METRICS))
        dim = int(rng.choice([rng.integers(1, 40), rng.integers(40, 400), rng.integers(400, 2100), rng.integers(2100, 9000)],
                             p=[0.3, 0.4, 0.25, 0.05]))
        n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 3000), rng.integers(3000, 40000)], p=[0.25, 0.5, 0.25]))
        if dim > 2100:
            n = min(n, 600)
        k = int(rng.choice([1, 2, 7, 20, 63, 64, 65, 130, 1000]))
        low = bool(rng.integers(0, 2)) and vt in (dg.U8, dg.I8)
        out.append((vt, metric, dim, n, k, low, int(rng.integers(0, 1 << 30))))
    return out


def _scale():
    """VG_FUZZ_SCALE=10 runs every sweep of this file ten times as wide (new seeds): the occasional long soak"""
    import os
    return max(1, int(os.environ.get("VG_FUZZ_SCALE", "1")))


@pytest.mark.parametrize("chunk", range(6))
def test_random_shapes_vs_oracle(pkg, orc, chunk):
    for vt, metric, dim, n, k, low, seed in _cases(1000 + chunk, 25 * _scale()):
        rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
        q = dg.query(vt, dim, seed + 1, low_entropy=low)
        want = orc.scan_distances(orc.AVX2, metric, vt, q, rows)
        c = pkg.Corpus(vt, dim)
        half = n // 2
        c.append(rows[:half]) if half else None
        c.append(rows[half:])
        got = c.scan_distances(metric, q)
        tag = (vt, metric, dim, n, k, low, seed)
        if vt in (dg.U8, dg.I8):
            assert dg.same_float_bits(got, want), tag
        else:
            _check_float_distances(got, want, vt, metric, q, rows)
        ids, dist = c.scan_topk(metric, q, k)
        oids, odist, _ = orc.topk_ordered(got, None, k)          # selection is exact on the GPU's own floats
        assert ids.tolist() == oids.tolist() and np.array_equal(dist, odist), tag
        c.close()


@pytest.mark.parametrize("chunk", range(4))
def test_random_batches_vs_single_scans(pkg, chunk):
    """batched scans (matrix-core kernels and their fallbacks) against the single-query kernel on random shapes:
    quantized types bit for bit, f16 / bf16 within 1e-6, f32 within 1e-5 with the same rowids unless two candidates differ by less than that."""
    rng = np.random.default_rng(2000 + chunk)
    for _ in range(20 * _scale()):
        vt = int(rng.choice([dg.F32, dg.U8, dg.I8, dg.F16, dg.BF16]))
        metric = int(rng.choice(dg.ALL_METRICS))
        dim = int(rng.integers(1, 513)) if vt == dg.F32 else (int(rng.integers(1, 2200)) if vt in (dg.U8, dg.I8) else int(rng.integers(1, 1100)))
        if vt in (dg.F32, dg.F16, dg.BF16) and rng.integers(0, 3) == 0:      # long rows: 513 .. 1024 (f32 through the bf16 filter), 1025 .. 3072
            dim = int(rng.integers(513, 3300))                               # (the K-split kernel), beyond (the scans)
        n = int(rng.choice([rng.integers(1, 100), rng.integers(100, 5000), rng.integers(5000, 30000)]))
        nq = int(rng.choice([1, 3, 33, 129, 260]))
        k = int(rng.choice([1, 5, 20, 32, 40]))
        low = bool(rng.integers(0, 2)) and vt in (dg.U8, dg.I8)
        seed = int(rng.integers(0, 1 << 30))
        rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
        qs = dg.corpus(vt, nq, dim, seed + 1, low_entropy=low)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        tag = (vt, metric, dim, n, nq, k, low, seed)
        for i in sorted(set([0, nq // 2, nq - 1])):
            one_ids, one_dist = c.scan_topk(metric, qs[i], k)
            assert cnt[i] == len(one_ids), tag
            if vt in (dg.F16, dg.BF16):         # matrix-core filter + the single scan's f64 arithmetic (another summation order)
                assert np.allclose(dist[i][:cnt[i]], one_dist, rtol=1e-6, atol=1e-7), tag
                assert len(set(ids[i][:cnt[i]].tolist()) ^ set(one_ids.tolist())) <= 2, tag
            elif vt != dg.F32:
                assert ids[i][:cnt[i]].tolist() == one_ids.tolist() and np.array_equal(dist[i][:cnt[i]], one_dist), tag
            else:
                scale = float(np.abs(qs[i]).sum()) * 4.0 if metric == dg.DOT else (1.0 if metric == dg.COSINE else 0.0)
                assert np.all(np.abs(dist[i][:cnt[i]] - one_dist) <= 1e-5 * (np.abs(one_dist) + scale) + 1e-6), tag
                assert len(set(ids[i][:cnt[i]].tolist()) ^ set(one_ids.tolist())) <= 2, tag
        c.close()


@pytest.mark.parametrize("chunk", range(2))
def test_random_int8_filter_batches_vs_single_scans(pkg, chunk, monkeypatch):
    """the int8 matrix-core filter (vg_batch_q8.hip; forced: its default starts at 2^20 rows and 257 queries) on random shapes - every
    row-length class (k-steps 4 / 8 / 12 / 16, K in two and three ring parts up to 1536 elements), the three float types, the four metrics
    it serves, ragged last tiles, batches that are nearly all padding: against the single-query kernel like the batches above."""
    monkeypatch.setenv("VG_BATCH_Q8", "1")
    rng = np.random.default_rng(2500 + chunk)
    for _ in range(6 * _scale()):
        vt = int(rng.choice([dg.F32, dg.F32, dg.F16, dg.BF16]))
        metric = int(rng.choice([dg.L2, dg.SQUARED_L2, dg.DOT, dg.COSINE]))
        dim = int(rng.choice([rng.integers(1, 129), rng.integers(129, 513), rng.integers(513, 1025), rng.integers(1025, 1537)]))
        n = int(rng.integers(65_537, 90_000)) if dim > 512 else int(rng.integers(65_537, 200_000))
        nq = int(rng.choice([7, 33, 129, 260, 300]))
        k = int(rng.choice([1, 5, 20, 32]))
        seed = int(rng.integers(0, 1 << 30))
        rows = dg.corpus(vt, n, dim, seed)
        qs = dg.corpus(vt, nq, dim, seed + 1)
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
        tag = (vt, metric, dim, n, nq, k, seed, c.last_batch_path(), c.batch_q8_status())
        assert c.last_batch_path() == 7, tag
        for i in sorted(set([0, nq // 2, nq - 1])):
            one_ids, one_dist = c.scan_topk(metric, qs[i], k)
            assert cnt[i] == len(one_ids), tag
            if vt in (dg.F16, dg.BF16):         # the pairs that pass carry the single scan's f64 arithmetic: the same floats
                assert ids[i][:cnt[i]].tolist() == one_ids.tolist() and dg.same_float_bits(dist[i][:cnt[i]], one_dist), tag
            else:                               # (f32: the single scan's chain, another launch shape's summation order)
                scale = float(np.abs(qs[i]).sum()) * 4.0 if metric == dg.DOT else (1.0 if metric == dg.COSINE else 0.0)
                assert np.all(np.abs(dist[i][:cnt[i]] - one_dist) <= 1e-5 * (np.abs(one_dist) + scale) + 1e-6), tag
                assert len(set(ids[i][:cnt[i]].tolist()) ^ set(one_ids.tolist())) <= 2, tag
        c.close()


def _tie_cases(seed, count):
    """(type, metric, dim, rows, k, value levels, planted duplicates, filter scans forced, shards, seed): corpora whose distances tie
    at every density from "never" to "constantly", on both sides of the sizes where the reference-order scan changes its form
    (store-mode replay below 2^17 rows, fused replay above; k + 1 > 64: store mode again)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        vt = int(rng.choice([dg.U8, dg.I8, dg.F32, dg.F16]))
        metric = int(rng.choice(dg.ALL_METRICS))
        dim = int(rng.choice([rng.integers(1, 12), rng.integers(12, 100), rng.integers(100, 400)], p=[0.4, 0.4, 0.2]))
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20000), rng.integers(131072, 400000)], p=[0.3, 0.35, 0.35]))
        k = int(rng.choice([1, 2, 5, 20, 40, 63, 64, 100]))
        levels = int(rng.choice([2, 4, 16, 256]))
        dups = int(rng.choice([0, 0, 3, 40]))
        filt = bool(rng.integers(0, 2)) and n >= 131072 and metric != dg.L1
        shards = int(rng.choice([1, 1, 3]))
        out.append((vt, metric, dim, n, k, levels, dups, filt, shards, int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("chunk", range(4))
def test_reference_order_fuzz(pkg, orc, chunk, monkeypatch):
    """tie_order = reference on random tie-heavy shapes, every path it can take (plain / filter kernels, one corpus / shards, the
    fused replay / the store-mode replay / no replay at all): rowids AND order must be the reference's slot algorithm's
    (orc.topk_reference, pinned to the reference extension) over the GPU's own distances.  VG_FUZZ_CASES scales the sweep."""
    import os
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    per_chunk = max(1, int(os.environ.get("VG_FUZZ_CASES", str(48 * _scale()))) // 4)
    for vt, metric, dim, n, k, levels, dups, filt, shards, seed in _tie_cases(3000 + chunk, per_chunk):
        rng = np.random.default_rng(seed)
        if vt == dg.U8:
            rows = rng.integers(0, levels, (n, dim)).astype(np.uint8)
        elif vt == dg.I8:
            rows = rng.integers(-(levels // 2), levels // 2 + 1, (n, dim)).astype(np.int8)
        else:
            rows = dg.to_storage(vt, (rng.integers(0, levels, (n, dim)) - levels // 2).astype(np.float32) / 2.0)
        q = rows[int(rng.integers(0, n))].copy()
        for _ in range(dups):                                               # exact copies: ties at distance 0 and elsewhere
            rows[int(rng.integers(0, n))] = rows[int(rng.integers(0, n))]
        tag = (vt, metric, dim, n, k, levels, dups, filt, shards, seed)
        if shards == 1:
            c = pkg.Corpus(vt, dim)
            c.append(rows)
        else:
            monkeypatch.setenv("VECTORGPU_SHARD_THREADS", str(seed & 1))    # the per-shard work on host threads, or issued by this one
            c = pkg.Shards(vt, dim, [0] * shards, block_rows=int(rng.choice([257, 4099, 65536])))
            c.append(rows)
        c.set_scan_filter(1 if filt else 0)
        c.set_tie_order(pkg.TIE_REFERENCE)
        dist = c.scan_distances(metric, q)
        want_ids, want_d = orc.topk_reference(dist, None, k)
        ids, d = c.scan_topk(metric, q, k)
        assert ids.tolist() == want_ids.tolist() and np.array_equal(d, want_d), tag
        c.set_tie_order(pkg.TIE_POSITION)
        oids, od, _ = orc.topk_ordered(dist, None, k)
        ids, d = c.scan_topk(metric, q, k)
        assert ids.tolist() == oids.tolist() and np.array_equal(d, od), tag
        c.close()
