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


@pytest.mark.parametrize("chunk", range(6))
def test_random_shapes_vs_oracle(pkg, orc, chunk):
    for vt, metric, dim, n, k, low, seed in _cases(1000 + chunk, 25):
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
    for _ in range(20):
        vt = int(rng.choice([dg.F32, dg.U8, dg.I8, dg.F16, dg.BF16]))
        metric = int(rng.choice(dg.ALL_METRICS))
        dim = int(rng.integers(1, 513)) if vt == dg.F32 else (int(rng.integers(1, 2200)) if vt in (dg.U8, dg.I8) else int(rng.integers(1, 1100)))
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
