"""Adversarial inputs for the bf16 shadow-copy filter (vg_scan_filter.h, vg_batch_h.hip with an f32 corpus).

The filter's "lower bound" of a row's distance rests on |s~ - s| <= c |q||x| for the dot product s~ of the bf16-ROUNDED
query and row.  Round 1 shipped c = 2^-8 (1 + 2^-8) - one input's rounding; both are rounded, the true constant is
2^-7 (1 + 2^-9).  Gaussian test data never noticed (the rounding errors cancel like sqrt(D)); rows whose elements all
round the SAME way do: an exact duplicate of the query (true distance 0) was dropped.  Every case here

  * is checked by a plain-numpy restatement of the kernel's bound formula to be a counter-example for the old constant
    (so the file documents what was wrong) and to be admitted by the current one,
  * must return the plain f32 scan's rowids and distance bits (filter off) AND match the oracle.

The reference behaviour matched is distance-avx2.c:67-100,128-151 + the strict '<' slot insertion sqlite-vector.c:2102-2106.
"""
import numpy as np
import pytest

import datagen as dg
from test_gpu_scan import pkg, _check_float_distances  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

OLD_C = 2.0 ** -8 * (1 + 2.0 ** -8)       # round 1 (unsound)
NEW_C = 2.0 ** -7 + 2.0 ** -16            # (1 + u)^2 - 1, u = 2^-8
DOWN = np.float32(1 + 2.0 ** -8 - 2.0 ** -20)    # just below a bf16 midpoint: rounds DOWN to 1.0
UP = np.float32(1 + 2.0 ** -8 + 2.0 ** -20)      # just above it: rounds UP to 1 + 2^-7


@pytest.fixture(autouse=True, params=("bf16", "int8"))
def shadow(request, monkeypatch):
    """every case of this file runs through BOTH shadow copies of an f32 corpus: bf16 (half the bytes, rounding-error bound) and
    int8 (a quarter of the bytes, residual-norm bound - vg_scan_filter.h, Q8).  Whatever the data, the answers are the plain scan's."""
    monkeypatch.setenv("VG_SCAN_FILTER_SHADOW", request.param)
    return request.param


@pytest.fixture(autouse=True)
def _no_selectivity_guard(request, monkeypatch):
    """these tests compare the FILTER kernel with the plain one: the selectivity guard (which sends an unselective corpus back
    to the plain kernel) must not quietly make such a comparison vacuous; its own test switches it back on"""
    if "guard" not in request.node.name:
        monkeypatch.setenv("VG_SCAN_FILTER_NO_GUARD", "1")


def bf16_round(x):
    return dg.bf16_bits_to_f32(dg.f32_to_bf16_bits(x))


def kernel_lower_bound(q, x, c_base, metric):
    """vg_scan_filter.h's bound, restated in f64 (its own f32 rounding is covered by the (D + 64) 2^-21 / 2^-22 slacks)"""
    D = q.size
    cerr = c_base + (D + 64) * 2.0 ** -21
    rel = (D + 64) * 2.0 ** -22
    st = float((bf16_round(q).astype(np.float64) * bf16_round(x).astype(np.float64)).sum())
    qq = float((q.astype(np.float64) ** 2).sum())
    nn = float((x.astype(np.float64) ** 2).sum())
    E = cerr * np.sqrt(qq) * np.sqrt(nn)
    if metric == dg.DOT:
        return -(st + E) - rel * np.sqrt(qq * nn)
    return qq + nn - 2.0 * (st + E) - rel * (qq + nn)            # squared distance


def true_distance(q, x, metric):
    q64, x64 = q.astype(np.float64), x.astype(np.float64)
    if metric == dg.DOT:
        return -float((q64 * x64).sum())
    return float(((q64 - x64) ** 2).sum())                        # squared


def adversarial_case(dim, metric, n_comp):
    """(query, target row, competitor rows): the target is the true best row, the competitors are bf16-EXACT rows (their
    bound is tight) slightly worse than the target - with the old constant the target's bound lands above them."""
    if metric == dg.DOT:
        q = np.full(dim, -UP, np.float32)                         # q~ x~ = -(1 + 2^-7)^2: more negative than q x
        target = np.full(dim, UP, np.float32)
        comps = []
        for j in range(n_comp):
            c = np.ones(dim, np.float32)
            c[:dim // 2 + 1 + (j % 3)] = np.float32(1 + 2.0 ** -7)
            comps.append(np.roll(c, j))
    else:
        q = np.full(dim, DOWN, np.float32)
        target = q.copy()                                         # distance exactly 0
        comps = []
        for j in range(n_comp):
            c = np.ones(dim, np.float32)
            c[j % dim] = np.float32(1 + 2.0 ** -7) if (j % 4 == 3) else np.float32(1.0)
            comps.append(c)
    return q, target, np.stack(comps)


def check_case_is_adversarial(q, target, comps, metric, k):
    d_t = true_distance(q, target, metric)
    d_c = np.sort([true_distance(q, c, metric) for c in comps])
    assert d_t < d_c[0], "the target must be the true best row"
    thr = d_c[k - 1]                                              # the list's k-th best once k competitors are in
    assert kernel_lower_bound(q, target, OLD_C, metric) > thr, "not a counter-example for the round-1 constant"
    assert kernel_lower_bound(q, target, NEW_C, metric) <= d_t, "the corrected bound must stay below the true distance"


def filler(n, dim, seed):
    """far-away rows (never candidates) so that the corpus has every launch shape's ragged tail"""
    return (dg.corpus(dg.F32, n, dim, seed) * np.float32(3.0) + np.float32(8.0)).astype(np.float32)


@pytest.mark.parametrize("dim", (33, 384))
@pytest.mark.parametrize("metric", (dg.L2, dg.SQUARED_L2, dg.DOT))
def test_filter_keeps_the_exact_duplicate_when_every_element_rounds_the_same_way(pkg, orc, dim, metric, shadow, monkeypatch):
    """no pre-pass here (n < 2^20): every wavefront tightens its OWN list, so the competitors are dense - every 4th row -
    and each of the ~4096 wavefronts has met far more than k of them when the target arrives near the end of the scan"""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    k = 20
    q, target, comps = adversarial_case(dim, metric, 3 * k)
    check_case_is_adversarial(q, target, comps, metric, k)
    n = 600_011
    rows = filler(n, dim, 4100 + dim)
    if metric == dg.DOT:
        rows = np.abs(rows)                                       # q < 0: large positive rows have large positive (far) distances
    pos_comp = np.arange(0, n - 1000, 4)
    rows[pos_comp] = comps[np.arange(len(pos_comp)) % len(comps)]
    pos_t = n - 77                                                # the target late: the lists are warm when it arrives
    rows[pos_t] = target
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    want = orc.scan_distances(orc.AVX2, metric, dg.F32, q, rows)
    oids, _, _ = orc.topk_ordered(want, None, 1)
    assert oids[0] == pos_t + 1
    for kk in (1, k, 64):
        c.set_scan_filter(1)
        assert c.kernel_name(metric).startswith("scan_filter_f32") and ("_q8_" in c.kernel_name(metric)) == (shadow == "int8")
        ids1, d1 = c.scan_topk(metric, q, kk)
        c.set_scan_filter(0)
        assert not c.kernel_name(metric).startswith("scan_filter")
        ids0, d0 = c.scan_topk(metric, q, kk)
        assert ids0[0] == pos_t + 1
        assert ids1.tolist() == ids0.tolist(), (dim, metric, kk)
        assert dg.same_float_bits(d1, d0), (dim, metric, kk)
        if metric != dg.DOT:
            assert d1[0] == 0.0
        _check_float_distances(d1.astype(np.float32), want[ids1 - 1], dg.F32, metric, q, rows[ids1 - 1])
    c.set_scan_filter(-1)
    c.close()


@pytest.mark.parametrize("dim,metric", ((33, dg.L2), (64, dg.DOT), (96, dg.SQUARED_L2)))
def test_filter_prepass_branch_on_clustered_unit_norm_rows(pkg, orc, dim, metric, monkeypatch):
    """n >= 2^20 rows: the filter scan starts from the threshold of its plain-f32 pre-pass (vg_api.hip).  Unit-norm rows
    clustered around a few centres (|q||x| slack large against the distance gaps), the query equal to a late row, plus the
    same-way-rounding block from above scaled to unit norm."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    n = (1 << 20) + 4099
    rng = np.random.default_rng(5200 + dim)
    centres = rng.standard_normal((7, dim)).astype(np.float32)
    rows = centres[rng.integers(0, 7, n)] + np.float32(0.05) * rng.standard_normal((n, dim), dtype=np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
    k = 20
    q, target, comps = adversarial_case(dim, metric, 2 * k)
    check_case_is_adversarial(q, target, comps, metric, k)
    # far from the clusters?  not needed: the assertions below compare with the plain scan, whatever the neighbours are
    rows[5000:5000 + len(comps)] = comps                          # inside the pre-pass prefix (first max(65536, n/64) rows)
    pos_t = n - 1234
    rows[pos_t] = target
    twin = n - 99
    queries = [q, rows[twin].copy(), rows[70000].copy(), centres[3] / np.linalg.norm(centres[3])]
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    for qi, qq in enumerate(queries):
        qq = np.ascontiguousarray(qq, dtype=np.float32)
        for kk in (1, k):
            c.set_scan_filter(1)
            ids1, d1 = c.scan_topk(metric, qq, kk)
            c.set_scan_filter(0)
            ids0, d0 = c.scan_topk(metric, qq, kk)
            assert ids1.tolist() == ids0.tolist(), (dim, metric, qi, kk)
            assert dg.same_float_bits(d1, d0), (dim, metric, qi, kk)
        if qi == 0 and metric != dg.DOT:                          # (dot: unit-norm rows with a negative element sum rank ahead)
            assert ids1[0] == pos_t + 1 and d1[0] == 0.0
        want = orc.scan_distances(orc.AVX2, metric, dg.F32, qq, rows)
        _check_float_distances(d1.astype(np.float32), want[ids1 - 1], dg.F32, metric, qq, rows[ids1 - 1])
    c.set_scan_filter(1)
    c.filter_exact_evals()                                        # (reads and resets)
    c.scan_topk(metric, queries[2], k)
    evals = c.filter_exact_evals()                                # instrumentation: the counter moves and is bounded by N
    assert 0 < evals <= n
    c.close()


@pytest.mark.parametrize("dim,force", ((768, "0"), (384, "1")))
@pytest.mark.parametrize("metric", (dg.L2, dg.DOT))
def test_batch_bf16_filter_keeps_the_exact_duplicate(pkg, orc, dim, force, metric, monkeypatch):
    """the same rows through vg_scan_topk_batch: f32 rows of 513 .. 1024 floats always go through the bf16-filter kernel,
    shorter rows with VG_F32_FILTER=1 (vg_batch_h.hip, type_code 2 shares the constant)."""
    monkeypatch.setenv("VG_F32_FILTER", force)
    k = 20
    q, target, comps = adversarial_case(dim, metric, 3 * k)
    check_case_is_adversarial(q, target, comps, metric, k)
    n = 9_001
    rows = filler(n, dim, 4300 + dim)
    if metric == dg.DOT:
        rows = np.abs(rows)
    rows[40:40 + len(comps)] = comps
    pos_t = n - 500
    rows[pos_t] = target
    qs = dg.corpus(dg.F32, 37, dim, 4400 + dim)
    qs[0] = q
    qs[20] = q
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    monkeypatch.setenv("VG_BATCH_MFMA", "1")
    ids, dist, cnt = c.scan_topk_batch(metric, qs, k)
    c.set_scan_filter(0)
    for i in (0, 20):
        assert ids[i][0] == pos_t + 1, (dim, metric, i, ids[i][:4])
        ids0, d0 = c.scan_topk(metric, qs[i], k)
        assert ids0[0] == pos_t + 1
        if metric != dg.DOT:
            assert dist[i][0] == 0.0
        assert cnt[i] == len(d0) and np.allclose(dist[i][:cnt[i]], d0, rtol=1e-5, atol=1e-6)   # (the competitors tie: compare distances)
    c.close()


def _coherent_fuzz_case(rng, dim, n):
    """rows and a query whose bf16 rounding errors all push the dot product the same way: every element is a signed power
    of two times DOWN (rounds toward zero) with the query's sign pattern, so every product shrinks under rounding"""
    sign = rng.choice(np.array([-1.0, 1.0], np.float32), dim)
    e = rng.integers(-2, 3, dim)
    u = (sign * np.exp2(e)).astype(np.float32)                    # bf16-exact direction
    q = (u * DOWN).astype(np.float32)
    rows = dg.corpus(dg.F32, n, dim, int(rng.integers(1 << 30))) * np.float32(0.5)      # zero-mean filler: far for L2, ~orthogonal for dot
    m = n // 3
    base = np.where(rng.random((m, 1)) < 0.5, np.float32(1.0), DOWN).astype(np.float32)      # bf16-exact rows and rounders
    near = (u[None, :] * base).astype(np.float32)
    nflip = rng.integers(0, 6, m)
    for j in range(m):                                            # a few coordinates moved by one bf16 step
        idx = rng.integers(0, dim, nflip[j])
        near[j, idx] = u[idx] * rng.choice(np.array([1.0, 1 + 2.0 ** -7, 1 - 2.0 ** -8, DOWN, UP], np.float32), nflip[j])
    pos = rng.permutation(n)[:m]
    rows[pos] = near
    rows[int(rng.integers(0, n))] = q
    return q, rows


@pytest.mark.parametrize("chunk", range(4))
def test_filter_fuzz_sign_coherent_rounding(pkg, chunk, monkeypatch):
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    rng = np.random.default_rng(7700 + chunk)
    for _ in range(6):
        dim = int(rng.choice([rng.integers(2, 40), rng.integers(40, 400), rng.integers(400, 1100)]))
        n = int(rng.integers(600, 30000))
        q, rows = _coherent_fuzz_case(rng, dim, n)
        c = pkg.Corpus(pkg.F32, dim)
        c.append(rows)
        for metric in (dg.L2, dg.SQUARED_L2, dg.DOT):
            qq = q
            for k in (1, 20, 64):
                c.set_scan_filter(1)
                ids1, d1 = c.scan_topk(metric, qq, k)
                c.set_scan_filter(0)
                ids0, d0 = c.scan_topk(metric, qq, k)
                assert ids1.tolist() == ids0.tolist(), (dim, n, metric, k)
                assert dg.same_float_bits(d1, d0), (dim, n, metric, k)
        c.close()


def test_selectivity_guard_on_a_corpus_of_identical_rows(pkg, orc, monkeypatch):
    """rows that are all alike cannot be separated by any bound: every row is a candidate and the exact evaluations would cost
    more than the plain scan.  The launch watches the kernels' evaluation counter and sends such a corpus back to the plain
    kernel (for 256 scans, then it tries again); the answers are the plain scan's throughout."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    n, dim = (1 << 20) + 3, 24
    base = dg.query(dg.F32, dim, 4242)
    rows = np.tile(base, (n, 1))
    rows[n // 2] = base * np.float32(1.5)                              # one row that differs, so that not all distances tie
    q = (base * np.float32(1.01)).astype(np.float32)
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    c.set_scan_filter(0)
    ids0, d0 = c.scan_topk(dg.L2, q, 20)
    c.set_scan_filter(1)
    c.filter_exact_evals()
    seen = []
    for i in range(6):
        ids1, d1 = c.scan_topk(dg.L2, q, 20)
        assert ids1.tolist() == ids0.tolist() and dg.same_float_bits(d1, d0)
        seen.append(c.filter_exact_evals())
    assert seen[0] > n // 2 and seen[1] > n // 2                      # unselective: (nearly) every row evaluated exactly ...
    assert seen[3] == 0 and seen[4] == 0 and seen[5] == 0              # ... so the following scans take the plain kernel
    c.close()


# ---------------------------------------------------------------------------------------------- the int8 shadow copy
def q8_split(v):
    """the kernel's split of a vector: v = s * vi + e, s = max|v| / 127 (vg_f32_to_q8_kernel / the query's part of the filter kernel)"""
    v64 = v.astype(np.float64)
    s = float(np.float32(np.abs(v).max()) / np.float32(127.0))
    vi = np.clip(np.rint(v64 / s), -127, 127)
    return s, vi, v64 - s * vi


def q8_lower_bound(q, x, metric):
    """vg_scan_filter.h's int8 bound restated in f64: |q.x - sq sx (qi.xi)| <= sq |qi| |ex| + |eq| |x|"""
    sq, qi, eq = q8_split(q)
    sx, xi, ex = q8_split(x)
    st = sq * sx * float((qi * xi).sum())
    E = sq * np.linalg.norm(qi) * np.linalg.norm(ex) + np.linalg.norm(eq) * np.linalg.norm(x.astype(np.float64))
    qq, nn = float((q.astype(np.float64) ** 2).sum()), float((x.astype(np.float64) ** 2).sum())
    if metric == dg.DOT:
        return -(st + E)
    return qq + nn - 2.0 * (st + E)


def q8_adversarial_case(dim, n_comp, seed):
    """every element of the target sits just below a quantization MIDPOINT (s (m + 0.5)): all residuals are +s/2, coherent -
    the worst case for an error estimate that assumed they cancel; query == target.  Competitors are grid-exact rows
    (residual 0: their bound is tight) a little worse than the target."""
    rng = np.random.default_rng(seed)
    s = 1.0 / 127.0
    m = rng.integers(10, 100, dim).astype(np.float64)
    target = (s * (m + 0.499)).astype(np.float32)
    target[0] = np.float32(1.0)                                    # sets the scale: max = 127 s
    m[0] = 127.0
    comps = []
    for j in range(n_comp):
        c = (s * m).astype(np.float32)
        c[1 + (j % (dim - 1))] += np.float32(s * (j % 3))
        comps.append(c)
    return target.copy(), target, np.stack(comps)


@pytest.mark.parametrize("dim", (17, 384, 1000))
@pytest.mark.parametrize("metric", (dg.L2, dg.SQUARED_L2, dg.COSINE))
def test_int8_shadow_keeps_the_exact_duplicate_with_coherent_residuals(pkg, orc, dim, metric, shadow, monkeypatch):
    """the int8 counterpart of the rounding cases above: residuals of the same sign in every element of the query and of the
    true best row.  The restated bound must admit the target (<= its true distance 0) although the ESTIMATE s~ alone would put
    it behind the grid-exact competitors; the scan must return the plain scan's rows and bits."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    k = 20
    q, target, comps = q8_adversarial_case(dim, 3 * k, 6100 + dim)
    sq, qi, eq = q8_split(q)
    sx, xi, ex = q8_split(target)
    st = sq * sx * float((qi * xi).sum())
    naive = float((q.astype(np.float64) ** 2).sum() + (target.astype(np.float64) ** 2).sum() - 2.0 * st)   # estimate without the error term
    d_c = np.sort([true_distance(q, c, dg.L2) for c in comps])
    assert naive > d_c[k - 1] > 0.0, "the estimate alone must rank the target behind k competitors"
    assert q8_lower_bound(q, target, dg.L2) <= 0.0, "the bound must admit the exact duplicate"
    n = 200_003
    rows = filler(n, dim, 6200 + dim)
    pos_comp = np.arange(0, n - 1000, 4)
    rows[pos_comp] = comps[np.arange(len(pos_comp)) % len(comps)]
    pos_t = n - 77
    rows[pos_t] = target
    c = pkg.Corpus(pkg.F32, dim)
    c.append(rows)
    want = orc.scan_distances(orc.AVX2, metric, dg.F32, q, rows)
    for kk in (1, k, 64):
        c.set_scan_filter(1)
        ids1, d1 = c.scan_topk(metric, q, kk)
        c.set_scan_filter(0)
        ids0, d0 = c.scan_topk(metric, q, kk)
        assert ids0[0] == pos_t + 1 and d0[0] == 0.0
        assert ids1.tolist() == ids0.tolist(), (dim, metric, kk)
        assert dg.same_float_bits(d1, d0), (dim, metric, kk)
        _check_float_distances(d1.astype(np.float32), want[ids1 - 1], dg.F32, metric, q, rows[ids1 - 1])
    c.close()


@pytest.mark.parametrize("vt", (dg.F32, dg.F16, dg.BF16))
@pytest.mark.parametrize("chunk", range(2))
def test_int8_shadow_fuzz_outliers_and_scales(pkg, chunk, vt, shadow, monkeypatch):
    """rows whose scale is set by ONE outlier (everything else quantizes to a handful of levels - large residuals), rows of very
    different magnitudes, near-duplicates of the query, zero / Inf / NaN rows, f32 / f16 / bf16 corpora: the filter's answers are
    the plain scan's"""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    rng = np.random.default_rng(8800 + chunk + 10 * vt)
    for _ in range(4):
        dim = int(rng.choice([rng.integers(2, 40), rng.integers(40, 400), rng.integers(400, 1100)]))
        n = int(rng.integers(600, 30000))
        x = rng.standard_normal((n, dim)).astype(np.float32)
        span = 6 if vt == dg.F16 else 20
        x *= np.exp2(rng.integers(-span, span, (n, 1))).astype(np.float32)           # magnitudes over many decades
        out = rng.random(n) < 0.3
        x[out, rng.integers(0, dim)] *= np.float32(300.0)                           # one outlier sets the row's scale
        q = x[int(rng.integers(0, n))].copy()
        dup = rng.permutation(n)[:50]
        x[dup] = q * (1 + rng.standard_normal((50, 1)).astype(np.float32) * np.float32(1e-3))
        x[int(rng.integers(0, n))] = 0.0
        x[int(rng.integers(0, n)), 0] = np.float32(np.inf)
        x[int(rng.integers(0, n)), dim - 1] = np.float32(np.nan)
        rows = dg.to_storage(vt, x)
        qs = dg.to_storage(vt, q[None, :])[0]
        c = pkg.Corpus(vt, dim)
        c.append(rows)
        for metric in (dg.L2, dg.DOT, dg.COSINE):
            for k in (1, 20, 64):
                c.set_scan_filter(1)
                ids1, d1 = c.scan_topk(metric, qs, k)
                c.set_scan_filter(0)
                ids0, d0 = c.scan_topk(metric, qs, k)
                assert ids1.tolist() == ids0.tolist(), (vt, dim, n, metric, k)
                assert dg.same_float_bits(d1, d0), (vt, dim, n, metric, k)
        c.close()


# ---------------------------------------------------------------------------------------------- the nibble filter (uint8 / int8)
def n4_bounds(q, x):
    """vg_scan_filter_n4.h's interval for q.x, restated exactly (integers / f64): A + mq Ls +- ||q'|| ||l'||"""
    q64, x64 = q.astype(np.float64), x.astype(np.float64)
    h = np.floor(x64 / 16.0)
    lo = x64 - 16.0 * h
    A = float((q64 * 16.0 * h).sum())
    mq = q64.mean()
    est = A + mq * lo.sum()
    cs = np.linalg.norm(q64 - mq) * np.linalg.norm(lo - 7.5)
    return est - cs, est + cs


@pytest.mark.parametrize("vt", (dg.U8, dg.I8))
def test_nibble_bound_restated_holds_for_structured_bytes(vt):
    """(CPU arithmetic only) the interval contains q.x for random, constant, extreme and nibble-structured vectors"""
    rng = np.random.default_rng(31 + vt)
    lo_v, hi_v = (0, 255) if vt == dg.U8 else (-128, 127)
    for dim in (1, 7, 32, 100, 768):
        cases = [rng.integers(lo_v, hi_v + 1, dim) for _ in range(40)]
        cases += [np.full(dim, lo_v), np.full(dim, hi_v), np.full(dim, 15), np.full(dim, 16), np.arange(dim) % 16,
                  (np.arange(dim) % 2) * 15 + 16 * rng.integers(0, 4, dim), rng.integers(0, 2, dim) * 15]
        cases = [np.clip(c, lo_v, hi_v).astype(np.int64) for c in cases]
        for q in cases:
            for x in cases:
                lb, ub = n4_bounds(q, x)
                true = float((q * x).sum())
                assert lb - 1e-6 * (1 + abs(true)) <= true <= ub + 1e-6 * (1 + abs(true)), (vt, dim)


@pytest.mark.parametrize("vt", (dg.U8, dg.I8))
@pytest.mark.parametrize("dim", (3, 33, 384, 768, 1536))
def test_nibble_filter_scan_is_bit_identical_to_the_plain_scan(pkg, orc, vt, dim, monkeypatch):
    """uint8 / int8 top-k scans through the high-nibble shadow copy: rowids and distance bits equal the plain scan's and the
    oracle's - random bytes, low-entropy bytes (many exact ties), rows whose LOW nibbles are all 15 or all 0 (the part the
    shadow copy drops, coherent with a query of the same structure), constant rows, the extremes, zero rows and a zero query,
    exact duplicates of the query early and late in the scan, rows appended after the first scan."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    monkeypatch.setenv("VG_SCAN_FILTER_NO_GUARD", "1")
    n = 70_003
    rng = np.random.default_rng(6600 + dim + vt)
    rows = dg.corpus(vt, n, dim, 6700 + dim)
    rows[1000:3000] = dg.corpus(vt, 2000, dim, 6701, low_entropy=True)
    lo_v, hi_v = (0, 255) if vt == dg.U8 else (-128, 127)
    npd = rows.dtype
    base = rng.integers(0, 8, (400, dim)) * 16 + (lo_v if vt == dg.I8 else 0)
    rows[5000:5200] = (base[:200] + 15).astype(np.int64).clip(lo_v, hi_v).astype(npd)      # low nibbles all 15
    rows[5200:5400] = base[200:].astype(np.int64).clip(lo_v, hi_v).astype(npd)            # low nibbles all 0
    rows[6000] = lo_v; rows[6001] = hi_v; rows[6002] = 0; rows[6003] = 15; rows[6004] = 16
    q_struct = (base[7] + 15).clip(lo_v, hi_v).astype(npd)
    rows[100] = q_struct; rows[n - 50] = q_struct
    queries = [dg.query(vt, dim, 6800 + i) for i in range(3)] + [q_struct, np.zeros(dim, npd), np.full(dim, hi_v, npd),
                                                               np.full(dim, lo_v, npd), dg.query(vt, dim, 6810, low_entropy=True)]
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    tag = dg.TYPE_NAMES[vt]
    for metric in (dg.L2, dg.SQUARED_L2, dg.DOT, dg.COSINE):
        c.set_scan_filter(1)
        assert c.kernel_name(metric).startswith("scan_filter_%s" % tag) and "_n4_" in c.kernel_name(metric), c.kernel_name(metric)
        for qi, q in enumerate(queries):
            want = orc.scan_distances(orc.AVX2, metric, vt, q, rows)
            for k in (1, 20, 64):
                c.set_scan_filter(1)
                ids1, d1 = c.scan_topk(metric, q, k)
                c.set_scan_filter(0)
                ids0, d0 = c.scan_topk(metric, q, k)
                assert ids1.tolist() == ids0.tolist(), (vt, dim, metric, qi, k)
                assert dg.same_float_bits(d1, d0), (vt, dim, metric, qi, k)
            assert dg.same_float_bits(d1.astype(np.float32), want[ids1 - 1]), (vt, dim, metric, qi)
    c.set_scan_filter(1)
    assert not c.kernel_name(dg.L1).startswith("scan_filter")     # (no bound for L1: the plain kernel)
    more = dg.corpus(vt, 500, dim, 6900 + dim)
    more[17] = queries[0]
    c.append(more)                                                   # the shadow copy and the row statistics are extended
    ids1, d1 = c.scan_topk(dg.L2, queries[0], 3)
    assert ids1[0] == n + 18 and d1[0] == 0.0
    c.close()


@pytest.mark.parametrize("vt", [dg.F32, dg.F16, dg.U8])
def test_filter_scans_judge_empty_vectors(pkg, orc, vt, monkeypatch):
    """a corpus where one row in twenty-five is all zeros (empty documents): the filter scans judge such a row like any other (its
    estimate and its error term are exactly 0; cosine gives it the reference's distance 1.0) instead of evaluating every one of them
    exactly for every query - same rowids and distance bits as the plain scan, exact evaluations far below the number of empty rows,
    and a query whose best rows ARE the empty ones (everything else beyond cosine distance 1) still finds them."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    monkeypatch.setenv("VG_SCAN_FILTER_NO_GUARD", "1")
    n, dim, k = 1_200_000, 128, 20                                          # (past 2^20 rows: the filter scans run with their pre-pass)
    rows = dg.corpus(vt, n, dim, 9700)
    rows[::25] = 0
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    qs = [dg.query(vt, dim, 9701 + i) for i in range(3)]
    for metric in (dg.L2, dg.DOT, dg.COSINE):
        for q in qs:
            c.set_scan_filter(1)
            c.scan_topk(metric, q, k)                                       # (the shadow copy; a probe)
            c.filter_exact_evals()
            ids1, d1 = c.scan_topk(metric, q, k)
            evals = c.filter_exact_evals()
            assert c.kernel_name(metric).startswith("scan_filter"), c.kernel_name(metric)
            c.set_scan_filter(0)
            ids0, d0 = c.scan_topk(metric, q, k)
            assert ids1.tolist() == ids0.tolist() and dg.same_float_bits(d1, d0), (vt, metric)
            if metric != dg.L2:              # (under L2 the empty rows ARE the nearest rows of a random query, tied at |q|: all of them are looked at)
                assert evals < n // 25 // 2, (vt, metric, evals)            # (48k empty rows: before, every one was a candidate)
    c.close()
    if vt == dg.U8:
        return
    # all-positive rows and an all-negative query: every non-empty row lies beyond cosine distance 1, the empty ones AT 1.0 (the
    # reference's value for a zero norm) - they are the best rows, in scan order
    n = 300_000
    pos = np.abs(dg.storage_to_f64(vt, dg.corpus(vt, n, dim, 9710))).astype(np.float32) + np.float32(0.01)
    pos[::25] = 0
    rows2 = pos if vt == dg.F32 else dg.to_storage(vt, pos)
    qneg = -(np.abs(dg.storage_to_f64(vt, dg.corpus(vt, 1, dim, 9711))).astype(np.float32) + np.float32(0.01))[0]
    qneg = qneg if vt == dg.F32 else dg.to_storage(vt, qneg[None, :])[0]
    c = pkg.Corpus(vt, dim)
    c.append(rows2)
    c.set_scan_filter(1)
    c.scan_topk(dg.COSINE, qneg, k)
    ids1, d1 = c.scan_topk(dg.COSINE, qneg, k)
    assert c.kernel_name(dg.COSINE).startswith("scan_filter")
    c.set_scan_filter(0)
    ids0, d0 = c.scan_topk(dg.COSINE, qneg, k)
    assert ids1.tolist() == ids0.tolist() == [1 + 25 * i for i in range(k)] and np.all(d1 == 1.0) and np.all(d0 == 1.0)
    c.close()


@pytest.mark.parametrize("vt,dim", ((dg.U8, 64), (dg.I8, 48)))
def test_nibble_filter_with_its_prepass_on_clustered_bytes(pkg, orc, vt, dim, monkeypatch):
    """n >= 2^20: the pre-pass threshold; bytes quantized from clustered unit-norm embeddings (values crowd a few levels: the
    high nibbles of whole clusters coincide), near-duplicates of the query"""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    n = (1 << 20) + 911
    rng = np.random.default_rng(7100 + dim)
    centres = rng.standard_normal((11, dim)).astype(np.float32)
    x = centres[rng.integers(0, 11, n)] + np.float32(0.05) * rng.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    lo, hi = x.min(), x.max()
    if vt == dg.U8:
        rows = np.clip(np.rint((x - lo) * (255.0 / (hi - lo))), 0, 255).astype(np.uint8)
    else:
        rows = np.clip(np.rint(x * (127.0 / max(abs(lo), abs(hi)))), -128, 127).astype(np.int8)
    c = pkg.Corpus(vt, dim)
    c.append(rows)
    for qi, q in enumerate((rows[n - 4321].copy(), rows[100].copy(), rows[500000].copy())):
        for metric in (dg.L2, dg.DOT, dg.COSINE):
            c.set_scan_filter(1)
            ids1, d1 = c.scan_topk(metric, q, 20)
            c.set_scan_filter(0)
            ids0, d0 = c.scan_topk(metric, q, 20)
            assert ids1.tolist() == ids0.tolist(), (vt, metric, qi)
            assert dg.same_float_bits(d1, d0), (vt, metric, qi)
            want = orc.scan_distances(orc.AVX2, metric, vt, q, rows[ids1 - 1])
            assert dg.same_float_bits(d1.astype(np.float32), want), (vt, metric, qi)
    c.set_scan_filter(1)
    c.filter_exact_evals()
    c.scan_topk(dg.COSINE, rows[100].copy(), 20)
    assert 0 < c.filter_exact_evals() < n // 4
    c.close()


def test_nibble_filter_probes_the_corpus_before_it_spends_memory_on_it(pkg, orc, monkeypatch):
    """default mode (no explicit scan_filter=1): the first eligible scan of a uint8 corpus runs the nibble filter over a prefix
    and counts the candidates.  Bytes quantized from clustered embeddings are selective -> the filter serves the following scans;
    independent random bytes are not -> the corpus keeps the plain kernel and no shadow copy.  Answers equal the plain scan's
    throughout (the probing scan included)."""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    monkeypatch.delenv("VG_SCAN_FILTER_N4", raising=False)
    n, dim = (1 << 20) + 333, 64
    rng = np.random.default_rng(7300)
    centres = rng.standard_normal((400, dim)).astype(np.float32)       # (a cluster's rows are all candidates of a query inside it: 0.25 % each)
    x = centres[rng.integers(0, 400, n)] + np.float32(0.05) * rng.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    clustered = np.clip(np.rint((x - x.min()) * (255.0 / (x.max() - x.min()))), 0, 255).astype(np.uint8)
    # independent bytes quantized from N(0,1) values (+- 4.5 sigma over 0 .. 255) at 768 elements: the distances concentrate, the
    # slack is ~3 of their standard deviations and most rows are candidates (profiles/r3d, r3f); uniform bytes or short rows do
    # not concentrate that much and pass the probe
    random_bytes = np.clip(np.rint(rng.standard_normal((n, 768), dtype=np.float32) * np.float32(28.3) + np.float32(127.5)), 0, 255).astype(np.uint8)
    for name, rows, want_filter in (("clustered", clustered, True), ("random", random_bytes, False)):
        dim = rows.shape[1]
        c = pkg.Corpus(pkg.U8, dim)
        c.append(rows)
        qs = [rows[n - 77].copy(), rows[5].copy(), rows[400000].copy(), rows[9].copy()]
        c.set_scan_filter(0)
        plain = [c.scan_topk(dg.L2, q, 20) for q in qs]
        c.set_scan_filter(-1)                                      # default: probe, then decide
        c.filter_exact_evals()
        got = [c.scan_topk(dg.L2, q, 20) for q in qs]
        for (a_ids, a_d), (b_ids, b_d) in zip(got, plain):
            assert a_ids.tolist() == b_ids.tolist() and dg.same_float_bits(a_d, b_d), name
        probe_and_first = c.filter_exact_evals()
        assert probe_and_first > 0, name                           # the probe evaluated something either way
        c.scan_topk(dg.L2, qs[1], 20)
        later = c.filter_exact_evals()
        assert (later > 0) == want_filter, (name, later)
        assert c.kernel_name(dg.L2).startswith("scan_filter_u8") == want_filter, (name, c.kernel_name(dg.L2))
        c.close()
