"""CPU-only checks of the mathematics behind the filter scans' lower bounds (no GPU, no extension): the inequalities the kernels
rely on, restated in numpy float64 / exact integers and tried on random, structured and adversarial vectors.

  int8 shadow copy (vg_scan_filter.h, Q8):   |q.x - sq sx (qi.xi)| <= sq |qi| |ex| + |eq| |x|
  high-nibble copy (vg_scan_filter_n4.h):    A + mq Ls - |q'| |l'|  <=  q.x  <=  A + mq Ls + |q'| |l'|
  bf16 shadow copy (vg_scan_filter.h):       |q~.x~ - q.x| <= (2^-7 + 2^-16) |q| |x|      (and NOT 2^-8 (1 + 2^-8): round 1's constant)

The GPU tests (tests/test_gpu_filter_bound.py) check that the kernels built on these return the plain scan's answers."""
import numpy as np
import pytest

import datagen as dg


def q8_split(v):
    v64 = v.astype(np.float64)
    m = float(np.abs(v64).max())
    if m == 0.0:
        return 0.0, np.zeros_like(v64), v64
    s = float(np.float32(np.float32(m) / np.float32(127.0)))
    vi = np.clip(np.rint(v64 * float(np.float32(1.0) / np.float32(s))), -127, 127)
    return s, vi, v64 - s * vi


def adversarial_vectors(rng, dim):
    s = 1.0 / 127.0
    m = rng.integers(-100, 100, dim).astype(np.float64)
    out = [rng.standard_normal(dim), rng.standard_normal(dim) * 1e-6, rng.standard_normal(dim) * 1e6,
           s * (m + 0.499), s * (m - 0.499), s * (np.abs(m) + 0.5 - 1e-7), np.full(dim, 0.3), np.where(np.arange(dim) == 0, 50.0, 0.01),
           rng.standard_normal(dim) * np.exp2(rng.integers(-12, 12, dim)), np.zeros(dim)]
    v = out[3].copy(); v[0] = 1.0; out.append(v)
    return [x.astype(np.float32) for x in out]


@pytest.mark.parametrize("dim", (1, 5, 64, 384, 1000))
def test_int8_shadow_cauchy_schwarz_bound(dim):
    rng = np.random.default_rng(100 + dim)
    vs = adversarial_vectors(rng, dim)
    for q in vs:
        sq, qi, eq = q8_split(q)
        for x in vs:
            sx, xi, ex = q8_split(x)
            true = float((q.astype(np.float64) * x.astype(np.float64)).sum())
            est = sq * sx * float((qi * xi).sum())
            bound = sq * np.linalg.norm(qi) * np.linalg.norm(ex) + np.linalg.norm(eq) * np.linalg.norm(x.astype(np.float64))
            # (+ the f64 cancellation of this test's own true - est; the kernel carries 4e-7 |estimate| for its f32 version of it)
            assert abs(true - est) <= bound * (1 + 1e-9) + 1e-12 * abs(true) + 1e-300, (dim, true, est, bound)


@pytest.mark.parametrize("vt", (dg.U8, dg.I8))
@pytest.mark.parametrize("dim", (1, 7, 32, 100, 768))
def test_high_nibble_interval(vt, dim):
    rng = np.random.default_rng(31 + vt + dim)
    lo_v, hi_v = (0, 255) if vt == dg.U8 else (-128, 127)
    cases = [rng.integers(lo_v, hi_v + 1, dim) for _ in range(30)]
    cases += [np.full(dim, lo_v), np.full(dim, hi_v), np.full(dim, 15), np.full(dim, 16), np.arange(dim) % 16,
              (np.arange(dim) % 2) * 15 + 16 * rng.integers(0, 4, dim), rng.integers(0, 2, dim) * 15]
    cases = [np.clip(c, lo_v, hi_v).astype(np.int64) for c in cases]
    for q in cases:
        mq = q.mean()
        qp = np.linalg.norm(q - mq)
        for x in cases:
            h = np.floor_divide(x, 16)
            l = x - 16 * h
            assert l.min() >= 0 and l.max() <= 15
            A = int((q * 16 * h).sum())
            est = A + mq * l.sum()
            cs = qp * np.linalg.norm(l - 7.5)
            true = int((q * x).sum())
            assert est - cs - 1e-6 * (1 + abs(true)) <= true <= est + cs + 1e-6 * (1 + abs(true)), (vt, dim)
            # the kernel's unpacking: (nibble << 4) as a byte IS x - l, signed for int8, unsigned for uint8
            assert np.array_equal(16 * h, x - l)


def test_bf16_two_sided_rounding_constant():
    """round 1 shipped 2^-8 (1 + 2^-8) - one input's rounding; both the query and the row are rounded"""
    dim = 384
    down = np.float32(1 + 2.0 ** -8 - 2.0 ** -20)                   # just below a bf16 midpoint: rounds down to 1.0
    q = np.full(dim, down, np.float32)
    x = q.copy()
    rq = dg.bf16_bits_to_f32(dg.f32_to_bf16_bits(q)).astype(np.float64)
    rx = dg.bf16_bits_to_f32(dg.f32_to_bf16_bits(x)).astype(np.float64)
    true = float((q.astype(np.float64) * x.astype(np.float64)).sum())
    est = float((rq * rx).sum())
    nn = float(np.linalg.norm(q.astype(np.float64)) * np.linalg.norm(x.astype(np.float64)))
    assert abs(true - est) > 2.0 ** -8 * (1 + 2.0 ** -8) * nn       # the old constant does not cover it
    assert abs(true - est) <= (2.0 ** -7 + 2.0 ** -16) * nn          # the current one does
    rng = np.random.default_rng(5)
    for _ in range(200):                                            # and on signed, scaled, random vectors
        a = (rng.standard_normal(dim) * np.exp2(rng.integers(-8, 8))).astype(np.float32)
        b = (rng.standard_normal(dim) * np.exp2(rng.integers(-8, 8))).astype(np.float32)
        ra = dg.bf16_bits_to_f32(dg.f32_to_bf16_bits(a)).astype(np.float64)
        rb = dg.bf16_bits_to_f32(dg.f32_to_bf16_bits(b)).astype(np.float64)
        t = float((a.astype(np.float64) * b.astype(np.float64)).sum())
        assert abs(t - float((ra * rb).sum())) <= (2.0 ** -7 + 2.0 ** -16) * float(np.linalg.norm(a.astype(np.float64)) * np.linalg.norm(b.astype(np.float64)))


@pytest.mark.parametrize("vt", [dg.F16, dg.BF16])
def test_a_special_element_always_shows_in_the_f64_total(vt):
    """vg_half.h's fast path no longer tests every element pair for Inf / NaN: it reads "this row needs the exact slow path" off the
    row's f64 total.  The claim behind that - every Inf / NaN element of the ROW makes the total non-finite, whatever the (finite)
    query holds, for each of the fast path's four accumulations - restated in numpy with the fast path's own arithmetic (f32
    difference / product for f16, f64 difference for bf16, f64 accumulation), incl. query zeros under an Inf (Inf * 0 = NaN),
    opposite infinities (Inf - Inf = NaN) and several specials per row.  The converse costs time only (finite bf16 rows whose f32
    product overflows take the slow path, which is the reference's own algorithm) and is shown to occur."""
    rng = np.random.default_rng(7)
    specials = [np.inf, -np.inf, np.nan]
    old = np.seterr(all="ignore")
    try:
        for trial in range(400):
            dim = int(rng.integers(1, 70))
            q = dg.storage_to_f64(vt, dg.to_storage(vt, rng.standard_normal((1, dim), dtype=np.float32) * float(np.exp2(rng.integers(-8, 8)))))[0]
            x = dg.storage_to_f64(vt, dg.to_storage(vt, rng.standard_normal((1, dim), dtype=np.float32)))[0]
            if trial % 3 == 0:
                q[rng.integers(0, dim, max(1, dim // 2))] = 0.0
            for _ in range(int(rng.integers(1, 4))):
                x[rng.integers(0, dim)] = specials[int(rng.integers(0, 3))]
            q32, x32 = q.astype(np.float32), x.astype(np.float32)
            if vt == dg.F16:
                d = (q32 - x32).astype(np.float64)            # f32 subtract (distance-avx2.c:186-205)
            else:
                d = q - x                                       # f64 subtract (:383-409)
            totals = {"l2": np.sum(d * d), "l1": np.sum(np.abs(d)), "dot": np.sum((q32 * x32).astype(np.float64))}
            for name, t in totals.items():
                assert not np.isfinite(t), (dg.TYPE_NAMES[vt], name, q, x, t)
        # finite inputs: f64 cannot overflow on 16-bit elements (|term| < 2^256), so a finite row is flagged only through an
        # overflowing f32 PRODUCT - possible for bf16 (3e38 * 3e38), impossible for f16 (65504^2 < 2^32)
        big = np.float32(3.0e38)
        assert not np.isfinite(np.float64(big * big)) and np.isfinite(np.float64(big) - np.float64(-big)) and np.isfinite((np.float64(big) * 2) ** 2)
        assert np.isfinite(np.float32(65504.0) * np.float32(65504.0))
    finally:
        np.seterr(**old)
