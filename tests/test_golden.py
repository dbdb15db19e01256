"""Oracle vs committed golden fixtures (tests/golden/*.npz, produced by executing the reference with
tests/golden/make_golden.py).  Needs neither /root/reference nor a GPU: this is the pin that travels."""
import os
import sys

import numpy as np
import pytest

import datagen as dg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402  (only its case tables are used here, nothing is generated)


@pytest.fixture(scope="module")
def kernels():
    return np.load(os.path.join(HERE, "golden", "kernels.npz"))


@pytest.fixture(scope="module")
def sql():
    return np.load(os.path.join(HERE, "golden", "sql.npz"))


def _eq_bits(got_f32, want_u32):
    return dg.same_float_bits(got_f32, want_u32.view(np.float32))


@pytest.mark.parametrize("which", ["cpu", "avx2"])
@pytest.mark.parametrize("vt", dg.ALL_TYPES)
def test_oracle_kernels_match_golden(orc, kernels, which, vt):
    be = orc.CPU if which == "cpu" else orc.AVX2
    for (dim, n, seed) in mg.KERNEL_CASES:
        rows = dg.corpus(vt, n, dim, seed)
        q = dg.query(vt, dim, seed + 7)
        for m in dg.ALL_METRICS:
            want = kernels["%s/rand/%s/%s/%d" % (which, dg.TYPE_NAMES[vt], dg.METRIC_NAMES[m], dim)]
            got = np.array([orc.distance(be, m, vt, q, rows[i]) for i in range(n)], dtype=np.float32)
            assert _eq_bits(got, want), (which, vt, m, dim)
    for dim in mg.EDGE_DIMS:
        _, rows = dg.edge_rows(vt, dim, 1000 + dim)
        for qi, q in enumerate(dg.edge_queries(vt, dim, 2000 + dim)):
            for m in dg.ALL_METRICS:
                want = kernels["%s/edge/%s/%s/%d/%d" % (which, dg.TYPE_NAMES[vt], dg.METRIC_NAMES[m], dim, qi)]
                got = np.array([orc.distance(be, m, vt, q, rows[i]) for i in range(rows.shape[0])], dtype=np.float32)
                assert _eq_bits(got, want), (which, vt, m, dim, qi)


@pytest.mark.parametrize("which", ["cpu", "avx2"])
@pytest.mark.parametrize("case", mg.SQL_SCAN_CASES, ids=[c[0] for c in mg.SQL_SCAN_CASES])
def test_oracle_full_scan_matches_golden(orc, sql, which, case):
    """scan + clamp + reference slot top-k == what vector_full_scan returned (rowids AND distance bits)."""
    name, vt, metric, n, dim, k, seed, low = case
    be = orc.CPU if which == "cpu" else orc.AVX2
    rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
    q = dg.query(vt, dim, seed + 1, low_entropy=low)
    ids, dist = orc.scan_topk_reference(be, metric, vt, q, rows, None, k)
    assert ids.tolist() == sql["%s/%s/rowids" % (which, name)].tolist()
    assert _eq_bits(dist.astype(np.float32), sql["%s/%s/dist" % (which, name)])
    # the (distance, position) total order gives the same distance sequence; rowids too unless a tie straddles k
    d = orc.scan_distances(be, metric, vt, q, rows)
    ids2, dist2, _ = orc.topk_ordered(d, None, k)
    assert _eq_bits(dist2.astype(np.float32), sql["%s/%s/dist" % (which, name)])
    kth = np.sort(d[d < np.inf])[: k + 1]
    if len(np.unique(kth)) == len(kth):
        assert ids2.tolist() == ids.tolist()


@pytest.mark.parametrize("which", ["cpu", "avx2"])
@pytest.mark.parametrize("case", mg.SQL_QUANT_CASES, ids=[c[0] for c in mg.SQL_QUANT_CASES])
def test_oracle_quantize_matches_golden(orc, sql, which, case):
    name, vt, qopt, n, dim, k, seed, nonneg = case
    be = orc.CPU if which == "cpu" else orc.AVX2
    rows = dg.corpus(vt, n, dim, seed)
    if nonneg:
        rows = np.abs(rows)
    q = dg.query(vt, dim, seed + 1)
    qt0 = {None: 0, "UINT8": orc.QUANT_U8, "INT8": orc.QUANT_S8}[qopt]
    qt, scale, offset = orc.quant_params(vt, rows, qt0)
    want = sql["%s/%s/qparams" % (which, name)]
    assert qt == int(want[0])
    assert np.float32(scale) == np.float32(want[1]) and np.float32(offset) == np.float32(want[2])
    qrows = np.stack([orc.quantize(vt, rows[i], offset, scale, qt) for i in range(n)]).view(np.uint8)
    assert np.array_equal(qrows[:64], sql["%s/%s/qhead" % (which, name)])
    assert np.array_equal(qrows.astype(np.uint32).sum(axis=1).astype(np.uint32), sql["%s/%s/qrowsum" % (which, name)])
    qq = orc.quantize(vt, q, offset, scale, qt)
    ivt = dg.U8 if qt == orc.QUANT_U8 else dg.I8
    ids, dist = orc.scan_topk_reference(be, dg.COSINE, ivt, qq, qrows.view(dg.NP_DTYPE[ivt]), None, k)
    assert ids.tolist() == sql["%s/%s/rowids" % (which, name)].tolist()
    assert _eq_bits(dist.astype(np.float32), sql["%s/%s/dist" % (which, name)])


def test_conversions_roundtrip(orc):
    """f16/bf16 conversions agree with numpy's IEEE implementation over all 65536 bit patterns."""
    bits = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    f = np.array([orc.lib().orc_f16_to_f32(int(b)) for b in bits], dtype=np.float32)
    want = bits.view(np.float16).astype(np.float32)
    assert dg.same_float_bits(f, want)
    back = np.array([orc.lib().orc_f32_to_f16(float(x)) if not np.isnan(x) else 0 for x in want], dtype=np.uint16)
    ok = np.isnan(want) | (back == bits)
    assert ok.all()
    b32 = np.array([orc.lib().orc_bf16_to_f32(int(b)) for b in bits], dtype=np.float32)
    assert dg.same_float_bits(b32, dg.bf16_bits_to_f32(bits))
    rng = np.random.default_rng(1)
    x = rng.standard_normal(5000).astype(np.float32) * np.float32(1e3)
    got = np.array([orc.lib().orc_f32_to_bf16(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got, dg.f32_to_bf16_bits(x))
    got16 = np.array([orc.lib().orc_f32_to_f16(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got16, dg.f32_to_f16_bits(x))
