"""CPU-only checks of the drop-in boundary: the C-ABI library loads here (no GPU) and exports every symbol
include/vectorgpu.h declares; without a device the compute entry points fail loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    g._load_build().build_gpu_library()
    return g.load_package()


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", hdr))


def test_every_declared_symbol_is_exported(pkg):
    """include/vectorgpu.h is the surface a binding keeps stable; include/vectorgpu_diag.h the test / bench hooks (profiling, plans,
    counters, building blocks) - both sets must be exported, and the stable one must stay free of the hooks"""
    core, diag = _declared("vectorgpu.h"), _declared("vectorgpu_diag.h")
    assert len(core) >= 25 and not (core & diag)
    for hook in core:
        assert not re.search(r"(_plan|_stats|_ms|_ms_ex|_evals|_profiling|_kernel_name|_stat_)", hook), hook
    declared = core | diag
    lib = pkg.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libvectorgpu.so does not export %s" % name
    # and the ctypes view binds exactly the declared set
    assert declared == set(lib._sig.keys())


def test_no_silent_cpu_fallback_without_device(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.VectorGpuError) as ei:
        pkg.Corpus(pkg.F32, 8)
    assert "no HIP device" in str(ei.value)
    assert "no device" in pkg.backend_name()


def test_query_quantizer_matches_oracle(pkg, orc):
    """vg_quantize_query is host C in the product; it must agree bit for bit with the pinned oracle quantizer."""
    import datagen as dg
    rng = np.random.default_rng(3)
    for vt in dg.ALL_TYPES:
        for qt in (pkg.QUANT_U8, pkg.QUANT_S8):
            for trial in range(4):
                dim = int(rng.integers(1, 300))
                src = dg.corpus(vt, 1, dim, 100 + trial)[0]
                if vt in (dg.F16, dg.BF16) and dim > 4:
                    src[1] = dg.F16_INF if vt == dg.F16 else dg.BF_INF
                    src[2] = dg.F16_NAN if vt == dg.F16 else dg.BF_NAN
                    src[3] = dg.F16_NINF if vt == dg.F16 else dg.BF_NINF
                if vt == dg.F32 and dim > 3:
                    src[1] = np.float32(np.nan); src[2] = np.float32(1e30); src[3] = np.float32(-np.inf)
                scale = float(np.float32(rng.uniform(0.5, 80.0)))
                offset = float(np.float32(rng.uniform(-2.0, 2.0))) if qt == pkg.QUANT_U8 else 0.0
                got = pkg.quantize_query(vt, src, scale, offset, qt)
                want = orc.quantize(vt, src, offset, scale, qt)
                assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (vt, qt, dim)


def test_merge_keys_host_helper(pkg):
    """k-way merge of per-shard key lists: ties resolve in global scan order (shard index, then position)."""
    lib = pkg.lib()

    def key(d, pos):
        b = np.float32(d).view(np.uint32).item()
        s = b ^ (0xFFFFFFFF if b >> 31 else 0x80000000)
        return (s << 32) | pos

    E = pkg.KEY_EMPTY
    lists = np.array([[key(1.0, 5), key(2.0, 1), key(2.0, 9), E],
                      [key(-3.0, 2), key(2.0, 0), E, E],
                      [E, E, E, E]], dtype=np.uint64)
    pos, dist = pkg.merge_keys(lists, [0, 100, 200], 5)
    assert dist.tolist() == [-3.0, 1.0, 2.0, 2.0, 2.0]
    assert pos.tolist() == [102, 5, 1, 9, 100]
    assert abs(lib.vg_key_distance(C.c_uint64(key(0.25, 7))) - 0.25) == 0 and lib.vg_key_position(C.c_uint64(key(0.25, 7))) == 7
    pos, dist = pkg.merge_keys(lists, None, 10)
    assert len(pos) == 5


def test_launch_shape_rules(pkg, monkeypatch):
    """vg_plan_scan_shape (pure host logic): the (lanes per row x 16-byte chunks per lane) decomposition the plain scan kernels are
    launched with, pinned for the BASELINE configurations and for every rule DESIGN.md section 3 states with a measurement behind it -
    so that a change to one rule cannot silently move another shape."""
    for name in ("VG_LPR_LOG2", "VG_U", "VG_SHAPE_PREF_ROUND1", "VG_SHAPE_F16_ROUND3", "VG_SHAPE_BF16_L2_U3", "VG_SHAPE_INT_SHORT_ROUND3", "VG_FORCE_LONG"):
        monkeypatch.delenv(name, raising=False)
    F32, F16, BF16, U8, I8 = pkg.F32, pkg.F16, pkg.BF16, pkg.U8, pkg.I8
    L2, COS, DOT, L1 = pkg.L2, pkg.COSINE, pkg.DOT, pkg.L1
    want = {
        # the BASELINE configurations: 3 chunks per lane, twice the lanes (512 contiguous bytes of a row per load instruction)
        (F32, 384, L2): (32, 3), (F32, 384, DOT): (32, 3), (U8, 768, COS): (16, 3), (I8, 768, L2): (16, 3), (U8, 384, L2): (8, 3),
        (F32, 768, L2): (64, 3),
        # f16 follows the same rule since the batch loads are unconditional; bf16: at most 3 chunks per lane, except L2
        (F16, 384, L2): (16, 3), (F16, 384, COS): (16, 3), (F16, 768, DOT): (32, 3),
        (BF16, 384, COS): (16, 3), (BF16, 384, DOT): (16, 3), (BF16, 384, L1): (16, 3), (BF16, 384, L2): (8, 6), (BF16, 768, COS): (32, 3),
        # 4 chunks per lane are not traded for 2 at twice the lanes
        (F32, 128, L2): (8, 4), (F32, 256, L2): (16, 4), (F32, 512, L2): (32, 4), (F32, 1024, L2): (64, 4), (U8, 512, L2): (8, 4),
        (U8, 1024, COS): (16, 4), (F16, 256, L2): (8, 4), (F16, 512, COS): (16, 4), (F16, 1024, L2): (32, 4),
        # short rows (3 .. 8 chunks): 2 chunks per lane for the float types, one for uint8 / int8
        (F32, 32, L2): (4, 2), (F32, 32, COS): (4, 2), (F16, 64, L2): (4, 2), (F16, 64, DOT): (4, 2), (BF16, 64, COS): (4, 2),
        (U8, 64, L2): (4, 1), (U8, 64, COS): (4, 1), (U8, 64, DOT): (4, 1), (I8, 128, L2): (8, 1), (U8, 128, COS): (8, 1), (U8, 100, L2): (8, 1),
        (U8, 256, L2): (8, 2), (F16, 128, L2): (8, 2),
        # rows no shape covers exactly: 2 chunks per lane over 4, up to 16 lanes per row
        (F32, 100, L2): (16, 2), (F32, 100, COS): (16, 2), (F16, 200, L2): (16, 2), (F32, 200, L2): (16, 4),
    }
    got = {key: pkg.plan_scan_shape(*key)[:2] for key in want}
    assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}
    # every shape covers its row, wastes less than half of its lane slots, and very long rows take the long-row kernel
    for vt in (F32, F16, BF16, U8, I8):
        es = pkg.TYPE_SIZE[vt]
        for dim in list(range(1, 200)) + [255, 256, 300, 384, 512, 700, 768, 1000, 1024, 1536, 2048]:
            for metric in (L2, COS, DOT, L1):
                lpr, u, long_rows = pkg.plan_scan_shape(vt, dim, metric)
                nch = (dim * es + 15) // 16
                if long_rows:
                    assert nch > 64
                    continue
                assert lpr * u >= nch and lpr in (1, 2, 4, 8, 16, 32, 64) and u in (1, 2, 3, 4, 6, 8), (vt, dim, metric, lpr, u)
                assert lpr * u < 2 * nch or nch == 1, (vt, dim, metric, lpr, u)
    assert pkg.plan_scan_shape(F32, 4096, L2)[2] and pkg.plan_scan_shape(F16, 4096, L2)[2]
    with pytest.raises(pkg.VectorGpuError):
        pkg.plan_scan_shape(9, 384, L2)
    # the experiment overrides (VG_LPR_LOG2 / VG_U, the four VG_SHAPE_* rules of earlier rounds) are -DVG_LAB switches since round 6:
    # a product build does not read them
    monkeypatch.setenv("VG_LPR_LOG2", "4"); monkeypatch.setenv("VG_U", "6")
    pkg.reload_switches()
    try:
        assert pkg.plan_scan_shape(F32, 384, L2)[:2] == (32, 3)
    finally:
        monkeypatch.delenv("VG_LPR_LOG2"); monkeypatch.delenv("VG_U")
        pkg.reload_switches()


def test_half_batch_workgroup_form_plan(pkg, monkeypatch):
    """vg_batch_h_plan (host logic): two 4-wavefront workgroups per CU only for rows of up to 768 bytes, while BOTH workgroups' LDS
    (tiles + 32 k-key lists per wavefront) fits the CU's 160 KB and the batch fills two workgroups per partition; one workgroup of
    eight otherwise (one of four for rows of 1 - 2 KiB, whose query block alone takes the register file)."""
    monkeypatch.delenv("VG_BATCH_H_WAVES", raising=False)
    plan = pkg.plan_batch_half_form
    # the split form (VG_BATCH_H_SPLIT=1: filter kernel + exact-evaluation kernel): ONE workgroup per CU, a tile streamed once for
    # all its queries, whatever k and the batch size (the lists do not live in the streaming kernel's LDS)
    monkeypatch.setenv("VG_BATCH_H_SPLIT", "1")
    assert plan(768, 20, 1024) == (8, 1) and plan(768, 32, 100) == (8, 1) and plan(64, 1, 4096) == (8, 1) and plan(1024, 20, 1024) == (8, 1)
    assert plan(1536, 20, 1024) == (4, 1) and plan(2048, 32, 1024) == (4, 1)
    assert plan(2049, 10, 1024) is None and plan(768, 0, 10) is None and plan(768, 33, 10) is None
    monkeypatch.delenv("VG_BATCH_H_SPLIT")               # the default: the fused kernel's forms
    assert plan(768, 20, 1024) == (4, 2) and plan(768, 27, 1024) == (4, 2) and plan(768, 28, 1024) == (8, 1) and plan(768, 32, 300) == (8, 1)
    assert plan(768, 20, 129) == (4, 2) and plan(768, 20, 128) == (8, 1) and plan(768, 20, 1) == (8, 1)
    assert plan(512, 32, 1024) == (4, 2) and plan(256, 32, 300) == (4, 2) and plan(64, 1, 4096) == (4, 2)
    assert plan(1024, 20, 1024) == (8, 1) and plan(800, 20, 1024) == (8, 1)          # 1 KiB rows: no second form
    assert plan(1536, 20, 1024) == (4, 1) and plan(2048, 10, 1024) == (4, 1)          # long rows: one 4-wavefront workgroup
    assert plan(2049, 10, 1024) is None and plan(768, 0, 10) is None and plan(768, 33, 10) is None
    for k in range(1, 33):                                                           # the chosen form always fits
        for stride in (32, 256, 512, 768, 1024, 1536, 2048):
            if plan(stride, k, 1024) is None:
                assert stride == 2048 and k >= 30                                     # (the one combination whose lists do not fit at all)
                continue
            w, b = plan(stride, k, 1024)
            ntb = [n for n in (8, 16, 24, 32, 48, 64) if n * 32 >= stride][0]
            lds = 2 * ntb * 1024 + 1024 + w * 32 * 16 + w * 32 * k * 8
            assert b * lds <= 160 * 1024, (stride, k, w, b, lds)
    monkeypatch.setenv("VG_BATCH_H_WAVES", "8")
    assert plan(768, 20, 1024) == (8, 1)
    monkeypatch.setenv("VG_BATCH_H_WAVES", "4")
    assert plan(768, 20, 100) == (4, 2) and plan(768, 30, 1024) == (8, 1) and plan(1024, 20, 1024) == (8, 1)
