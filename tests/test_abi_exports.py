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


def test_every_declared_symbol_is_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "vectorgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
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
