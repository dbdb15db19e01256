"""The default paths on data that is NOT iid: 10M x 384 f32 drawn from 4 096 Gaussian clusters, L2-normalised - the shape of
a table of sentence embeddings, what the reference's own example feeds it (examples/semantic_search/semantic_search.py:125-165) - cosine
and dot, queries near cluster centres; and the same clusters quantized to uint8 x 768 (the reference's formula over the corpus' min / max).

Held to the REFERENCE'S OWN KERNEL over every row exactly as tests/test_gpu_reference_parity.py holds the N(0,1) corpora (RefScanner:
oracle/_ref/libref_avx2.so, distance-avx2.c inside the reference's top-k loop, block by block): the plain kernel, the default
single-query path (int8 shadow filter / nibble filter or whatever the probe chose) and the default 1024-query batch (int8 matrix-core
filter).  f32: distances within 1e-5 relative, rowids identical at every separated rank.  uint8: bit-exact distances.

Plus the adversarial corpus: every row within ~1e-3 of every query.  No lower bound separates anything; the selectivity guard must hand
the queries to the plain kernel, with the plain kernel's answer, at <= 1.15 x the plain kernel's time.
"""
import time

import numpy as np
import pytest

import datagen as dg
from test_gpu_reference_parity import RefScanner, check_against_reference, dot_tol

pytestmark = pytest.mark.gpu

BLOCK = 500_000


@pytest.fixture(scope="module")
def env():
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    if pkg.device_count() < 1:
        pytest.fail("needs a GPU")
    return pkg, torch


def test_clustered_unit_norm_f32_default_paths_equal_the_reference_kernels_at_10m(env, orc):
    pkg, torch = env
    n, dim, k, nq = 10_000_000, 384, 20, 1024
    centres = dg.clustered_centres(torch, 42, dim)
    qs = dg.clustered_queries(torch, centres, 42, nq)
    assert np.allclose(np.linalg.norm(qs, axis=1), 1.0, atol=1e-5)
    sample = list(range(0, nq, 128))                                        # 8 queries
    metrics = (dg.COSINE, dg.DOT)
    scanners = {m: RefScanner(orc, m, qs[sample], k + 1) for m in metrics}
    pinned = torch.empty((BLOCK, dim), dtype=torch.float32).pin_memory()
    c = pkg.Corpus(pkg.F32, dim, capacity=n)
    for b in range(n // BLOCK):
        t = dg.clustered_block(torch, centres, 42, b, BLOCK)
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), BLOCK, dim * 4)
        pinned.copy_(t)
        torch.cuda.synchronize()
        for m in metrics:
            scanners[m].feed(pinned.numpy(), b * BLOCK)
        del t
    checked = 0
    for m in metrics:
        # unit-norm rows: |d| of a winner is ~0.1 (cosine) / ~0.9 (dot) against sum |q_i x_i| ~ 0.6: the literal 1e-5 relative bar, the
        # sum |q_i x_i| term only for a cancelled dot product
        tol = (lambda d: 1e-5 * abs(d) + 2e-7) if m == dg.COSINE else dot_tol(1.0)
        c.set_scan_filter(0)
        plain = [c.scan_topk(m, qs[qi], k) for qi in sample]
        assert c.kernel_name(m).startswith("scan_f32"), c.kernel_name(m)
        c.set_scan_filter(-1)
        c.scan_topk(m, qs[0], k)                                            # (builds the shadow copy)
        c.filter_exact_evals()
        dflt = [c.scan_topk(m, qs[qi], k) for qi in sample]
        evals = c.filter_exact_evals() / float(len(sample))
        assert c.kernel_name(m).startswith("scan_filter_f32"), c.kernel_name(m)      # clustered data IS selective under the int8 bound
        assert evals < 0.01 * n, evals
        ids_b, dist_b, cnt_b = c.scan_topk_batch(m, qs, k)
        assert c.last_batch_path() == 7, c.last_batch_path()                # the int8 matrix-core filter
        assert np.all(cnt_b == k)
        for j, qi in enumerate(sample):
            ref = scanners[m].result(j)
            for name, (ids, dist) in (("plain", plain[j]), ("default", dflt[j]), ("batch", (ids_b[qi], dist_b[qi]))):
                checked += check_against_reference((m, name, qi), ids, dist, ref, k, tol)
            # the single-query filter hands its survivors to the plain kernel's arithmetic: the same rows, the same floats
            assert np.array_equal(plain[j][0], dflt[j][0]) and np.array_equal(plain[j][1], dflt[j][1]), (m, qi)
            # the batch's exact evaluation sums a row over 64 lanes, the plain kernel over 32: another f32 summation order, a few ulp apart
            # (each is held to the reference above); the rows are the same wherever two distances are not within that
            assert np.allclose(plain[j][1], dist_b[qi], rtol=4e-6, atol=4e-7), (m, qi)
            differ = np.nonzero(plain[j][0] != ids_b[qi])[0]
            for s_ in differ.tolist():
                near = [t_ for t_ in (s_ - 1, s_ + 1) if 0 <= t_ < k]
                assert plain[j][0][s_] in [ids_b[qi][t_] for t_ in near] or s_ == k - 1, (m, qi, s_)
    assert checked >= 2 * 3 * len(sample) * (k - 6), checked
    c.close()


def test_clustered_uint8_768_default_paths_equal_the_reference_kernels_at_10m(env, orc):
    pkg, torch = env
    n, dim, k, nq = 10_000_000, 768, 20, 1024
    centres = dg.clustered_centres(torch, 43, dim)
    lo, hi = float("inf"), float("-inf")
    for b in range(0, n // BLOCK, 5):
        t = dg.clustered_block(torch, centres, 43, b, BLOCK)
        lo, hi = min(lo, float(t.min())), max(hi, float(t.max()))
        del t
    scale = 255.0 / (hi - lo)

    def q8(t):                                                               # sqlite-vector.c:517-548 with offset = lo, scale = 255 / (hi - lo)
        return torch.clamp(torch.floor((t - lo) * scale + 0.5), 0, 255).to(torch.uint8)
    qs = q8(dg.clustered_block(torch, centres, 43 + 977, 0, nq, noise=dg.QUERY_NOISE)).cpu().numpy()
    sample = list(range(0, nq, 256))                                        # 4 queries (the reference's uint8 cosine is three passes per row)

    class U8Scanner(RefScanner):
        def _one(self, qi, rows, row0):
            if self.ref is not None:
                ids, d = self.ref.scan_topk(self.metric, dg.U8, self.queries[qi], rows, self.k1)
            else:
                ids, d = self.orc.scan_topk_reference(self.orc.AVX2, self.metric, dg.U8, self.queries[qi], rows, None, self.k1)
            return qi, [(float(np.float32(x)), int(i) - 1 + row0) for i, x in zip(ids.tolist(), d.tolist())]
    scanner = U8Scanner(orc, dg.COSINE, qs[sample], k + 1)
    pinned = torch.empty((BLOCK, dim), dtype=torch.uint8).pin_memory()
    c = pkg.Corpus(pkg.U8, dim, capacity=n)
    for b in range(n // BLOCK):
        t = q8(dg.clustered_block(torch, centres, 43, b, BLOCK))
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), BLOCK, dim)
        pinned.copy_(t)
        torch.cuda.synchronize()
        scanner.feed(pinned.numpy(), b * BLOCK)
        del t
    c.set_scan_filter(0)
    plain = [c.scan_topk(dg.COSINE, qs[qi], k) for qi in sample]
    c.set_scan_filter(-1)
    c.scan_topk(dg.COSINE, qs[0], k)
    dflt = [c.scan_topk(dg.COSINE, qs[qi], k) for qi in sample]
    ids_b, dist_b, cnt_b = c.scan_topk_batch(dg.COSINE, qs, k)
    assert np.all(cnt_b == k)
    for j, qi in enumerate(sample):
        ref = scanner.result(j)
        rd = np.array([d for d, _ in ref[:k]], dtype=np.float32)
        for name, (ids, dist) in (("plain", plain[j]), ("default", dflt[j]), ("batch", (ids_b[qi], dist_b[qi]))):
            # integer sums are exact: the distances are the reference's bit for bit; rowids wherever the reference's distances are distinct
            assert np.array_equal(np.asarray(dist, dtype=np.float32), rd), (name, qi, dist, rd)
            full = [d for d, _ in ref]
            for i in range(k):
                if (i == 0 or full[i] != full[i - 1]) and full[i + 1] != full[i]:
                    assert int(ids[i]) == ref[i][1] + 1, (name, qi, i)
        assert np.array_equal(plain[j][0], dflt[j][0]) and np.array_equal(plain[j][0], ids_b[qi]), qi
    c.close()


def test_adversarial_corpus_goes_back_to_the_plain_kernel(env, orc):
    """every row within ~1e-3 of every query: the int8 bound passes everything.  The guard must notice and hand the queries to the plain
    kernel - same answer, and no more than 1.15 x the plain kernel's time per query"""
    pkg, torch = env
    n, dim, k = 2_000_000, 384, 20
    c = pkg.Corpus(pkg.F32, dim, capacity=n)
    for b in range(n // BLOCK):
        t = dg.adversarial_block(torch, 7, b, BLOCK, dim)
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), BLOCK, dim * 4)
        del t
    qs = dg.adversarial_block(torch, 7, 9999, 32, dim).cpu().numpy()

    def timed(nrep=20):
        for i in range(4):
            c.scan_topk(dg.COSINE, qs[i], k)
        t0 = time.perf_counter()
        for i in range(nrep):
            c.scan_topk(dg.COSINE, qs[(4 + i) % 32], k)
        return (time.perf_counter() - t0) / nrep
    c.set_scan_filter(0)
    t_plain = min(timed(), timed())
    plain = [c.scan_topk(dg.COSINE, qs[i], k) for i in range(8)]
    c.set_scan_filter(-1)
    for i in range(6):                                                      # the first queries run the filter and trip the guard
        c.scan_topk(dg.COSINE, qs[i], k)
    t_dflt = min(timed(), timed())
    dflt = [c.scan_topk(dg.COSINE, qs[i], k) for i in range(8)]
    assert c.filter_guard_cooldown() > 0, "the guard did not trip"          # the next scans of this corpus take the plain kernel
    for i in range(8):
        assert np.array_equal(plain[i][0], dflt[i][0]) and np.array_equal(plain[i][1], dflt[i][1]), i
    assert t_dflt <= 1.15 * t_plain, (t_dflt, t_plain)
    # and a batch over it (32 queries: the f32 matrix-core kernel - another summation order over rows that all but tie): every returned
    # distance within the f32 bar of the plain scan's at the same rank, and none of them better than the plain scan's best
    ids_b, dist_b, cnt_b = c.scan_topk_batch(dg.COSINE, qs, k)
    assert np.all(cnt_b == k)
    for i in range(8):
        assert np.allclose(plain[i][1], dist_b[i], rtol=1e-5, atol=2e-6), (i, c.last_batch_path(), plain[i][1], dist_b[i])   # (1 - cos of all-but-identical vectors: ~3e-6, all cancellation)
    c.close()
