"""Seeded synthetic inputs shared by the oracle tests, the golden-fixture generator and the GPU parity tests.

Everything is a pure function of (seed, shape, type) so that fixtures can be regenerated anywhere.
Types/metrics use the reference's enum values (distance-cpu.h:36-58).
"""
import numpy as np

F32, F16, BF16, U8, I8 = 1, 2, 3, 4, 5
L2, SQUARED_L2, COSINE, DOT, L1 = 1, 2, 3, 4, 5
ALL_TYPES = (F32, F16, BF16, U8, I8)
ALL_METRICS = (L2, SQUARED_L2, COSINE, DOT, L1)
TYPE_NAMES = {F32: "f32", F16: "f16", BF16: "bf16", U8: "u8", I8: "i8"}
METRIC_NAMES = {L2: "l2", SQUARED_L2: "sql2", COSINE: "cosine", DOT: "dot", L1: "l1"}
NP_DTYPE = {F32: np.float32, F16: np.uint16, BF16: np.uint16, U8: np.uint8, I8: np.int8}


def f32_to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_f16_bits(x):
    with np.errstate(over="ignore"):
        return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)


def to_storage(vtype, x):
    """float32 values -> storage array of vtype (raw bits for f16/bf16, rounded+clipped for ints)."""
    x = np.asarray(x, dtype=np.float32)
    if vtype == F32:
        return np.ascontiguousarray(x)
    if vtype == F16:
        return f32_to_f16_bits(x)
    if vtype == BF16:
        return f32_to_bf16_bits(x)
    if vtype == U8:
        return np.clip(np.rint(x), 0, 255).astype(np.uint8)
    if vtype == I8:
        return np.clip(np.rint(x), -128, 127).astype(np.int8)
    raise ValueError(vtype)


def storage_to_f64(vtype, a):
    if vtype == F32:
        return a.astype(np.float64)
    if vtype == F16:
        return a.view(np.float16).astype(np.float64)
    if vtype == BF16:
        return bf16_bits_to_f32(a).astype(np.float64)
    return a.astype(np.float64)


def corpus(vtype, n, dim, seed, low_entropy=False):
    """(n, dim) storage array.  Floats ~ N(0,1); u8 ~ U[0,256); i8 ~ U[-128,128).
    low_entropy: values in a 16-level range so that integer distances tie often."""
    rng = np.random.default_rng(seed)
    if vtype in (F32, F16, BF16):
        x = rng.standard_normal((n, dim), dtype=np.float32)
        if low_entropy:
            x = np.rint(x * 2.0).astype(np.float32) / 2.0
        return to_storage(vtype, x)
    if vtype == U8:
        hi = 16 if low_entropy else 256
        return rng.integers(0, hi, (n, dim), dtype=np.int64).astype(np.uint8)
    hi = 8 if low_entropy else 128
    return rng.integers(-hi, hi, (n, dim), dtype=np.int64).astype(np.int8)


def query(vtype, dim, seed, low_entropy=False):
    return corpus(vtype, 1, dim, seed, low_entropy)[0].copy()


# special bit patterns
F16_INF, F16_NINF, F16_NAN, F16_MAX, F16_SUB = 0x7C00, 0xFC00, 0x7E01, 0x7BFF, 0x0001
BF_INF, BF_NINF, BF_NAN, BF_MAX, BF_SUB = 0x7F80, 0xFF80, 0x7FC1, 0x7F7F, 0x0001


def edge_rows(vtype, dim, seed):
    """A small corpus of hand-made rows around a random base: zeros, exact copies of the query (distance 0),
    huge/tiny magnitudes, and for f16/bf16 NaN / +-Inf / subnormal lanes in both block and tail positions."""
    rng = np.random.default_rng(seed)
    q = query(vtype, dim, seed + 1)
    rows = [q.copy(), np.zeros(dim, dtype=NP_DTYPE[vtype])]
    base = corpus(vtype, 8, dim, seed + 2)
    rows += [base[i] for i in range(8)]
    if vtype == F32:
        r = base[0].copy(); r[:] = 1e-30; rows.append(r)
        r = base[1].copy(); r[:] *= np.float32(1e18); rows.append(r)           # squares overflow -> inf
        r = q.copy(); r[0] = np.nextafter(r[0], np.float32(np.inf)); rows.append(r)   # |d| below the 8*eps clamp
        r = q.copy(); r[-1] = np.float32(np.nan); rows.append(r)              # NaN distance: never in top-k
        r = q.copy(); r[dim // 2] = np.float32(np.inf); rows.append(r)
    elif vtype in (F16, BF16):
        inf, ninf, nan, mx, sub = (F16_INF, F16_NINF, F16_NAN, F16_MAX, F16_SUB) if vtype == F16 else \
                                  (BF_INF, BF_NINF, BF_NAN, BF_MAX, BF_SUB)
        pos_list = sorted(set([0, min(3, dim - 1), min(8, dim - 1), dim - 1]))
        for pos in pos_list:
            for pat in (inf, ninf, nan, mx, sub):
                r = base[int(rng.integers(0, 8))].copy(); r[pos] = pat; rows.append(r)
        r = base[2].copy(); r[0] = inf; r[dim - 1] = ninf; rows.append(r)
        r = base[3].copy(); r[:] = mx; rows.append(r)
        r = base[4].copy(); r[:] = sub; rows.append(r)
    else:
        info = np.iinfo(NP_DTYPE[vtype])
        r = base[0].copy(); r[:] = info.max; rows.append(r)
        r = base[1].copy(); r[:] = info.min; rows.append(r)
        r = q.copy(); r[0] = r[0] + 1 if r[0] < info.max else r[0] - 1; rows.append(r)
    return q, np.ascontiguousarray(np.stack(rows))


def edge_queries(vtype, dim, seed):
    """Queries that exercise the query-side special cases (zero norm, Inf/NaN lanes)."""
    q = query(vtype, dim, seed)
    out = [q, np.zeros(dim, dtype=NP_DTYPE[vtype])]
    if vtype in (F16, BF16):
        inf, nan = (F16_INF, F16_NAN) if vtype == F16 else (BF_INF, BF_NAN)
        a = q.copy(); a[0] = inf; out.append(a)
        a = q.copy(); a[dim - 1] = nan; out.append(a)
    if vtype == F32:
        a = q.copy(); a[0] = np.float32(np.inf); out.append(a)
    return out


def float_bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def same_float_bits(a, b):
    """bitwise equality with all NaNs considered equal."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))


# ---- clustered, unit-norm data (round 6): what sentence-embedding tables look like, instead of iid N(0,1) - the shape the reference's own
# example feeds it (examples/semantic_search/semantic_search.py:125-165: L2-normalised embeddings, cosine).  Pure functions of (seed, block):
# bench.py's `also.clustered` leg and tests/test_gpu_clustered.py regenerate the same rows.  `torch` is passed in (this module stays importable
# without it).
CLUSTERS = 4096
CLUSTER_NOISE = 0.5            # norm of the within-cluster noise against the unit centre: cos(row, its centre) ~ 0.89
QUERY_NOISE = 0.3


def clustered_centres(torch, seed, dim, n_clusters=CLUSTERS, device="cuda"):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed * 7919 + 11)
    c = torch.randn((n_clusters, dim), generator=gen, device=device, dtype=torch.float32)
    return c / c.norm(dim=1, keepdim=True)


def clustered_block(torch, centres, seed, b, n, noise=CLUSTER_NOISE):
    """rows [b n, (b + 1) n) of the seeded stream: a random centre + N(0, noise^2 / dim) per element, L2-normalised; float32 on the centres' device"""
    dev = centres.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed * 100_003 + b)
    idx = torch.randint(0, centres.shape[0], (n,), generator=gen, device=dev)
    x = centres[idx] + torch.randn((n, centres.shape[1]), generator=gen, device=dev, dtype=torch.float32) * (noise / centres.shape[1] ** 0.5)
    return x / x.norm(dim=1, keepdim=True)


def clustered_queries(torch, centres, seed, nq, noise=QUERY_NOISE):
    """nq unit-norm queries near randomly chosen cluster centres (host float32 array)"""
    return clustered_block(torch, centres, seed + 977, 0, nq, noise=noise).cpu().numpy()


def adversarial_block(torch, seed, b, n, dim, device="cuda", spread=1.0e-4):
    """every row within ~1e-3 of ONE unit vector (and so of every query made the same way): no lower bound can separate them"""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed * 7919 + 13)
    base = torch.randn((1, dim), generator=gen, device=device, dtype=torch.float32)
    base = base / base.norm()
    gen.manual_seed(seed * 100_003 + b)
    return base + torch.randn((n, dim), generator=gen, device=device, dtype=torch.float32) * spread
