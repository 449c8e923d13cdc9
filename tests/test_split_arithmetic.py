"""The arithmetic claim behind the split GEMM (csrc/conv_wino.hip, wino_gemm_split*_kernel), checked on the host with
numpy: an fp32 value x is EXACTLY h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest
even, residuals computed in fp32), and the six partial products the kernel keeps reproduce a*b to within one fp32
rounding.  No GPU and no oracle involved: this pins the number format reasoning, the kernels themselves are tested
against fp64 in test_gpu_ops.py::test_wino_gemm_matrix_pipes."""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    q = (r - m).astype(np.float32)
    return h, m, bf16_rne(q), r, q


def samples(n, seed, lo=40, hi=215):
    g = np.random.default_rng(seed)
    mant = g.integers(0, 1 << 23, n, dtype=np.uint32)
    expo = g.integers(lo, hi, n, dtype=np.uint32)          # default 2^-87 .. 2^87: residuals stay normal
    sign = g.integers(0, 2, n, dtype=np.uint32)
    x = ((sign << 31) | (expo << 23) | mant).view(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 0.1, 1e-3, 3.0, 255.0, 256.0, 257.0,
                     1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 - 2.0 ** -9, float.fromhex('0x1.fffffep0'), float.fromhex('0x1.ff7fffp0'),
                     float.fromhex('0x1.008001p0')], np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_parts_reproduce_fp32_exactly():
    x = samples(1_000_000, 1)
    h, m, l, r, q = split3(x)
    # the residuals are exact in fp32 ...
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(q.astype(np.float64), r.astype(np.float64) - m.astype(np.float64))
    # ... the last one is representable in bf16 ...
    assert np.array_equal(l, q)
    # ... so the three parts are the value
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # part sizes (round to nearest): |m| <= 2^-9 |x|, |l| <= 2^-18 |x| up to the bf16 ulp granularity
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_partial_products_are_one_rounding_from_the_product():
    a = samples(400_000, 2, 90, 165)                        # products and their parts stay inside the fp32 range
    b = samples(400_000, 3, 90, 165)[::-1].copy()
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    f = np.float64
    kept = (ah.astype(f) * bh + ah.astype(f) * bm + am.astype(f) * bh + am.astype(f) * bm + ah.astype(f) * bl + al.astype(f) * bh)
    exact = a.astype(f) * b.astype(f)
    nz = exact != 0
    rel = np.abs(kept[nz] - exact[nz]) / np.abs(exact[nz])
    # dropped: am*bl + al*bm + al*bl <= (2^-9 2^-17 * 2 + 2^-34) |ab| ~ 2^-25 |ab|; allow the bf16-granularity slack
    assert rel.max() <= 2.0 ** -23, rel.max()
    # each kept product of two bf16 numbers has <= 16 significant bits: exact in an fp32 accumulator
    for p, qv in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)):
        prod = p.astype(f) * qv.astype(f)
        assert np.array_equal(prod.astype(np.float32).astype(f), prod)
