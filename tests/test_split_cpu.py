"""CPU half of the split-product claim: the decomposition the kernels use is exact, every piece is a bf16 value, and the
six issued terms reproduce a product to 2^-24 (half an fp32 ulp) -- on random data over the whole exponent range and on the delicate values.  The GPU half
(tests/test_split_gpu.py) holds the kernels against float64."""
import numpy as np

from oracle import split_ref


def _samples():
    rng = np.random.default_rng(0)
    a = rng.standard_normal(200000).astype(np.float32) * (10.0 ** rng.uniform(-30, 30, 200000)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, -(2.0 - 2.0 ** -23), 16777215.0, 0.1, 1.0 / 3.0,
                        65504.0, 3.0e38, -3.0e38, 1.0e-30, 2.0 ** -100, 123456.789], np.float32)
    return np.concatenate([a, special])


def test_split_is_exact_and_every_piece_is_a_bf16_value():
    x = _samples()
    h, m, l = split_ref.split3(x)
    for p in (h, m, l):
        assert np.all(p.view(np.uint32) & np.uint32(0xFFFF) == 0), "a piece has bits below bf16's mantissa"
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # magnitudes: |m| <= 2^-8 |x| (half a bf16 ulp, the ulp taken at the binade below), |l| <= 2^-8 |m|
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_terms_reproduce_a_product_to_half_an_fp32_ulp():
    x = _samples()
    rng = np.random.default_rng(1)
    y = rng.permutation(x)
    keep = (np.abs(x.astype(np.float64) * y.astype(np.float64)) < 1e38) & (x != 0) & (y != 0)
    x, y = x[keep], y[keep]
    exact = x.astype(np.float64) * y.astype(np.float64)
    rel = np.abs(split_ref.product6(x, y) - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -24, rel.max()
    # the truncating split (first version of the kernels) for comparison: up to 2^-20
    rel_t = np.abs(split_ref.product6(x, y, split_ref.trunc_bf16) - exact) / np.abs(exact)
    assert 2.0 ** -23 < rel_t.max() <= 2.0 ** -20
    # and what a rounded-operand product (one bf16 piece, as with --operand bf16) gives, for scale: ~2^-8
    h_only = split_ref.rne_bf16(x).astype(np.float64) * split_ref.rne_bf16(y).astype(np.float64)
    assert (np.abs(h_only - exact) / np.abs(exact)).max() > 2.0 ** -9
