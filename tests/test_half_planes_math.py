"""CPU (numpy): the arithmetic the "half" GEMM mode and the fp16-plane recurrences rest on (eesen_amd/csrc/gemm.hip):
an fp32 value held as TWO fp16 planes -- hi = fp16(a 2^s), lo = fp16(a 2^s - hi), round to nearest at both levels -- and a product formed
as hi*hi' + hi*lo' + lo*hi'.  What is asserted here is what the kernels' comments claim: the representation is good to 2^-22 in the worst
case (2^-23 is the largest seen), the power-of-two scale (half_scale, restated below bit for bit) keeps every plane inside fp16's range and is exact, and
against an fp64 product the three-product form is as close as an fp32 GEMM.  (The device side -- the same numbers through
v_mfma_f32_32x32x16_f16, fp16 denormals multiplied exactly -- is tests/test_gpu_gemm.py.)"""
import numpy as np


def half_scale(amax):
    """gemm.hip / lstm_persistent.hip: half_scale -- the power of two that brings a bound into [2^14, 2^15), and its inverse, from the bits."""
    e = (np.float32(amax).view(np.uint32) >> np.uint32(23)) & np.uint32(0xFF)
    s = int(min(max(127 + 14 + 127 - int(e), 1), 253))
    return np.uint32(s << 23).view(np.float32), np.uint32((254 - s) << 23).view(np.float32)


def planes(a, scale):
    ap = (a.astype(np.float32) * np.float32(scale)).astype(np.float32)          # exact: a power of two
    hi = ap.astype(np.float16)
    r = (ap - hi.astype(np.float32)).astype(np.float32)
    assert np.array_equal(r.astype(np.float64), ap.astype(np.float64) - hi.astype(np.float64))   # the residual is exact in fp32
    lo = r.astype(np.float16)
    return hi, lo


def test_scale_is_an_exact_power_of_two_that_fits_fp16():
    for amax in (1.0, 1.9999, 0.3, 7e-4, 123.0, 6.5e4, 3e38, 1e-30, 1e-38):
        sc, inv = half_scale(amax)
        m, ex = np.frexp(np.float64(sc))
        assert m == 0.5 and np.float64(sc) * np.float64(inv) == 1.0            # 2^k and 2^-k
        if 1e-33 < amax < 1e33:                                                 # (beyond: the scale saturates at 2^+-126, the planes underflow / the product overflows like fp32 would)
            assert 2.0 ** 14 <= np.float64(amax) * np.float64(sc) < 2.0 ** 15
    sc, inv = half_scale(0.0)                                                   # an all-zero operand: any finite scale will do, the inverse stays a normal number
    assert np.isfinite(sc) and inv > 0
    sc, inv = half_scale(np.inf)                                                # Inf / NaN bounds give finite scales: the planes carry the Inf / NaN through, as an fp32 GEMM would
    assert np.isfinite(sc) and np.isfinite(inv)


def test_two_planes_hold_an_fp32_value_to_22_bits():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 0, 200000))).astype(np.float32)
    a[0] = 1.0                                                                  # the tensor's largest sets the scale
    a = np.clip(a, -1.0, 1.0)
    sc, inv = half_scale(np.abs(a).max())
    hi, lo = planes(a, sc)
    assert np.all(np.isfinite(hi.astype(np.float32))) and np.abs(hi.astype(np.float32)).max() < 2.0 ** 15 + 16
    back = (hi.astype(np.float64) + lo.astype(np.float64)) * np.float64(inv)
    err = np.abs(back - a.astype(np.float64))
    # |a - (hi + lo)| <= max(2^-22 |a|, 2^-25 / scale): two roundings to nearest of 11 bits each where lo is a normal number (both at half
    # an ulp is the worst case; over these 2 x 10^5 values the largest is ~2^-23), the denormal grid (2^-24 apart) where the residual
    # falls below 2^-14 -- i.e. 22-23 bits for everything within 2^-15 of the largest element, and never worse than 2^-39 of the largest below
    big = np.abs(a).astype(np.float64) * np.float64(sc) >= 0.5
    assert np.max(err[big] / np.abs(a[big])) <= 2.0 ** -22
    assert np.max(err[big] / np.abs(a[big])) <= 2.0 ** -23 * 1.1                # (measured; not a bound)
    floor = 2.0 ** -25 * np.float64(inv)
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(a.astype(np.float64)), floor * 1.0001)) and floor <= 2.0 ** -39


def _three_products(A, B):
    want = A.astype(np.float64) @ B.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    # per row of A / column of B, as the kernel scales them
    sa = np.array([half_scale(np.abs(A[i]).max()) for i in range(A.shape[0])])
    sb = np.array([half_scale(np.abs(B[:, j]).max()) for j in range(B.shape[1])])
    ah, al = planes(A, sa[:, :1]); bh, bl = planes(B, sb[:, 0][None, :])
    ah, al, bh, bl = (x.astype(np.float64) for x in (ah, al, bh, bl))
    got = (ah @ bh + ah @ bl + al @ bh) * sa[:, 1:].astype(np.float64) * sb[:, 1][None, :].astype(np.float64)
    return np.max(np.abs(got - want) / den), np.max(np.abs((A @ B).astype(np.float64) - want) / den)


def test_three_products_are_as_close_to_fp64_as_an_fp32_gemm():
    rng = np.random.default_rng(1)
    for K in (16, 64, 1024):
        # operands spanning six decades INSIDE every dot product (the GPU test's distribution): as close to fp64 as an fp32 GEMM
        A = (rng.standard_normal((96, K)) * np.exp(rng.uniform(-7, 7, (96, K)))).astype(np.float32)
        B = (rng.standard_normal((K, 80)) * np.exp(rng.uniform(-7, 7, (K, 80)))).astype(np.float32)
        e_planes, e_f32 = _three_products(A, B)
        assert e_planes < max(1.5 * e_f32, 4e-7), (K, e_planes, e_f32)
        # operands within 2^-15 of their row's / column's largest: the bound of the comments, 3 x 2^-22 per product relative to sum |a||b|
        # (measured: a tenth of it -- the representation errors are random in sign)
        A = (rng.standard_normal((96, K)) * np.exp(rng.uniform(-2, 2, (96, K)))).astype(np.float32)
        B = (rng.standard_normal((K, 80)) * np.exp(rng.uniform(-2, 2, (K, 80)))).astype(np.float32)
        e_planes, e_f32 = _three_products(A, B)
        assert e_planes < 3 * 2.0 ** -22 and e_planes < 4e-7, (K, e_planes)
