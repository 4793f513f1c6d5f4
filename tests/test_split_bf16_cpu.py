"""The arithmetic behind csrc/gemm_bf16x3.hip, restated in numpy (CPU, no GPU): an fp32 number is the sum of three bf16 numbers to
2^-24 of its magnitude, and six of the nine bf16 x bf16 cross products -- each exact in an fp32 accumulator -- reproduce an fp32 dot
product to fp32 accuracy.  The GPU test (tests/test_gpu_ops_misc.py::test_gemm_slices_x3_is_fp32_accurate) measures the kernel; this
one pins the claim itself, including where it stops holding (three products = a bf16x3 'fast' mode would NOT be fp32)."""
import numpy as np


def bf16_rne(x):
    """round fp32 -> bf16 (round to nearest even, as v_cvt_pk_bf16_f32 does), returned as fp32"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    l = bf16_rne(r2)
    return h, m, l, r1, r2


def _values(n, seed):
    g = np.random.default_rng(seed)
    x = g.standard_normal(n).astype(np.float32) * (10.0 ** g.integers(-12, 13, n)).astype(np.float32)
    x[:8] = [0.0, 1.0, -1.0, 3.1415927, 1.0000001, 65504.0, 1e-30, -7.0e20]
    return x


def test_three_bf16_numbers_are_an_fp32_number():
    x = _values(200000, 0)
    h, m, l, r1, r2 = split3(x)
    # both differences are exact in fp32 (checked in float64) ...
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    # ... each part has at most 8 significant bits, and the sum is x to 2^-24 (in fact to 2^-25: three 8-bit parts + signs)
    for part in (h, m, l):
        assert np.array_equal(bf16_rne(part), part)
    err = np.abs(x.astype(np.float64) - (h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)))
    assert np.all(err <= np.abs(x.astype(np.float64)) * 2.0 ** -24)
    # the parts shrink by 2^-8 each: |m| <= 2^-8 |h| (half an ulp of bf16 is 2^-9 relative), |l| <= 2^-8 |m|
    nz = h != 0
    assert np.all(np.abs(m[nz]) <= np.abs(h[nz]) * 2.0 ** -8)
    nzm = m != 0
    assert np.all(np.abs(l[nzm]) <= np.abs(m[nzm]) * 2.0 ** -8)


def _dot_terms(a, b):
    """the nine plane products of a dot product, each accumulated in float64 (the MFMA adds bf16 x bf16 products exactly into fp32;
    float64 here isolates the effect of DROPPING terms from the rounding of the accumulation)"""
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    A = [p.astype(np.float64) for p in (ah, am, al)]
    B = [p.astype(np.float64) for p in (bh, bm, bl)]
    return {(i, j): float(np.dot(A[i], B[j])) for i in range(3) for j in range(3)}


def test_six_of_nine_products_are_fp32_accurate_and_three_are_not():
    g = np.random.default_rng(1)
    worst6, worst3 = 0.0, 0.0
    for K in (32, 256, 2304):
        for _ in range(40):
            a = g.standard_normal(K).astype(np.float32)
            b = g.standard_normal(K).astype(np.float32)
            exact = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
            mag = float(np.dot(np.abs(a).astype(np.float64), np.abs(b).astype(np.float64)))
            t = _dot_terms(a, b)
            six = t[0, 0] + t[0, 1] + t[1, 0] + t[0, 2] + t[1, 1] + t[2, 0]       # the kernel's products
            three = t[0, 0] + t[0, 1] + t[1, 0]                                     # a "bf16x3 fast" mode would stop here
            worst6 = max(worst6, abs(six - exact) / mag)
            worst3 = max(worst3, abs(three - exact) / mag)
    # dropped: m l', l m', l l' -- each below 2^-24 of |a||b| elementwise, random in sign
    assert worst6 < 2.0 ** -24, worst6              # below ONE fp32 rounding of the result's magnitude (measured ~1e-9)
    assert worst3 > 50 * worst6                     # three products lose the 2^-16 terms: not fp32 (measured ~1e-6)
    assert worst3 < 2.0 ** -14


def test_bf16_rounding_matches_torch():
    import torch
    x = _values(50000, 2)
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(bf16_rne(x), ref)
