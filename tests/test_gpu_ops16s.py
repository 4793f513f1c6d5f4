"""Round 3: the operators on tensors STORED in the 16-bit compute dtype (csrc/conv_glds16.hip, the storage-generic norm / activation
kernels of csrc/elementwise.hip, csrc/st16.h) through the C ABI.

The reference is fp32-only; the contract is the build's own: under a 16-bit compute dtype the wide
activations and their gradients live in HBM in that dtype.  A convolution operand has the same value whether its producer or the
conv loader rounded it, so the exactness statement of tests/test_gpu_ops16.py carries over unchanged: feeding the oracle the SAME
16-bit operands in fp64 reproduces forward / dgrad / wgrad to fp32 summation-order accuracy (2e-4); a 16-bit OUTPUT is that result
rounded once more (one ulp of the dtype).  Border handling is the point of the dgrad cases: out-of-range filter taps are served by
buffer loads beyond the tensor's descriptor (must read zero), the reflection is folded by an ordered gather."""
import ctypes as C

import pytest
import torch

from oracle import aclgan_oracle as O

pytestmark = pytest.mark.gpu

EXACT_TOL = 2e-4
ULP = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}
TDT = {"bf16": torch.bfloat16, "fp16": torch.float16}


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import os
    os.environ.setdefault("ACLGAN_WGRAD16S_MINPIX", "64")      # exercise the pixel-major weight-gradient kernel at test sizes too
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib
    return _lib


@pytest.fixture(params=[1, 4], ids=["tile128", "tilemax"])
def tile(request, L):
    """1 = the default 128-row tiles; 4 = the largest tile that still fills the chip (256 x 128 / 256 x 256 eight-wave workgroups:
    measured slower inside the step, kept as a tested option -- profiles/r03_experiments.md)"""
    old = L.lib.aclgan_set_tuning(b"glds_tile", request.param)
    assert old >= 0
    yield request.param
    L.lib.aclgan_set_tuning(b"glds_tile", old)


# (B, Hi, Wi, Ci, Co, k, s, p, act)
CASES = [
    (2, 64, 64, 256, 256, 3, 1, 1, "none"),     # ResBlock conv: 64 x 2 tiles of 128 x 128, 36 k-tiles over 9 taps
    (3, 20, 28, 64, 128, 4, 2, 1, "relu"),      # CE1 / D2: stride-2 parity classes, Cin 64 (one k-tile per tap), ragged M
    (1, 32, 32, 128, 64, 3, 1, 1, "lrelu"),     # Cout 64: the 128 x 64 tile
    (2, 16, 16, 256, 512, 4, 2, 1, "lrelu"),    # late discriminator conv (small map, K = 4096): forward keeps the split-K kernel, dgrad runs here
    (2, 12, 12, 128, 128, 1, 1, 0, "none"),     # 1x1, no padding
    (1, 9, 7, 64, 64, 3, 1, 1, "none"),         # a single ragged tile
    # grids large enough for the 8-wave tiles (csrc/conv_glds16.hip glds_tile: the largest tile with >= 224 workgroups)
    (7, 64, 64, 64, 256, 3, 1, 1, "relu"),      # forward: 112 x 2 tiles of 256 x 128
    (14, 64, 64, 64, 256, 3, 1, 1, "lrelu"),    # forward: 224 tiles of 256 x 256
    (7, 62, 66, 256, 64, 3, 1, 1, "none"),      # dgrad: N = Cin = 256 -> 256 x 128 tiles over the padded grid, ragged last tile
    (15, 62, 62, 256, 64, 3, 1, 1, "none"),     # dgrad: 256 x 256 tiles
    (8, 64, 64, 128, 256, 4, 2, 1, "none"),     # stride 2 with big tiles on the dgrad side (four parity classes of the padded grid)
    (16, 32, 32, 128, 256, 3, 1, 1, "relu"),    # round 5: the patch-resident forward kernel on a 32-wide map (8 image rows per tile, two channel blocks)
]


def _t(case, seed):
    B, Hi, Wi, Ci, Co, k, s, p, act = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, Hi, Wi, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    b = torch.randn(Co, generator=g) * 0.1
    return x, w, b


def _packs(L, w_ohwi, dt):
    Co, kh, kw, Ci = w_ohwi.shape
    w16 = torch.empty(w_ohwi.numel(), dtype=torch.int16, device="cuda")
    w16t = torch.empty(w_ohwi.numel(), dtype=torch.int16, device="cuda")
    L.check(L.lib.aclgan_pack_weights16(L.ptr(w_ohwi), L.ptr(w16), L.ptr(w16t), Co, kh * kw, Ci, L.DTYPE[dt], L.stream_ptr()), "pack_weights16")
    return w16, w16t


def _r(t, dt):
    return t.to(TDT[dt]).float()


def _patch_on(L):
    prev = C.c_int()
    L.check(L.lib.aclgan_tuning(b"fwd16_patch", 1, C.byref(prev)), "tuning")
    L.check(L.lib.aclgan_tuning(b"fwd16_patch", prev.value, None), "tuning")
    return prev.value != 0


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_cast_storage_round_trip(L, dt):
    x = torch.randn(4 * 1031, generator=torch.Generator().manual_seed(0)).cuda()
    h = torch.empty(x.numel(), dtype=TDT[dt], device="cuda")
    L.check(L.lib.aclgan_cast_storage(L.ptr(x), 0, L.ptr(h), L.DTYPE[dt], x.numel(), L.stream_ptr()), "cast_storage")
    assert torch.equal(h, x.to(TDT[dt]))             # round to nearest even, like torch
    back = torch.empty_like(x)
    L.check(L.lib.aclgan_cast_storage(L.ptr(h), L.DTYPE[dt], L.ptr(back), 0, x.numel(), L.stream_ptr()), "cast_storage")
    assert torch.equal(back, h.float())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd16s(L, case, dt, tile):
    from gpu_util import conv_desc, out_hw, nhwc, nchw, ohwi
    B, Hi, Wi, Ci, Co, k, s, p, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, 0, act)
    if not L.lib.aclgan_conv16s_ok(C.byref(d), 0):
        assert case[3] * case[5] ** 2 > 1024       # only the small-grid / long-K layers are left to the split-K kernel
        pytest.skip("forward of this shape stays on the split-K kernel (conv_fast16.hip, 16-bit x through its A16 path)")
    x, w, b = _t(case, 0)
    wg, bg = ohwi(w).cuda(), b.cuda()
    x16 = nhwc(x).cuda().to(TDT[dt])
    w16, _ = _packs(L, wg, dt)
    Ho, Wo = out_hw(Hi, Wi, k, s, p, 0)
    exact = O.conv_block(_r(x, dt).double(), _r(w, dt).double(), b.double(), s, p, act)
    code = L.DTYPE[dt]
    y32 = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
    L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y32), 0, L.stream_ptr()), "conv2d_fwd16s")
    assert _rel(nchw(y32), exact) < EXACT_TOL
    y16 = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda").to(TDT[dt])
    L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y16), code, L.stream_ptr()), "conv2d_fwd16s")
    assert torch.equal(y16, y32.to(TDT[dt]))           # the 16-bit output IS the fp32 output rounded once
    # same bits from the register-staged kernel reading the same 16-bit x (its A16 path), and run to run
    y_old = torch.empty_like(y32)
    scr = torch.empty(L.lib.aclgan_conv2d_fwd16_scratch_bytes(C.byref(d)) // 4 + 64, device="cuda")
    L.check(L.lib.aclgan_conv2d_fwd16_x16(C.byref(d), code, L.ptr(x16), L.ptr(wg), L.ptr(w16), L.ptr(bg), L.ptr(y_old), L.ptr(scr), L.stream_ptr()), "fwd16_x16")
    assert _rel(y32, y_old) < 1e-5
    y2 = torch.empty_like(y32)
    L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y2), 0, L.stream_ptr()))
    assert torch.equal(y32, y2)


PATCH_CASES = [CASES[0], CASES[6], CASES[7], CASES[11], (4, 64, 64, 128, 128, 3, 1, 1, "lrelu")]


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", PATCH_CASES)
def test_conv_fwd16p_patch_kernel(L, case, dt):
    """conv_fwd16p_kernel (round 5, csrc/conv_glds16.hip): the 3x3 stride-1 reflect-pad-1 layers with the input patch of a 256-pixel tile
    resident in LDS for all nine taps.  Taken by default where the shape fits (these cases: 64- and 32-wide maps, one / two / four channel
    blocks, both image borders inside one tile); against the exact oracle on the rounded operands like every 16-bit kernel, against
    conv_fwd16s (tuning fwd16_patch = 0: same products, another summation order), 16-bit output == fp32 output rounded once, run to run bits,
    and the launch really is the patch kernel (one launch either way, so the check is on the statistics chunk it reports: 256 rows)."""
    from gpu_util import conv_desc, nhwc, nchw, ohwi
    B, Hi, Wi, Ci, Co, k, s, p, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, 0, act)
    x, w, b = _t(case, 0)
    wg, bg = ohwi(w).cuda(), b.cuda()
    x16 = nhwc(x).cuda().to(TDT[dt])
    w16, _ = _packs(L, wg, dt)
    exact = O.conv_block(_r(x, dt).double(), _r(w, dt).double(), b.double(), s, p, act)
    code = L.DTYPE[dt]
    out = {}
    prev = C.c_int()
    L.check(L.lib.aclgan_tuning(b"fwd16_patch", 1, C.byref(prev)), "tuning")
    try:
        for mode in (1, 2, 0):      # 1: all waves in lockstep (default), 2: the two waves of a SIMD in counter-phase, 0: conv_fwd16s
            L.check(L.lib.aclgan_tuning(b"fwd16_patch", mode, None), "tuning")
            chunk = L.lib.aclgan_conv2d_fwd16s_stats_chunk(C.byref(d))
            y32 = torch.full((B, Hi, Wi, Co), float("nan"), device="cuda")
            L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y32), 0, L.stream_ptr()), "conv2d_fwd16s")
            y16 = torch.full((B, Hi, Wi, Co), float("nan"), device="cuda").to(TDT[dt])
            L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y16), code, L.stream_ptr()), "conv2d_fwd16s")
            y2 = torch.empty_like(y32)
            L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y2), 0, L.stream_ptr()))
            out[mode] = (y32, y16, y2, chunk)
    finally:
        L.check(L.lib.aclgan_tuning(b"fwd16_patch", prev.value, None), "tuning")
    assert out[1][3] == 256 and out[2][3] == 256, "the patch kernel was not taken for %r (statistics chunk %d)" % (case, out[1][3])
    for mode in (1, 2, 0):
        y32, y16, y2, _ = out[mode]
        assert _rel(nchw(y32), exact) < EXACT_TOL, (mode, _rel(nchw(y32), exact))
        assert torch.equal(y16, y32.to(TDT[dt]))
        assert torch.equal(y32, y2)
    assert torch.equal(out[1][0], out[2][0])      # the two schedules of the patch kernel add the same products in the same order
    assert _rel(out[1][0], out[0][0]) < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("yst", [0, 1])
@pytest.mark.parametrize("case", [CASES[0], (4, 32, 32, 64, 64, 3, 1, 1, "none"), CASES[6], CASES[7], (16, 32, 32, 64, 128, 3, 1, 1, "relu")])
def test_conv_fwd16s_epilogue_statistics(L, case, dt, yst, tile):
    """The (mean, M2) chunk partials the forward launch emits for the normalisation layer == the statistics of the outputs AS STORED
    (rounded when y is 16-bit), chunk = the launch's row tile (128 / 256 rows)."""
    from gpu_util import conv_desc, out_hw, nhwc, ohwi
    B, Hi, Wi, Ci, Co, k, s, p, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, 0, act)
    R = L.lib.aclgan_conv2d_fwd16s_stats_chunk(C.byref(d))
    assert R in (128, 256), "every case here is a shape the step fuses the statistics for"
    if case in (CASES[6], CASES[7]): assert R == (256 if (tile == 4 or _patch_on(L)) else 128)      # (3x3 on a 64-wide map: the patch kernel's 256-row tiles when it is on)
    x, w, b = _t(case, 2)
    wg, bg = ohwi(w).cuda(), b.cuda()
    x16 = nhwc(x).cuda().to(TDT[dt])
    w16, _ = _packs(L, wg, dt)
    Ho, Wo = out_hw(Hi, Wi, k, s, p, 0)
    code = L.DTYPE[dt]
    st = code if yst else 0
    M = B * Ho * Wo
    y = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda").to(TDT[dt] if yst else torch.float32)
    stats = torch.full((M // R, Co, 2), float("nan"), device="cuda")
    L.check(L.lib.aclgan_conv2d_fwd16s_stats(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y), st, L.ptr(stats), L.stream_ptr()), "fwd16s_stats")
    y_plain = torch.empty_like(y)
    L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y_plain), st, L.stream_ptr()), "fwd16s")
    assert torch.equal(y, y_plain)                       # the statistics do not change what is stored
    yc = y.double().reshape(M // R, R, Co)
    mean = yc.mean(1)
    m2 = ((yc - mean[:, None]) ** 2).sum(1)
    sd = yc.std().item()
    assert (stats[..., 0].double() - mean).abs().max().item() < 1e-5 * max(sd, 1e-3)
    assert ((stats[..., 1].double() - m2).abs() / m2.clamp_min(1e-6 * R * sd * sd)).max().item() < 1e-4
    s2 = torch.empty_like(stats)
    L.check(L.lib.aclgan_conv2d_fwd16s_stats(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y), st, L.ptr(s2), L.stream_ptr()))
    assert torch.equal(stats, s2)


@pytest.fixture(params=[0, 1], ids=["fold", "direct"])
def direct(request, L):
    """1: pixels without mirrored partners are written by the GEMM epilogue straight into a 16-bit dx (csrc/conv_glds16.hip store_acc_dx),
    the fold touches the border band only -- the same bits as the full fold (default), which the assertions below pin"""
    old = L.lib.aclgan_set_tuning(b"dgrad16s_direct", request.param)
    assert old >= 0
    yield request.param
    L.lib.aclgan_set_tuning(b"dgrad16s_direct", old)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES)
def test_conv_dgrad16s(L, case, dt, tile, direct):
    from gpu_util import conv_desc, nhwc, nchw, ohwi
    B, Hi, Wi, Ci, Co, k, s, p, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, 0, "none")
    assert L.lib.aclgan_conv16s_ok(C.byref(d), 1) == 1
    x, w, b = _t(case, 1)
    xr = x.double().requires_grad_(True)
    y = O.conv_block(xr, _r(w, dt).double(), b.double(), s, p, "none")
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(_r(dy, dt).double())
    code = L.DTYPE[dt]
    wg = ohwi(w).cuda()
    dy16 = nhwc(dy).cuda().to(TDT[dt])
    _, w16t = _packs(L, wg, dt)
    scr = torch.empty(L.lib.aclgan_conv2d_dgrad16s_scratch_bytes(C.byref(d)) // 4 + 64, device="cuda")
    dx32 = torch.full((B, Hi, Wi, Ci), float("nan"), device="cuda")
    L.check(L.lib.aclgan_conv2d_dgrad16s(C.byref(d), code, L.ptr(dy16), L.ptr(w16t), L.ptr(dx32), 0, 0, L.ptr(scr), L.stream_ptr()), "conv2d_dgrad16s")
    # the padded-grid partial sums are stored in the 16-bit dtype before the fold: exact to one rounding of the dtype
    assert _rel(nchw(dx32), xr.grad) < 2 * ULP[dt]
    dx16 = torch.full((B, Hi, Wi, Ci), float("nan"), device="cuda").to(TDT[dt])
    L.check(L.lib.aclgan_conv2d_dgrad16s(C.byref(d), code, L.ptr(dy16), L.ptr(w16t), L.ptr(dx16), code, 0, L.ptr(scr), L.stream_ptr()), "conv2d_dgrad16s")
    assert torch.equal(dx16, dx32.to(TDT[dt]))
    # accumulate into a 16-bit dx: read, add in fp32, round once
    base = torch.randn(B, Hi, Wi, Ci, generator=torch.Generator().manual_seed(6)).cuda().to(TDT[dt])
    acc = base.clone()
    L.check(L.lib.aclgan_conv2d_dgrad16s(C.byref(d), code, L.ptr(dy16), L.ptr(w16t), L.ptr(acc), code, 1, L.ptr(scr), L.stream_ptr()))
    assert torch.equal(acc, (base.float() + dx32).to(TDT[dt]))
    # no atomics: bit-reproducible
    dx2 = torch.empty_like(dx32)
    L.check(L.lib.aclgan_conv2d_dgrad16s(C.byref(d), code, L.ptr(dy16), L.ptr(w16t), L.ptr(dx2), 0, 0, L.ptr(scr), L.stream_ptr()))
    assert torch.equal(dx32, dx2)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES[:4] + [(2, 8, 8, 256, 128, 5, 1, 2, "up"), (1, 16, 16, 128, 64, 5, 1, 2, "up")])
@pytest.mark.parametrize("stx,stdy", [(1, 1), (1, 0), (0, 1)])
def test_conv_wgrad16_any_storage(L, case, dt, stx, stdy):
    """the ordered-slice weight-gradient kernel with either operand stored in the 16-bit dtype (sub-pixel layers: x 16-bit, dy fp32)"""
    from gpu_util import conv_desc, nhwc, ohwi
    B, Hi, Wi, Ci, Co, k, s, p, act = case
    up = 1 if act == "up" else 0
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, "none")
    x, w, b = _t(case, 2)
    wr = w.double().requires_grad_(True); br = b.double().requires_grad_(True)
    y = O.conv_block(_r(x, dt).double(), wr, br, s, p, "none", upsample=bool(up))
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
    y.backward(_r(dy, dt).double())
    code = L.DTYPE[dt]
    xg = nhwc(x).cuda(); dyg = nhwc(dy).cuda()
    xs = xg.to(TDT[dt]) if stx else xg
    dys = dyg.to(TDT[dt]) if stdy else dyg
    dw = torch.zeros(Co, k, k, Ci, device="cuda"); db = torch.zeros(Co, device="cuda")
    scr = torch.empty(L.lib.aclgan_conv2d_wgrad16_scratch_bytes(C.byref(d)) // 4 + 64, device="cuda")
    L.check(L.lib.aclgan_conv2d_wgrad16_st(C.byref(d), code, L.ptr(xs), code if stx else 0, L.ptr(dys), code if stdy else 0, L.ptr(dw), L.ptr(db),
                                           L.ptr(scr), L.stream_ptr()), "conv2d_wgrad16_st")
    assert _rel(dw, ohwi(wr.grad)) < EXACT_TOL
    # bias gradient: summed from dy as stored (fp32 dy: unrounded; 16-bit dy: the rounded values, which is what br.grad saw)
    want_db = br.grad if stdy else dy.double().sum(dim=(0, 2, 3))
    assert _rel(db, want_db) < EXACT_TOL
    # the same numbers from the fp32-storage call on pre-rounded operands (bitwise when the same kernel ran; with both operands 16-bit and
    # Cin, Cout multiples of 128 the pixel-major LDS-DMA kernel runs instead: other summation order)
    dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
    xr32, dyr32 = xg.to(TDT[dt]).float(), (dyg.to(TDT[dt]).float() if stdy else dyg)
    L.check(L.lib.aclgan_conv2d_wgrad16_st(C.byref(d), code, L.ptr(xr32), 0, L.ptr(dyr32), 0, L.ptr(dw2), L.ptr(db2), L.ptr(scr), L.stream_ptr()))
    if stx and stdy and Ci % 128 == 0 and Co % 128 == 0 and not up:
        assert _rel(dw, dw2) < 1e-5          # (bitwise below the pixel-count threshold of the LDS-DMA kernel, where the same kernel ran)
    else:
        assert torch.equal(dw, dw2)
    # reproducible bit for bit either way (ordered slices)
    dw3 = torch.zeros_like(dw); db3 = torch.zeros_like(db)
    L.check(L.lib.aclgan_conv2d_wgrad16_st(C.byref(d), code, L.ptr(xs), code if stx else 0, L.ptr(dys), code if stdy else 0, L.ptr(dw3), L.ptr(db3),
                                           L.ptr(scr), L.stream_ptr()))
    assert torch.equal(dw, dw3) and torch.equal(db, db3)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("kind,act,res", [("in", "relu", False), ("adain", "none", True), ("ln", "relu", False)])
def test_norm_on_16bit_storage(L, dt, kind, act, res):
    """norm_fwd / norm_bwd with x, y, residual, dy, dx, dres stored in the 16-bit dtype == the fp32-storage kernels on the same (rounded)
    values, outputs rounded once.  Statistics and parameter gradients are fp32 in both."""
    B, H, W, Cn = 3, 12, 20, 64
    HW = H * W
    code = L.DTYPE[dt]
    g = torch.Generator().manual_seed(3)
    x = _r(torch.randn(B, HW, Cn, generator=g) * 2 + 0.5, dt).cuda()
    r = _r(torch.randn(B, HW, Cn, generator=g), dt).cuda() if res else None
    dy = _r(torch.randn(B, HW, Cn, generator=g), dt).cuda()
    K = {"in": L.NORM["in"], "adain": L.NORM["adain"], "ln": L.NORM["ln"]}[kind]
    if kind == "adain":
        w = torch.randn(B, Cn, generator=g).cuda(); b = torch.randn(B, Cn, generator=g).cuda(); ws = Cn
    elif kind == "ln":
        w = torch.rand(Cn, generator=g).cuda(); b = torch.randn(Cn, generator=g).cuda(); ws = 0
    else:
        w = b = None; ws = 0
    nst = Cn * B if kind != "ln" else B
    scr = torch.empty(L.lib.aclgan_norm_scratch_bytes(B, HW, Cn) // 4 + 64, device="cuda")

    def run(st16):
        T = TDT[dt] if st16 else torch.float32
        s = code if st16 else 0
        xs, rs, dys = x.to(T), (r.to(T) if res else None), dy.to(T)
        y = torch.empty(B, HW, Cn, device="cuda", dtype=T)
        mean = torch.empty(nst, device="cuda"); rstd = torch.empty(nst, device="cuda")
        stf = (C.c_int * 3)(s, s, s)
        L.check(L.lib.aclgan_norm_fwd_st(K, L.ACT[act], B, HW, Cn, L.ptr(xs), L.ptr(w), L.ptr(b), ws, L.ptr(rs), L.ptr(y), L.ptr(mean), L.ptr(rstd),
                                         L.ptr(scr), stf, L.stream_ptr()), "norm_fwd_st")
        dx = torch.empty(B, HW, Cn, device="cuda", dtype=T)
        dres = torch.empty(B, HW, Cn, device="cuda", dtype=T) if res else None
        dw = torch.zeros_like(w) if w is not None else None
        dbb = torch.zeros_like(b) if b is not None else None
        stb = (C.c_int * 6)(s, s, s, s, s, 0)
        L.check(L.lib.aclgan_norm_bwd_st(K, L.ACT[act], B, HW, Cn, L.ptr(xs), L.ptr(y), L.ptr(dys), L.ptr(w), ws, L.ptr(mean), L.ptr(rstd), L.ptr(dx),
                                         L.ptr(dw), L.ptr(dbb), L.ptr(dres), 0, L.ptr(scr), stb, L.stream_ptr()), "norm_bwd_st")
        return y, mean, rstd, dx, dres, dw, dbb
    y32, m32, r32, dx32, dr32, dw32, db32 = run(False)
    y16, m16, r16, dx16, dr16, dw16, db16 = run(True)
    assert torch.equal(m16, m32) and torch.equal(r16, r32)          # statistics: same inputs, same fp32 arithmetic
    assert torch.equal(y16, y32.to(TDT[dt]))
    # the backward of the 16-bit run reads the ROUNDED y for the activation mask -- same sign, so the same gradient, rounded once
    assert _rel(dx16.float(), dx32) < 2 * ULP[dt]
    if res:
        assert torch.equal(dr16, dr32.to(TDT[dt]))
    if dw32 is not None:
        assert _rel(dw16, dw32) < 1e-5 and _rel(db16, db32) < 1e-5
