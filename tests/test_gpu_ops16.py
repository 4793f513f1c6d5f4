"""The 16-bit MFMA convolution kernels (csrc/conv_fast16.hip: v_mfma_f32_32x32x16_bf16 / _f16) through the C ABI.

The reference has no reduced-precision path (fp32 only), so the contract is stated here and tested in two layers:

  1. INDEXING / ARITHMETIC, exact: the kernels round both MFMA operands to the 16-bit type (round to nearest even) and
     accumulate exact products in fp32.  Feeding the oracle the SAME rounded operands (x.to(dtype).float(), ...) in
     fp64 must therefore reproduce the kernel to fp32 summation-order accuracy (2e-4), for forward, dgrad and wgrad,
     including reflect padding, stride-2 parity classes, split-K, the sub-pixel path of the upsample+5x5 layers and
     ragged tiles.  (The sub-pixel path rounds the MERGED phase filters: its exactness case uses weights on a 1/8 grid,
     whose partial sums are exactly representable.)
  2. PRECISION vs the fp32 oracle on unrounded operands: bf16 <= 2e-2, fp16 <= 3e-3 (max-abs relative).
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle import aclgan_oracle as O

pytestmark = pytest.mark.gpu

EXACT_TOL = 2e-4
PREC_TOL = {"bf16": 2e-2, "fp16": 3e-3}
TDT = {"bf16": torch.bfloat16, "fp16": torch.float16}


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib
    return _lib


# (B, Hi, Wi, Ci, Co, k, s, p, up, act)
CASES16 = [
    (2, 16, 16, 64, 128, 4, 2, 1, 0, "none"),    # CE1 / SE1 / D*.1: stride-2 parity classes, Cin 64 (128 x 64 wgrad tile)
    (1, 16, 16, 128, 256, 4, 2, 1, 0, "relu"),   # CE2 / SE2
    (2, 8, 8, 256, 256, 3, 1, 1, 0, "none"),     # ResBlock conv
    (3, 12, 20, 256, 256, 3, 1, 1, 0, "relu"),   # ragged M tiles
    (2, 8, 8, 256, 128, 5, 1, 2, 1, "none"),     # DU0: sub-pixel path (4 phases + ring)
    (1, 16, 16, 128, 64, 5, 1, 2, 1, "none"),    # DU1: Cout 64 (256 x 64 forward tile, 64 x 128 wgrad tile)
    (2, 9, 13, 64, 64, 5, 1, 2, 1, "relu"),      # upsample+5x5 on a ragged map
    (2, 8, 8, 256, 512, 4, 2, 1, 0, "lrelu"),    # D last strided conv: small grid -> split-K with ordered partials
    (2, 4, 4, 256, 256, 4, 2, 1, 0, "relu"),     # SE4: 2x2 output
    (1, 10, 6, 32, 32, 3, 1, 1, 0, "none"),      # smallest channel counts the forward / dgrad kernels take
    (2, 16, 16, 64, 64, 3, 1, 1, 0, "none"),     # 64 x 64 wgrad tile
]


def _rounded(t, dt):
    return t.to(TDT[dt]).float()


def _tensors(case, seed, grid=False):
    B, Hi, Wi, Ci, Co, k, s, p, up, act = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, Hi, Wi, generator=g)
    if grid:   # weights on a 1/8 grid in [-3/8, 3/8]: sums of up to 4 of them stay exactly representable in bf16 / fp16
        w = torch.randint(-3, 4, (Co, Ci, k, k), generator=g).float() / 8.0
    else:
        w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    b = torch.randn(Co, generator=g) * 0.1
    return x, w, b


def _packs(L, w_ohwi, dt):
    Co, kh, kw, Ci = w_ohwi.shape
    w16 = torch.empty(w_ohwi.numel(), dtype=torch.int16, device="cuda")
    w16t = torch.empty(w_ohwi.numel(), dtype=torch.int16, device="cuda")
    L.check(L.lib.aclgan_pack_weights16(L.ptr(w_ohwi), L.ptr(w16), L.ptr(w16t), Co, kh * kw, Ci, L.DTYPE[dt], L.stream_ptr()), "pack_weights16")
    return w16, w16t


def _scratch(nbytes):
    return torch.empty(nbytes // 4 + 64, device="cuda") if nbytes else None


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_pack_weights16(L, dt):
    w = torch.randn(40, 3, 3, 36, generator=torch.Generator().manual_seed(0)).cuda()   # ragged 32x32 transpose tiles
    w16, w16t = _packs(L, w, dt)
    want = w.to(TDT[dt])
    assert torch.equal(w16.view(TDT[dt]).view(40, 9, 36), want.view(40, 9, 36))
    assert torch.equal(w16t.view(TDT[dt]).view(9, 36, 40), want.view(40, 9, 36).permute(1, 2, 0).contiguous())


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES16)
def test_conv_fwd16(L, case, dt):
    from gpu_util import conv_desc, out_hw, nhwc, nchw, ohwi, rel_err
    B, Hi, Wi, Ci, Co, k, s, p, up, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, act)
    assert L.lib.aclgan_conv16_eligible(C.byref(d), 0) == 1
    for grid in (False, True) if up else (False,):
        x, w, b = _tensors(case, 0, grid)
        wg, xg, bg = ohwi(w).cuda(), nhwc(x).cuda(), b.cuda()     # named: a temporary would be freed (and reused) before the launch
        w16, _ = _packs(L, wg, dt)
        Ho, Wo = out_hw(Hi, Wi, k, s, p, up)
        y = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
        scr = _scratch(L.lib.aclgan_conv2d_fwd16_scratch_bytes(C.byref(d)))
        L.check(L.lib.aclgan_conv2d_fwd16(C.byref(d), L.DTYPE[dt], L.ptr(xg), L.ptr(wg), L.ptr(w16), L.ptr(bg), L.ptr(y),
                                          L.ptr(scr), L.stream_ptr()), "conv2d_fwd16")
        exact = O.conv_block(_rounded(x, dt).double(), _rounded(w, dt).double(), b.double(), s, p, act, upsample=bool(up))
        full = O.conv_block(x, w, b, s, p, act, upsample=bool(up))
        if not up or grid:
            assert rel_err(nchw(y), exact) < EXACT_TOL, ("exact", grid)
        assert rel_err(nchw(y), full) < PREC_TOL[dt]
        # reproducible bit for bit (split-K layers reduce ordered partials)
        y2 = torch.empty_like(y)
        L.check(L.lib.aclgan_conv2d_fwd16(C.byref(d), L.DTYPE[dt], L.ptr(xg), L.ptr(wg), L.ptr(w16), L.ptr(bg), L.ptr(y2),
                                          L.ptr(scr), L.stream_ptr()))
        assert torch.equal(y, y2)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES16)
def test_conv_dgrad16(L, case, dt):
    from gpu_util import conv_desc, nhwc, nchw, ohwi, rel_err
    B, Hi, Wi, Ci, Co, k, s, p, up, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, "none")
    assert L.lib.aclgan_conv16_eligible(C.byref(d), 1) == 1
    for grid in (False, True) if up else (False,):
        x, w, b = _tensors(case, 1, grid)
        xr = x.double().requires_grad_(True)
        y = O.conv_block(xr, _rounded(w, dt).double(), b.double(), s, p, "none", upsample=bool(up))
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
        y.backward(_rounded(dy, dt).double())
        xf = x.clone().requires_grad_(True)
        O.conv_block(xf, w, b, s, p, "none", upsample=bool(up)).backward(dy)
        wg, dyg = ohwi(w).cuda(), nhwc(dy).cuda()
        _, w16t = _packs(L, wg, dt)
        scr = _scratch(L.lib.aclgan_conv2d_dgrad16_scratch_bytes(C.byref(d)))
        dx = torch.full((B, Hi, Wi, Ci), float("nan"), device="cuda")
        L.check(L.lib.aclgan_conv2d_dgrad16(C.byref(d), L.DTYPE[dt], L.ptr(dyg), L.ptr(wg), L.ptr(w16t), L.ptr(dx), 0, L.ptr(scr),
                                            L.stream_ptr()), "conv2d_dgrad16")
        if not up or grid:
            assert rel_err(nchw(dx), xr.grad) < EXACT_TOL, ("exact", grid)
        assert rel_err(nchw(dx), xf.grad) < PREC_TOL[dt]
        # accumulate mode
        base = torch.randn(B, Hi, Wi, Ci, generator=torch.Generator().manual_seed(6))
        acc = base.clone().cuda()
        L.check(L.lib.aclgan_conv2d_dgrad16(C.byref(d), L.DTYPE[dt], L.ptr(dyg), L.ptr(wg), L.ptr(w16t), L.ptr(acc), 1, L.ptr(scr),
                                            L.stream_ptr()))
        assert rel_err(nchw(acc).cpu() - nchw(base), xf.grad) < PREC_TOL[dt] + 5 * EXACT_TOL


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [c for c in CASES16 if c[3] % 64 == 0 and c[4] % 64 == 0])
def test_conv_wgrad16(L, case, dt):
    from gpu_util import conv_desc, nhwc, ohwi, rel_err
    B, Hi, Wi, Ci, Co, k, s, p, up, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, "none")
    assert L.lib.aclgan_conv16_eligible(C.byref(d), 2) == 1
    x, w, b = _tensors(case, 2)
    wr = w.double().requires_grad_(True); br = b.double().requires_grad_(True)
    y = O.conv_block(_rounded(x, dt).double(), wr, br, s, p, "none", upsample=bool(up))
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
    y.backward(_rounded(dy, dt).double())
    wf = w.clone().requires_grad_(True); bf = b.clone().requires_grad_(True)
    O.conv_block(x, wf, bf, s, p, "none", upsample=bool(up)).backward(dy)
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    db = torch.zeros(Co, device="cuda")
    scr = _scratch(L.lib.aclgan_conv2d_wgrad16_scratch_bytes(C.byref(d)))
    xg, dyg = nhwc(x).cuda(), nhwc(dy).cuda()
    L.check(L.lib.aclgan_conv2d_wgrad16(C.byref(d), L.DTYPE[dt], L.ptr(xg), L.ptr(dyg), L.ptr(dw), L.ptr(db), L.ptr(scr),
                                        L.stream_ptr()), "conv2d_wgrad16")
    assert rel_err(dw, ohwi(wr.grad)) < EXACT_TOL      # exact path too for the sub-pixel layers: wgrad merges nothing before rounding
    assert rel_err(dw, ohwi(wf.grad)) < PREC_TOL[dt]
    assert rel_err(db, bf.grad) < EXACT_TOL            # the bias gradient is summed from the UNROUNDED fp32 dy
    # no atomics anywhere in the 16-bit weight gradient: slices are reduced in order -> reproducible bit for bit
    dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
    L.check(L.lib.aclgan_conv2d_wgrad16(C.byref(d), L.DTYPE[dt], L.ptr(xg), L.ptr(dyg), L.ptr(dw2), L.ptr(db2), L.ptr(scr), L.stream_ptr()))
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


def test_ineligible_shapes_are_refused(L):
    from gpu_util import conv_desc
    for case, which in (((2, 32, 32, 3, 64, 4, 2, 1, 0, "lrelu"), 0), ((2, 16, 16, 64, 4, 7, 1, 3, 0, "tanh"), 0),
                        ((2, 4, 4, 512, 1, 1, 1, 0, 0, "none"), 1), ((1, 10, 6, 32, 32, 3, 1, 1, 0, "none"), 2)):
        d = conv_desc(L, *case)
        assert L.lib.aclgan_conv16_eligible(C.byref(d), which) == 0
    d = conv_desc(L, 2, 32, 32, 3, 64, 4, 2, 1, 0, "lrelu")
    t = torch.zeros(16, device="cuda")
    assert L.lib.aclgan_conv2d_fwd16(C.byref(d), 1, L.ptr(t), None, L.ptr(t), None, L.ptr(t), None, L.stream_ptr()) == -2
    assert "16-bit" in L.last_error()
