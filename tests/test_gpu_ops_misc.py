"""The small operators of the step through the C ABI -- dense layers (MLP / style head), global average pool, focus blend,
the three loss reductions -- each forward AND backward against the CPU oracle; and the reference-generated operator
vectors (tests/golden/op_vectors.npz: outputs of the REFERENCE's own Conv2dBlock / MsImageDis / focus_translation,
written by tests/golden/make_golden.py) replayed directly on the HIP path, not only on the CPU oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import aclgan_oracle as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL = 2e-4


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def opv():
    return np.load(os.path.join(GOLDEN, "op_vectors.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def _act(x, act):
    return O._act(x, act)


# ---------------------------------------------------------------------------------------------
# dense layers (MLP 8 -> 256 -> 256 -> 4096, style head 256 -> 8): networks.py:280-292, 373-418, 223
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,I,Oo,act", [(8, 8, 256, "relu"), (3, 256, 256, "relu"), (2, 256, 4096, "none"), (5, 256, 8, "none"), (1, 16, 16, "relu")])
def test_linear_fwd_bwd(L, B, I, Oo, act):
    from gpu_util import rel_err
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, I, generator=g).requires_grad_(True)
    w = (torch.randn(Oo, I, generator=g) * (2.0 / I) ** 0.5).requires_grad_(True)
    b = (torch.randn(Oo, generator=g) * 0.1).requires_grad_(True)
    y = _act(F.linear(x, w, b), act)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xg, wg, bg = x.detach().cuda(), w.detach().cuda(), b.detach().cuda()
    yg = torch.empty(B, Oo, device="cuda")
    L.check(L.lib.aclgan_linear_fwd(B, I, Oo, L.ptr(xg), L.ptr(wg), L.ptr(bg), L.ACT[act], L.ptr(yg), L.stream_ptr()), "linear_fwd")
    assert rel_err(yg, y) < TOL
    dyg = dy.cuda()
    dx = torch.full((B, I), float("nan"), device="cuda")
    dw = torch.zeros(Oo, I, device="cuda"); db = torch.zeros(Oo, device="cuda")
    L.check(L.lib.aclgan_linear_bwd(B, I, Oo, L.ptr(xg), L.ptr(yg), L.ptr(dyg), L.ptr(wg), L.ACT[act], L.ptr(dx), L.ptr(dw), L.ptr(db),
                                    L.stream_ptr()), "linear_bwd")
    assert rel_err(dx, x.grad) < TOL and rel_err(dw, w.grad) < TOL and rel_err(db, b.grad) < TOL
    # dw / db accumulate (a generator's MLP is used up to three times per step)
    dyg = dy.cuda()
    L.check(L.lib.aclgan_linear_bwd(B, I, Oo, L.ptr(xg), L.ptr(yg), L.ptr(dyg), L.ptr(wg), L.ACT[act], None, L.ptr(dw), L.ptr(db), L.stream_ptr()))
    assert rel_err(dw, 2 * w.grad) < TOL and rel_err(db, 2 * b.grad) < TOL


@pytest.mark.parametrize("B,S,M,Oo", [(8, 8, 256, 4096), (3, 8, 256, 4096), (11, 8, 64, 100), (1, 64, 128, 7), (16, 5, 192, 2048)])
def test_mlp3_fwd_is_three_linear_fwd(L, B, S, M, Oo):
    """MLP.forward (networks.py:280-292) in one launch: bit-identical to three aclgan_linear_fwd calls (and those are checked against
    F.linear above); batches beyond 8 rows, style widths up to 64, ragged output counts"""
    g = torch.Generator().manual_seed(9)
    s = torch.randn(B, S, generator=g).cuda()
    w0 = (torch.randn(M, S, generator=g) * (2.0 / S) ** 0.5).cuda(); b0 = (torch.randn(M, generator=g) * 0.1).cuda()
    w1 = (torch.randn(M, M, generator=g) * (2.0 / M) ** 0.5).cuda(); b1 = (torch.randn(M, generator=g) * 0.1).cuda()
    w2 = (torch.randn(Oo, M, generator=g) * (2.0 / M) ** 0.5).cuda(); b2 = (torch.randn(Oo, generator=g) * 0.1).cuda()
    r0 = torch.empty(B, M, device="cuda"); r1 = torch.empty(B, M, device="cuda"); r2 = torch.empty(B, Oo, device="cuda")
    st = L.stream_ptr()
    L.check(L.lib.aclgan_linear_fwd(B, S, M, L.ptr(s), L.ptr(w0), L.ptr(b0), L.ACT["relu"], L.ptr(r0), st))
    L.check(L.lib.aclgan_linear_fwd(B, M, M, L.ptr(r0), L.ptr(w1), L.ptr(b1), L.ACT["relu"], L.ptr(r1), st))
    L.check(L.lib.aclgan_linear_fwd(B, M, Oo, L.ptr(r1), L.ptr(w2), L.ptr(b2), L.ACT["none"], L.ptr(r2), st))
    m0 = torch.full((B, M), float("nan"), device="cuda"); m1 = torch.full((B, M), float("nan"), device="cuda")
    ap = torch.full((B, Oo), float("nan"), device="cuda")
    L.check(L.lib.aclgan_mlp3_fwd(B, S, M, Oo, L.ptr(s), L.ptr(w0), L.ptr(b0), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(m0), L.ptr(m1),
                                  L.ptr(ap), st), "mlp3_fwd")
    torch.cuda.synchronize()
    assert torch.equal(m0, r0) and torch.equal(m1, r1) and torch.equal(ap, r2)
    ref = F.linear(F.relu(F.linear(F.relu(F.linear(s.cpu(), w0.cpu(), b0.cpu())), w1.cpu(), b1.cpu())), w2.cpu(), b2.cpu())
    assert ((ap.cpu() - ref).abs().max() / ref.abs().max()).item() < TOL


def test_mlp3_fwd_refuses_widths_it_does_not_cover(L):
    z = torch.zeros(8 * 512, device="cuda")
    rc = L.lib.aclgan_mlp3_fwd(1, 8, 512, 8, L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.stream_ptr())
    assert rc == -2, rc      # ACLGAN_EUNSUPPORTED: the engine then runs the three launches


@pytest.mark.parametrize("B,HW,Cn", [(8, 256, 256), (2, 16, 64), (3, 35, 8), (1, 1, 16)])
def test_gap_fwd_bwd(L, B, HW, Cn):
    from gpu_util import rel_err
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, HW, Cn, generator=g)
    xg = x.cuda()
    y = torch.empty(B, Cn, device="cuda")
    L.check(L.lib.aclgan_gap_fwd(B, HW, Cn, L.ptr(xg), L.ptr(y), L.stream_ptr()), "gap_fwd")
    assert rel_err(y, x.mean(1)) < TOL
    dy = torch.randn(B, Cn, generator=g)
    dyg = dy.cuda()
    dx = torch.full((B, HW, Cn), float("nan"), device="cuda")
    L.check(L.lib.aclgan_gap_bwd(B, HW, Cn, L.ptr(dyg), L.ptr(dx), 0, L.stream_ptr()), "gap_bwd")
    want = (dy / HW).unsqueeze(1).expand(B, HW, Cn)
    assert rel_err(dx, want) < TOL
    L.check(L.lib.aclgan_gap_bwd(B, HW, Cn, L.ptr(dyg), L.ptr(dx), 1, L.stream_ptr()))
    assert rel_err(dx, 2 * want) < TOL


# ---------------------------------------------------------------------------------------------
# focus blend (trainer.py:85-88) + the 6-channel pair (trainer.py:132-133)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,with_pair", [(2, 16, 16, True), (3, 9, 13, False), (1, 64, 64, True)])
def test_focus_blend_fwd_bwd(L, B, H, W, with_pair):
    from gpu_util import nhwc, nchw, rel_err
    g = torch.Generator().manual_seed(5)
    dec = torch.tanh(torch.randn(B, 4, H, W, generator=g)).requires_grad_(True)
    bg = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).requires_grad_(True)
    first = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    out = O.focus_translation(dec[:, :3], bg, dec[:, 3:])
    d_out = torch.randn(B, 3, H, W, generator=g)
    d_pair = torch.randn(B, 6, H, W, generator=g)
    loss = (out * d_out).sum()
    if with_pair:
        pair = torch.cat((first, out), 1)
        loss = loss + (pair * d_pair).sum()
    loss.backward()
    decg, bgg, firstg = nhwc(dec.detach()).cuda(), nhwc(bg.detach()).cuda(), nhwc(first).cuda()
    outg = torch.empty(B, H, W, 3, device="cuda")
    pairg = torch.empty(B, H, W, 6, device="cuda") if with_pair else None
    L.check(L.lib.aclgan_focus_blend_fwd(B, H * W, L.ptr(decg), L.ptr(bgg), L.ptr(outg), L.ptr(firstg) if with_pair else None, L.ptr(pairg),
                                         L.stream_ptr()), "focus_blend_fwd")
    assert rel_err(nchw(outg), out) < TOL
    if with_pair:
        assert rel_err(nchw(pairg), pair) < TOL
    d_outg = nhwc(d_out).cuda()
    d_pairg = nhwc(d_pair).cuda() if with_pair else None
    d_dec = torch.zeros(B, H, W, 4, device="cuda")
    d_bg = torch.full((B, H, W, 3), float("nan"), device="cuda")
    L.check(L.lib.aclgan_focus_blend_bwd(B, H * W, L.ptr(decg), L.ptr(bgg), L.ptr(d_outg), L.ptr(d_pairg), L.ptr(d_dec), L.ptr(d_bg), 0,
                                         L.stream_ptr()), "focus_blend_bwd")
    assert rel_err(nchw(d_dec), dec.grad) < TOL
    assert rel_err(nchw(d_bg), bg.grad) < TOL


def test_focus_translation_nchw_and_reference_vector(L, opv):
    """the NCHW entry point sample() / test.py use, on the REFERENCE's own focus_translation output"""
    from gpu_util import rel_err
    fg, bg, fo, want = (T(opv[k]).cuda() for k in ("ft_fg", "ft_bg", "ft_focus", "ft_y"))
    B, _, H, W = fg.shape
    out = torch.empty_like(fg)
    L.check(L.lib.aclgan_focus_translation_nchw(L.ptr(fg), fg.stride(0), L.ptr(bg), bg.stride(0), L.ptr(fo), fo.stride(0), L.ptr(out), B, H * W,
                                                L.stream_ptr()), "focus_translation_nchw")
    assert rel_err(out, want) < 1e-6
    # channel slices of one 4-channel decoder output (batch stride 4*H*W)
    dec = torch.cat((fg, fo), 1).contiguous()
    out2 = torch.empty_like(fg)
    L.check(L.lib.aclgan_focus_translation_nchw(L.ptr(dec), dec.stride(0), L.ptr(bg), bg.stride(0), C.c_void_p(dec.data_ptr() + 4 * 3 * H * W),
                                                dec.stride(0), L.ptr(out2), B, H * W, L.stream_ptr()))
    assert torch.equal(out, out2)


# ---------------------------------------------------------------------------------------------
# losses (networks.py:67,83,98; trainer.py:61-62,146-158)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,target,weight", [(2048, 1.0, 0.5), (64, 0.0, 1.0), (7, 1.0, 1.0)])
def test_lsgan_loss(L, n, target, weight):
    from gpu_util import rel_err
    g = torch.Generator().manual_seed(6)
    o = torch.randn(n, generator=g).requires_grad_(True)
    loss = weight * torch.mean((o - target) ** 2)
    (0.2 * loss).backward()
    og = o.detach().cuda()
    slot = torch.full((1,), 0.25, device="cuda")
    d_o = torch.full((n,), float("nan"), device="cuda")
    L.check(L.lib.aclgan_lsgan_loss(L.ptr(og), n, target, weight, L.ptr(slot), L.ptr(d_o), 0.2, L.stream_ptr()), "lsgan_loss")
    assert abs(float(slot) - 0.25 - float(loss.detach())) <= 1e-5 * max(1.0, float(loss.detach()))     # accumulates into the slot
    assert rel_err(d_o, o.grad) < TOL


@pytest.mark.parametrize("npix,ach", [(8 * 64 * 64, 4), (1000, 3)])
def test_l1_loss(L, npix, ach):
    from gpu_util import rel_err
    g = torch.Generator().manual_seed(7)
    a = torch.randn(npix, ach, generator=g).requires_grad_(True)
    b = torch.randn(npix, 3, generator=g)
    loss = torch.mean(torch.abs(a[:, :3] - b))
    (1.5 * loss).backward()
    ag, bgg = a.detach().cuda(), b.cuda()
    slot = torch.zeros(1, device="cuda")
    d_a = torch.full((npix, ach), float("nan"), device="cuda")
    L.check(L.lib.aclgan_l1_loss(L.ptr(ag), ach, L.ptr(bgg), npix, L.ptr(slot), L.ptr(d_a), 1.5, 0, L.stream_ptr()), "l1_loss")
    assert abs(float(slot) - float(loss.detach())) <= 1e-5 * float(loss.detach())
    assert rel_err(d_a[:, :3], a.grad[:, :3]) < TOL
    if ach == 4:
        assert float(d_a[:, 3].abs().max()) == 0.0


@pytest.mark.parametrize("npix,shift", [(8 * 64 * 64, 0.0), (2 * 64 * 64, 0.6), (3000, -0.7)])
def test_focus_loss(L, npix, shift):
    """size / digit losses and their gradient; shift moves the mask mean so that each relu branch of the size loss is hit.
    The ordered, centred summation makes the size loss accurate to 1e-4 (the reference's own fp32 sum: ~1e-2) and the
    result bit-reproducible."""
    from gpu_util import rel_err
    g = torch.Generator().manual_seed(8)
    dec = torch.tanh(torch.randn(npix, 4, generator=g) * 0.5 + shift).requires_grad_(True)
    hp = dict(focus_upper=0.5, focus_lower=0.3, focus_delta=0.001, focus_epsilon=0.01)
    size, digit = O.focus_losses(dec.double()[:, 3], hp)
    scale = 0.025 / npix / 3
    (scale * (size + digit)).backward()
    decg = dec.detach().cuda()
    scr = torch.empty(L.lib.aclgan_focus_loss_scratch_bytes(npix) // 4 + 16, device="cuda")
    outs = []
    for _ in range(2):
        slots = torch.zeros(2, device="cuda")
        d_dec = torch.zeros(npix, 4, device="cuda")
        L.check(L.lib.aclgan_focus_loss(L.ptr(decg), npix, 0.001, 0.5, 0.3, 0.01, scale, L.ptr(slots), C.c_void_p(slots.data_ptr() + 4), L.ptr(d_dec),
                                        L.ptr(scr), L.stream_ptr()), "focus_loss")
        outs.append((slots.clone(), d_dec.clone()))
    slots, d_dec = outs[0]
    assert abs(float(slots[0]) - float(size)) <= 1e-4 * max(1e-6, float(size)), (float(slots[0]), float(size))
    assert abs(float(slots[1]) - float(digit)) <= 1e-5 * float(digit)
    assert rel_err(d_dec[:, 3], dec.grad[:, 3]) < TOL
    assert float(d_dec[:, :3].abs().max()) == 0.0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_focus_loss_global_shards(L):
    """data parallelism with global-batch semantics (aclgan_set_forward_sync): each shard, given the all-reduced sums,
    reports the full batch's losses and produces the full batch's gradient rows."""
    from gpu_util import rel_err
    g = torch.Generator().manual_seed(9)
    npix = 2 * 4096
    dec = torch.tanh(torch.randn(npix, 4, generator=g) * 0.5 + 0.6).cuda()       # mask mean high: the relu branch is active
    scale = 0.025 / npix / 3
    scr = torch.empty(L.lib.aclgan_focus_loss_scratch_bytes(npix) // 4 + 16, device="cuda")
    full_slots = torch.zeros(2, device="cuda"); full_d = torch.zeros(npix, 4, device="cuda")
    L.check(L.lib.aclgan_focus_loss(L.ptr(dec), npix, 0.001, 0.5, 0.3, 0.01, scale, L.ptr(full_slots), C.c_void_p(full_slots.data_ptr() + 4),
                                    L.ptr(full_d), L.ptr(scr), L.stream_ptr()))
    m = (dec[:, 3].double() + 1) / 2
    totals = torch.tensor([float((m - 0.5).sum()), float((1 / ((m - 0.5).abs() + 0.01)).sum())], device="cuda")
    for r in range(2):
        shard = dec[r * 4096:(r + 1) * 4096].contiguous()
        slots = torch.zeros(2, device="cuda"); d = torch.zeros(4096, 4, device="cuda")
        L.check(L.lib.aclgan_focus_loss_global(L.ptr(shard), 4096, L.ptr(totals), npix, 0.001, 0.5, 0.3, 0.01, scale, L.ptr(slots),
                                               C.c_void_p(slots.data_ptr() + 4), L.ptr(d), L.stream_ptr()), "focus_loss_global")
        assert rel_err(slots, full_slots) < 1e-5
        assert rel_err(d, full_d[r * 4096:(r + 1) * 4096]) < 1e-5
        assert float(full_slots[0]) > 0


# ---------------------------------------------------------------------------------------------
# reference-generated operator vectors, replayed on the HIP path
# ---------------------------------------------------------------------------------------------
def _gpu_conv_block(L, x, w, b, s, p, act, norm, args, up=0):
    """Conv2dBlock.forward (networks.py:365-371) out of the library's conv + norm operators, NCHW in / out."""
    from gpu_util import conv_desc, gpu_conv_fwd, nhwc, nchw, ohwi
    B, Ci, Hi, Wi = x.shape
    Co, k = w.shape[0], w.shape[2]
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, act if norm == "none" else "none")
    xg, wg, bg = nhwc(x).cuda(), ohwi(w).cuda(), b.cuda()
    y = gpu_conv_fwd(L, d, xg, wg, bg)
    if norm == "none":
        return nchw(y)
    Bc, Ho, Wo, _ = y.shape
    HW = Ho * Wo
    out = torch.empty_like(y)
    nstat = Bc if norm == "ln" else Bc * Co
    mean = torch.empty(nstat, device="cuda"); rstd = torch.empty(nstat, device="cuda")
    scratch = torch.empty(L.lib.aclgan_norm_scratch_bytes(Bc, HW, Co) // 4 + 16, device="cuda")
    wn = bn = None
    stride = 0
    if norm == "adain":
        wn, bn = args[0].reshape(Bc, Co).contiguous().cuda(), args[1].reshape(Bc, Co).contiguous().cuda()
        stride = Co
    elif norm == "ln":
        wn, bn = args[0].cuda(), args[1].cuda()
    L.check(L.lib.aclgan_norm_fwd(L.NORM[norm], L.ACT[act], Bc, HW, Co, L.ptr(y), L.ptr(wn), L.ptr(bn), stride, None, L.ptr(out), L.ptr(mean),
                                  L.ptr(rstd), L.ptr(scratch), L.stream_ptr()), "norm_fwd")
    return nchw(out)


def test_reference_conv_block_vectors_on_hip(L, opv):
    """every (pad, norm, activation) combination the shipped config reaches, B = 1 and B = 2 (the custom LayerNorm has
    separate code paths for the two in the reference, networks.py:523-529), outputs written by the reference itself"""
    combos = json.loads(str(opv["cb_combos"]))
    for i, (ci, co, k, s, p, norm, act, H) in enumerate(combos):
        for B in (1, 2):
            key = "cb%d_B%d" % (i, B)
            args = None
            if norm == "adain":
                args = (T(opv[key + "_adain_w"]), T(opv[key + "_adain_b"]))
            if norm == "ln":
                args = (T(opv[key + "_gamma"]), T(opv[key + "_beta"]))
            y = _gpu_conv_block(L, T(opv[key + "_x"]), T(opv[key + "_w"]), T(opv[key + "_b"]), s, p, act, norm, args)
            ref = T(opv[key + "_y"])
            assert tuple(y.shape) == tuple(ref.shape)
            assert (y.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), key
    y = _gpu_conv_block(L, T(opv["up_x"]), T(opv["up_w"]), T(opv["up_b"]), 1, 2, "relu", "ln", (T(opv["up_gamma"]), T(opv["up_beta"])), up=1)
    assert (y.cpu() - T(opv["up_y"])).abs().max().item() < 1e-4


def test_reference_avgpool_vectors_on_hip(L, opv):
    from gpu_util import nhwc, nchw

    def pool(x):
        B, Cn, H, W = x.shape
        xg = nhwc(x).cuda()
        y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cn, device="cuda")
        L.check(L.lib.aclgan_avgpool3s2_fwd(B, H, W, Cn, L.ptr(xg), L.ptr(y), L.stream_ptr()), "avgpool")
        return nchw(y).cpu()
    x = T(opv["pool_x"])
    assert (pool(x) - T(opv["pool_y1"])).abs().max().item() < 1e-6
    assert (pool(pool(x)) - T(opv["pool_y2"])).abs().max().item() < 1e-6
    assert (pool(T(opv["pool_odd_x"])) - T(opv["pool_odd_y"])).abs().max().item() < 1e-6


def test_reference_lsgan_conventions_on_hip(L, opv):
    """calc_dis_loss / calc_gen_loss / calc_gen_d2_loss target conventions (networks.py:60-106) with the discriminator
    forward AND the loss reduction on the HIP path, against the three loss values the reference computed."""
    import aclgan_amd  # noqa: F401
    from aclgan_amd.trainer import aclgan_Trainer
    cfg = O.default_config()
    cfg["display_size"] = 1
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1)
    cfg["dis"].update(dim=4)
    tr = aclgan_Trainer(cfg)
    P = {k[len("lsgan_D_"):]: T(opv[k]) for k in opv.files if k.startswith("lsgan_D_")}
    tr.dis_2.load_state_dict(P)                     # the fixture's D takes 6 channels: it is the consistency discriminator
    xf, xr = T(opv["lsgan_x_fake"]).cuda(), T(opv["lsgan_x_real"]).cuda()

    def total(terms):
        slot = torch.zeros(1, device="cuda")
        for outs, target in terms:
            for o in outs:
                L.check(L.lib.aclgan_lsgan_loss(L.ptr(o), o.numel(), target, 1.0, L.ptr(slot), None, 0.0, L.stream_ptr()), "lsgan_loss")
        return float(slot)
    of, orr = tr.dis_2(xf), tr.dis_2(xr)
    assert abs(total([(of, 0.0), (orr, 1.0)]) - float(opv["lsgan_dis_loss"])) <= 1e-4 * float(opv["lsgan_dis_loss"])      # fake -> 0, real -> 1
    assert abs(total([(of, 1.0)]) - float(opv["lsgan_gen_loss"])) <= 1e-4 * float(opv["lsgan_gen_loss"])                    # fake -> 1
    assert abs(total([(of, 1.0), (orr, 0.0)]) - float(opv["lsgan_gen_d2_loss"])) <= 1e-4 * float(opv["lsgan_gen_d2_loss"])  # pair_A1 -> 1, pair_A2 -> 0


@pytest.mark.parametrize("T_,K,N,ns", [(2048, 256, 256, 36), (200, 64, 48, 5), (961, 128, 64, 144)])
def test_gemm_slices_f32_vs_bmm(L, T_, K, N, ns):
    """aclgan_gemm_slices_f32 (the Winograd pipeline's batched-GEMM launch, exported for bench.py's dominant-kernel probe):
    C[f] = A[f] B[f]^T against a float64 bmm; ragged row / column tiles included."""
    g = torch.Generator().manual_seed(5)
    A = torch.randn(ns, T_, K, generator=g).cuda(); Bm = torch.randn(ns, N, K, generator=g).cuda()
    Cm = torch.full((ns, T_, N), float("nan"), device="cuda")
    L.check(L.lib.aclgan_gemm_slices_f32(L.ptr(A), L.ptr(Bm), L.ptr(Cm), T_, K, N, ns, L.stream_ptr()), "gemm_slices_f32")
    ref = torch.bmm(A.double(), Bm.double().transpose(1, 2))
    assert ((Cm.double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6


@pytest.mark.parametrize("T_,K,N,ns", [(2048, 256, 256, 36), (200, 64, 64, 5), (961, 128, 192, 7), (130, 32, 128, 3)])
@pytest.mark.parametrize("scale", ["unit", "wide"])
def test_gemm_slices_x3_is_fp32_accurate(L, T_, K, N, ns, scale):
    """aclgan_gemm_slices_x3 (csrc/gemm_bf16x3.hip: fp32 operands split exactly into three bf16 numbers, six bf16 x bf16 products per
    multiply accumulated in fp32 on the bf16 matrix cores) against a float64 bmm: the error must be that of fp32 arithmetic -- the
    same bound as the fp32 MFMA kernel above, and no worse than 2x that kernel's own error on the same data -- NOT that of a bf16 GEMM
    (4e-3).  `wide`: operand magnitudes spread over 12 decades row by row (the split is relative to each element: bf16 keeps fp32's
    exponent range)."""
    g = torch.Generator().manual_seed(7)
    A = torch.randn(ns, T_, K, generator=g); Bm = torch.randn(ns, N, K, generator=g)
    if scale == "wide":
        A = A * (10.0 ** torch.randint(-6, 7, (ns, T_, 1), generator=g).float())
        Bm = Bm * (10.0 ** torch.randint(-6, 7, (ns, N, 1), generator=g).float())
    A, Bm = A.cuda(), Bm.cuda()
    ref = torch.bmm(A.double(), Bm.double().transpose(1, 2))
    # error scale of an fp32 dot product: |a| . |b| per output element
    mag = torch.bmm(A.double().abs(), Bm.double().abs().transpose(1, 2)).clamp_min(1e-300)
    C3 = torch.full((ns, T_, N), float("nan"), device="cuda")
    scr = torch.empty(L.lib.aclgan_gemm_slices_x3_scratch_bytes(T_, K, N, ns) // 4 + 64, device="cuda")
    L.check(L.lib.aclgan_gemm_slices_x3(L.ptr(A), L.ptr(Bm), L.ptr(C3), T_, K, N, ns, L.ptr(scr), L.stream_ptr()), "gemm_slices_x3")
    C1 = torch.full((ns, T_, N), float("nan"), device="cuda")
    L.check(L.lib.aclgan_gemm_slices_f32(L.ptr(A), L.ptr(Bm), L.ptr(C1), T_, K, N, ns, L.stream_ptr()), "gemm_slices_f32")
    e3 = ((C3.double() - ref).abs() / mag).max().item()
    e1 = ((C1.double() - ref).abs() / mag).max().item()
    print("gemm_slices x3 vs f32 (T=%d K=%d N=%d, %s): componentwise rel err %.3g vs %.3g (fp32 eps 6e-8)" % (T_, K, N, scale, e3, e1))
    assert e3 <= 5e-7                          # a handful of fp32 roundings (eps 6e-8); a bf16 product would be ~4e-3
    assert e3 <= 2.0 * e1 + 1e-7
    if scale == "unit":
        assert ((C3.double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6
    C3b = torch.empty_like(C3)
    L.check(L.lib.aclgan_gemm_slices_x3(L.ptr(A), L.ptr(Bm), L.ptr(C3b), T_, K, N, ns, L.ptr(scr), L.stream_ptr()))
    assert torch.equal(C3, C3b)
