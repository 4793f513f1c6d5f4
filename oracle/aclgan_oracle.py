"""CPU oracle for the ACL-GAN training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain-PyTorch fp32 *restatement* of the algorithm of the reference's
``aclgan_Trainer.gen_update`` / ``dis_update`` hot path (reference: trainer.py:90-170,
trainer.py:247-293 and everything in networks.py they reach).  It exists so the HIP path
can be checked against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``acl-gan_amd/``) never does and fails loudly when its HIP library is missing.

Pinning: the reference ships no tests / golden vectors of its own, so
this oracle is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (which imports /root/reference on CPU) and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them.

Design notes
  * functional style: every network is a flat ``dict`` name -> tensor whose keys are the
    reference's ``state_dict`` keys (OIHW weights, NCHW activations), so reference
    checkpoints load without renaming;
  * style noise ``z`` is always passed in explicitly (the reference draws it from the CPU
    generator inside the update, trainer.py:99-101 / 254-256) so that parity tests are
    independent of RNG plumbing;
  * autograd is used for the backward pass (the reference does the same, trainer.py:169,292);
    Adam is restated by hand (torch.optim.Adam semantics, trainer.py:39-42).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------------------
# configuration helpers
# --------------------------------------------------------------------------------------

DEFAULT_HP = dict(
    # configs/male2female.yaml:12-36
    weight_decay=0.0001, beta1=0.5, beta2=0.999, init="kaiming", lr=0.0001,
    lr_policy="step", step_size=100000, gamma=0.5,
    gan_w=1, gan_cw=0.2, focus_loss=0.025, focus_delta=0.001, focus_upper=0.5,
    focus_lower=0.3, focus_epsilon=0.01, recon_x_w=1, vgg_w=0, alpha=1,
    G_update=2, D_update=1, display_size=16, batch_size=3,
    # configs/male2female.yaml:39-55
    gen=dict(dim=64, mlp_dim=256, style_dim=8, output_dim=4, activ="relu",
             n_downsample=2, n_res=4, pad_type="reflect"),
    dis=dict(dim=64, norm="none", activ="lrelu", n_layer=4, gan_type="lsgan",
             num_scales=3, pad_type="reflect"),
    input_dim_a=3, input_dim_b=6,
)


def default_config() -> dict:
    import copy
    return copy.deepcopy(DEFAULT_HP)


# --------------------------------------------------------------------------------------
# parameter construction (names follow the reference state_dict)
# --------------------------------------------------------------------------------------

def gen_param_shapes(input_dim: int, g: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameter name -> shape for one AdaINGen, in the reference's ``parameters()`` order
    (networks.py:112-135: enc_style, enc_content, dec, mlp; Conv2dBlock registers norm
    before conv, networks.py:327-363)."""
    dim, sd, nd, nr = g["dim"], g["style_dim"], g["n_downsample"], g["n_res"]
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(prefix, co, ci, k):
        out[prefix + ".weight"] = (co, ci, k, k)
        out[prefix + ".bias"] = (co,)

    # StyleEncoder(4, ...)  networks.py:212-228 (hard-coded 4 downsamples, networks.py:126)
    d = dim
    conv("enc_style.model.0.conv", d, input_dim, 7)
    for i in range(2):
        conv("enc_style.model.%d.conv" % (1 + i), 2 * d, d, 4)
        d *= 2
    for i in range(4 - 2):
        conv("enc_style.model.%d.conv" % (3 + i), d, d, 4)
    conv("enc_style.model.6", sd, d, 1)
    # ContentEncoder  networks.py:230-245
    d = dim
    conv("enc_content.model.0.conv", d, input_dim, 7)
    for i in range(nd):
        conv("enc_content.model.%d.conv" % (1 + i), 2 * d, d, 4)
        d *= 2
    for r in range(nr):
        for j in range(2):
            conv("enc_content.model.%d.model.%d.model.%d.conv" % (1 + nd, r, j), d, d, 3)
    # Decoder  networks.py:247-264
    for r in range(nr):
        for j in range(2):
            conv("dec.model.0.model.%d.model.%d.conv" % (r, j), d, d, 3)
    idx = 1
    for i in range(nd):
        out["dec.model.%d.norm.gamma" % (idx + 1)] = (d // 2,)
        out["dec.model.%d.norm.beta" % (idx + 1)] = (d // 2,)
        conv("dec.model.%d.conv" % (idx + 1), d // 2, d, 5)
        d //= 2
        idx += 2
    conv("dec.model.%d.conv" % idx, g["output_dim"], d, 7)
    # MLP(style_dim, num_adain, mlp_dim, 3)  networks.py:280-292
    n_adain = 2 * (dim * 2 ** nd) * 2 * nr  # get_num_adain_params, networks.py:165-171
    md = g["mlp_dim"]
    out["mlp.model.0.fc.weight"] = (md, sd); out["mlp.model.0.fc.bias"] = (md,)
    out["mlp.model.1.fc.weight"] = (md, md); out["mlp.model.1.fc.bias"] = (md,)
    out["mlp.model.2.fc.weight"] = (n_adain, md); out["mlp.model.2.fc.bias"] = (n_adain,)
    return out


def gen_buffer_shapes(g: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """The dummy AdaIN running_mean/var buffers (networks.py:488-489): never used or updated,
    but present in the reference state_dict."""
    d = g["dim"] * 2 ** g["n_downsample"]
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for r in range(g["n_res"]):
        for j in range(2):
            out["dec.model.0.model.%d.model.%d.norm.running_mean" % (r, j)] = (d,)
            out["dec.model.0.model.%d.model.%d.norm.running_var" % (r, j)] = (d,)
    return out


def dis_param_shapes(input_dim: int, dcfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """MsImageDis parameters (networks.py:21-48)."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for s in range(dcfg["num_scales"]):
        d = dcfg["dim"]
        out["cnns.%d.0.conv.weight" % s] = (d, input_dim, 4, 4)
        out["cnns.%d.0.conv.bias" % s] = (d,)
        for i in range(dcfg["n_layer"] - 1):
            out["cnns.%d.%d.conv.weight" % (s, i + 1)] = (2 * d, d, 4, 4)
            out["cnns.%d.%d.conv.bias" % (s, i + 1)] = (2 * d,)
            d *= 2
        out["cnns.%d.%d.weight" % (s, dcfg["n_layer"])] = (1, d, 1, 1)
        out["cnns.%d.%d.bias" % (s, dcfg["n_layer"])] = (1,)
    return out


def init_params(shapes, kind: str, gen: Optional[torch.Generator] = None) -> Params:
    """Statistically equivalent initialisation (utils.py:274-294, trainer.py:49-52):
    'kaiming' = N(0, 2/fan_in) on conv/linear weights, zero bias, LN gamma ~ U(0,1)
    (networks.py:517), beta 0; 'gaussian' = N(0, 0.02).  Bit-exact init parity with the
    reference RNG stream is not required."""
    p: Params = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith(".weight"):
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            std = math.sqrt(2.0 / fan_in) if kind == "kaiming" else 0.02
            p[name] = torch.randn(shp, generator=gen) * std
        elif name.endswith(".gamma"):
            p[name] = torch.rand(shp, generator=gen)
        else:
            p[name] = torch.zeros(shp)
    return p


def seeded_fill(shapes, kind: str, seed: int) -> Params:
    """Deterministic test fill used by the golden fixtures and parity tests: reference init
    statistics for the weights (see init_params) but NON-zero biases / LN beta (N(0, 0.05)),
    so that bias paths are exercised.  Drawn from a seeded CPU torch.Generator; fixtures store
    per-tensor checksums so RNG drift across torch versions would be detected."""
    gen = torch.Generator().manual_seed(seed)
    p = init_params(shapes, kind, gen)
    for name in p:
        if name.endswith(".bias") or name.endswith(".beta"):
            p[name] = torch.randn(p[name].shape, generator=gen) * 0.05
    return p


def test_nets(hp: dict, seed: int = 0) -> Dict[str, Params]:
    """The five networks of aclgan_Trainer (trainer.py:19-23) with the seeded test fill."""
    gs = gen_param_shapes(hp["input_dim_a"], hp["gen"])
    return {
        "gen_AB": seeded_fill(gs, "kaiming", seed * 10 + 1),
        "gen_BA": seeded_fill(gs, "kaiming", seed * 10 + 2),
        "dis_A": seeded_fill(dis_param_shapes(hp["input_dim_a"], hp["dis"]), "gaussian", seed * 10 + 3),
        "dis_B": seeded_fill(dis_param_shapes(hp["input_dim_a"], hp["dis"]), "gaussian", seed * 10 + 4),
        "dis_2": seeded_fill(dis_param_shapes(hp["input_dim_b"], hp["dis"]), "gaussian", seed * 10 + 5),
    }


# --------------------------------------------------------------------------------------
# blocks (networks.py:312-371, 477-536)
# --------------------------------------------------------------------------------------

# Activation-mask hook (test infrastructure for tests/test_gpu_maskfrozen.py; NOT reference behaviour).  Two correct fp32 implementations
# of this step disagree on the branch of a ReLU / LeakyReLU wherever a pre-activation lies within rounding of zero, and every such flip moves
# all upstream gradients ("mask lottery").  Inside `with act_masks(replay) as rec:` every activation of a Conv2dBlock (call i of _act, in
# call order) records its own mask (x > 0) in rec.recorded[i] and, when replay holds an entry i, APPLIES that mask instead: relu(x) = x * m,
# lrelu(x) = where(m, x, 0.2 x), with m treated as a constant by autograd -- the backward then runs with exactly the masks the HIP update
# ran with (include/aclgan_hip.h: aclgan_debug_capture_masks), and what remains is the error of the backward kernels themselves.
_MASK_HOOK = None


class act_masks:
    def __init__(self, replay=None, replay_signs=None):
        self.replay = dict(replay or {})
        self.recorded = []
        # the step's other two sign decisions: |m - 0.5| of the focus "digit" loss (trainer.py:151) and |x_recon - x| of the identity losses
        # (trainer.py:162-165): call j of _abs records (d > 0) in signs[j] and applies replay_signs[j] instead when given
        self.replay_signs = dict(replay_signs or {})
        self.signs = []

    def __enter__(self):
        global _MASK_HOOK
        self._prev = _MASK_HOOK
        _MASK_HOOK = self
        return self

    def __exit__(self, *a):
        global _MASK_HOOK
        _MASK_HOOK = self._prev

    def apply(self, x, act):
        i = len(self.recorded)
        own = (x > 0).detach()
        self.recorded.append(own)
        m = self.replay.get(i)
        if m is None:
            m = own
        assert m.shape == x.shape, (i, tuple(m.shape), tuple(x.shape))
        if act == "relu":
            return torch.where(m, x, torch.zeros_like(x))
        return torch.where(m, x, 0.2 * x)


def _abs(d):
    """|d| at the step's two non-smooth loss sites (see act_masks)"""
    if _MASK_HOOK is None:
        return torch.abs(d)
    h = _MASK_HOOK
    j = len(h.signs)
    own = (d > 0).detach()
    h.signs.append(own)
    m = h.replay_signs.get(j)
    if m is None:
        m = own
    assert m.shape == d.shape, (j, tuple(m.shape), tuple(d.shape))
    return torch.where(m, d, -d)


def _act(x, act):
    if _MASK_HOOK is not None and act in ("relu", "lrelu"):
        return _MASK_HOOK.apply(x, act)
    if act == "relu":
        return F.relu(x)
    if act == "lrelu":
        return F.leaky_relu(x, 0.2)  # networks.py:347
    if act == "tanh":
        return torch.tanh(x)
    assert act == "none", act
    return x


def instance_norm(x):
    """nn.InstanceNorm2d(affine=False): biased variance, eps 1e-5 inside the sqrt
    (networks.py:333)."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + 1e-5)


def adain(x, weight, bias):
    """AdaptiveInstanceNorm2d via batch_norm on a (1, B*C, H, W) view (networks.py:491-503):
    instance norm with biased variance, then per-(b,c) weight/bias of shape (B, C)."""
    b, c = x.shape[:2]
    return instance_norm(x) * weight.view(b, c, 1, 1) + bias.view(b, c, 1, 1)


def layer_norm_munit(x, gamma, beta):
    """The custom LayerNorm (networks.py:520-536): per-sample mean and UNBIASED std over
    C*H*W, eps added to the std (not inside a sqrt), then per-channel gamma/beta."""
    b = x.shape[0]
    flat = x.reshape(b, -1)
    mean = flat.mean(1).view(b, 1, 1, 1)
    std = flat.std(1).view(b, 1, 1, 1)  # unbiased
    y = (x - mean) / (std + 1e-5)
    return y * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


# --------------------------------------------------------------------------------------
# emulation of the build's reduced-precision COMPUTE contract (NOT reference behaviour: the reference
# is fp32 only).  Used by tests/test_gpu_step16.py to check that the bf16 / fp16 HIP path computes exactly
# what acl-gan_amd/csrc/conv_fast16.hip states: for every convolution whose channel counts are multiples of
# 32 (weight gradient: 64) both GEMM operands are rounded to the 16-bit type (round to nearest even), the
# products accumulate in (at least) fp32, everything else stays fp32.  The sub-pixel path of the
# "Upsample(2) + 5x5" layers rounds the MERGED 3x3 phase filters (forward / dgrad, interior pixels only).
# --------------------------------------------------------------------------------------
# Round 3 -- 16-bit STORAGE: under a 16-bit compute dtype the build also keeps the activations and activation gradients of its wide
# layers in HBM in that dtype (acl-gan_amd/csrc/engine.hip::conv_block):
#   * the output of every Conv2dBlock with Cout % 64 == 0 is stored rounded (after norm / activation / residual add) -- except the one
#     that feeds the 7x7 image-side output layer and the last layer of every discriminator scale (their consumers are fp32 kernels);
#   * the conv output in front of a normalisation layer is stored rounded when the layer runs on the 16-bit-storage kernels (no
#     upsample, Cin and Cout multiples of 64); statistics are then those of the stored values;
#   * a gradient is rounded (at the loss scale it carries) where it is stored: the gradient w.r.t. a 16-bit activation -- unless a
#     sub-pixel layer consumes it (those input-gradient kernels write fp32) -- and the gradient w.r.t. a 16-bit-stored conv output.
# For a convolution operand none of this changes a value (the conv loaders rounded the same fp32 numbers before); what changes is
# what the residual adds, the normalisation layers, the global average pool and the backward of norm / activation read.
_QDT = None   # torch.bfloat16 / torch.float16 while emulating, else None (= the plain fp32 oracle)
_QSCALE = 1.0  # fp16 loss scale S: every gradient that enters a 16-bit GEMM is S*dy (the build seeds the backward with S)
_QSTORE = True  # emulate the 16-bit storage as well (compute_dtype(..., storage=False): the round-2 contract, fp32 storage)
_QCO16 = True   # ... including the conv outputs in front of normalisation layers (the build's ACLGAN_CO16 switch)


class compute_dtype:
    """with compute_dtype("bf16"): ... / with compute_dtype("fp16", loss_scale=65536.0): ...
    conv_block emulates the 16-bit MFMA contract inside the block (gradients come out UNscaled)."""

    def __init__(self, name, loss_scale=1.0, storage=True):
        self.dt = {"fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[name]
        self.scale = float(loss_scale)
        self.storage = bool(storage)

    def __enter__(self):
        global _QDT, _QSCALE, _QSTORE
        self.prev, _QDT, _QSCALE, _QSTORE = (_QDT, _QSCALE, _QSTORE), self.dt, self.scale, self.storage

    def __exit__(self, *a):
        global _QDT, _QSCALE, _QSTORE
        _QDT, _QSCALE, _QSTORE = self.prev


def _q(t):
    return t.to(_QDT).to(t.dtype)


def _qg(dy):
    """a gradient operand: rounded at the loss scale it carries in the build"""
    return _q(dy * _QSCALE) / _QSCALE


class _StoreQ(torch.autograd.Function):
    """a tensor that lives in HBM in the 16-bit dtype: rounded on the way in; its gradient rounded likewise (g16) or kept fp32"""

    @staticmethod
    def forward(ctx, x, g16):
        ctx.g16 = g16
        return _q(x)

    @staticmethod
    def backward(ctx, dy):
        return (_qg(dy) if ctx.g16 else dy), None


def _plain_conv(x, w, stride, pad, upsample):
    if upsample:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    return F.conv2d(x, w, None, stride=stride)


_UP5_SETS = {0: ((0, 1), (2, 3), (4,)), 1: ((0,), (1, 2), (3, 4))}   # filter rows merged onto low-res row a, per output phase


def _up5_merged(w, py, px):
    """w (Co,Ci,5,5) -> the 3x3 phase filter of output phase (py,px) (csrc/conv_fast.hip: up5_merge_kernel)."""
    rows = [sum(w[:, :, ky, :] for ky in _UP5_SETS[py][a]) for a in range(3)]          # 3 x (Co,Ci,5)
    return torch.stack([torch.stack([sum(r[:, :, kx] for kx in _UP5_SETS[px][b]) for b in range(3)], -1) for r in rows], -2)


def _up5_forward(x, w, wq):
    """Upsample(2)+reflect-pad(2)+5x5: output ring of width 2 from the exact gather with wq; interior from the four
    VALID 3x3 phase convolutions of the low-res input with the ROUNDED MERGED filters."""
    y = _plain_conv(x, wq, 1, 2, True)
    H, W = x.shape[2], x.shape[3]
    out = y.clone()
    for py in (0, 1):
        for px in (0, 1):
            out[:, :, 2 + py: 2 * H - 2: 2, 2 + px: 2 * W - 2: 2] = F.conv2d(x, _q(_up5_merged(w, py, px)))
    return out


class _ConvQ(torch.autograd.Function):
    """One convolution under the 16-bit compute contract (see above); flags say which of forward / dgrad / wgrad
    have a 16-bit kernel for this shape (acl-gan_amd/csrc/conv_fast16.hip: fwd16_ok / dgrad16_ok / wgrad16_ok)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, upsample, f16, d16, w16):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, upsample, d16, w16)
        up5 = upsample and w.shape[2] == 5 and pad == 2 and stride == 1 and x.shape[2] >= 4 and x.shape[3] >= 4
        ctx.up5 = up5
        if not f16:
            y = _plain_conv(x, w, stride, pad, upsample)
        elif up5:
            y = _up5_forward(_q(x), w, _q(w))
        else:
            y = _plain_conv(_q(x), _q(w), stride, pad, upsample)
        return y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, upsample, d16, w16 = ctx.cfg
        db = dy.sum(dim=(0, 2, 3))                     # summed from the unrounded fp32 dy
        with torch.enable_grad():
            # ---- dgrad ----
            xv = x.detach().requires_grad_(True)
            if d16 and ctx.up5:
                dyq = _qg(dy)
                H, W = x.shape[2], x.shape[3]
                ring = torch.ones_like(dyq)
                ring[:, :, 2: 2 * H - 2, 2: 2 * W - 2] = 0
                tot = (_plain_conv(xv, _q(w), 1, 2, True) * (dyq * ring)).sum()
                for py in (0, 1):
                    for px in (0, 1):
                        tot = tot + (F.conv2d(xv, _q(_up5_merged(w, py, px))) * dyq[:, :, 2 + py: 2 * H - 2: 2, 2 + px: 2 * W - 2: 2]).sum()
                dx, = torch.autograd.grad(tot, xv)
            elif d16:
                dx, = torch.autograd.grad(_plain_conv(xv, _q(w), stride, pad, upsample), xv, _qg(dy))
            else:
                dx, = torch.autograd.grad(_plain_conv(xv, w, stride, pad, upsample), xv, dy)
            # ---- wgrad ----
            wv = w.detach().requires_grad_(True)
            if w16:
                dw, = torch.autograd.grad(_plain_conv(_q(x), wv, stride, pad, upsample), wv, _qg(dy))
            else:
                dw, = torch.autograd.grad(_plain_conv(x, wv, stride, pad, upsample), wv, dy)
        return dx, dw, db, None, None, None, None, None, None


def _store(y, g16=True):
    """16-bit storage emulation of an activation (no-op outside compute_dtype(..., storage=True) and for narrow tensors)"""
    if _QDT is None or not _QSTORE or y.shape[1] % 64 != 0:
        return y
    return _StoreQ.apply(y, g16)


def conv_block(x, w, b, stride, pad, act="none", norm="none", norm_args=None, upsample=False, out16=True, g16=True, residual_follows=False):
    """Conv2dBlock.forward (networks.py:365-371): reflect pad -> conv(bias) -> norm -> act.
    ``upsample`` restates the nn.Upsample(scale_factor=2) (nearest) that precedes the two
    5x5 decoder convs (networks.py:256).
    out16 / g16 / residual_follows only matter under compute_dtype(..., storage=True) (see _QSTORE above): out16 = the block's output
    is stored in the 16-bit dtype (when wide); g16 = so is its gradient; residual_follows = the caller adds the ResBlock residual
    first and stores the sum itself (the build fuses the add into the normalisation kernel, one rounding)."""
    co, ci = w.shape[0], w.shape[1]
    store = _QDT is not None and _QSTORE
    s_bwd = ci % 64 == 0 and co % 64 == 0 and not upsample       # the layer's backward runs on the 16-bit-storage kernels
    if _QDT is not None and ci % 32 == 0 and co % 32 == 0:
        ok64 = ci % 64 == 0 and co % 64 == 0
        y = _ConvQ.apply(x, w, b, stride, pad, upsample, True, True, ok64)
        if store and s_bwd and norm != "none" and _QCO16:
            y = _StoreQ.apply(y, True)                   # conv output in front of a normalisation layer, stored by the 16-bit-storage kernels
        y = _norm_act(y, act, norm, norm_args)
    else:
        if upsample:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        if pad > 0:
            x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
        y = _norm_act(F.conv2d(x, w, b, stride=stride), act, norm, norm_args)
    if store and out16 and not residual_follows:
        # a block without normalisation hands its gradient straight to its own backward kernels: 16-bit only when those are the
        # 16-bit-storage kernels
        y = _store(y, g16 and (norm != "none" or s_bwd))
    return y


def _norm_act(y, act, norm, norm_args):
    if norm == "in":
        y = instance_norm(y)
    elif norm == "adain":
        y = adain(y, *norm_args)
    elif norm == "ln":
        y = layer_norm_munit(y, *norm_args)
    else:
        assert norm == "none", norm
    return _act(y, act)


# --------------------------------------------------------------------------------------
# generator (networks.py:112-171, 212-310)
# --------------------------------------------------------------------------------------

def style_encode(P: Params, x, g: dict):
    """StyleEncoder.forward (networks.py:212-228): 7x7, 4x(4x4 s2), GAP, 1x1; norm none, relu."""
    h = conv_block(x, P["enc_style.model.0.conv.weight"], P["enc_style.model.0.conv.bias"], 1, 3, "relu")
    for i in range(1, 5):
        h = conv_block(h, P["enc_style.model.%d.conv.weight" % i], P["enc_style.model.%d.conv.bias" % i], 2, 1, "relu")
    h = h.mean(dim=(2, 3), keepdim=True)  # AdaptiveAvgPool2d(1), networks.py:222
    return F.conv2d(h, P["enc_style.model.6.weight"], P["enc_style.model.6.bias"])


def content_encode(P: Params, x, g: dict):
    """ContentEncoder.forward (networks.py:230-245): 7x7 IN relu, n_downsample x (4x4 s2 IN
    relu), n_res IN ResBlocks (networks.py:297-310)."""
    nd, nr = g["n_downsample"], g["n_res"]
    h = conv_block(x, P["enc_content.model.0.conv.weight"], P["enc_content.model.0.conv.bias"], 1, 3, "relu", "in")
    for i in range(1, nd + 1):
        h = conv_block(h, P["enc_content.model.%d.conv.weight" % i], P["enc_content.model.%d.conv.bias" % i], 2, 1, "relu", "in")
    for r in range(nr):
        pre = "enc_content.model.%d.model.%d.model." % (nd + 1, r)
        t = conv_block(h, P[pre + "0.conv.weight"], P[pre + "0.conv.bias"], 1, 1, "relu", "in")
        t = conv_block(t, P[pre + "1.conv.weight"], P[pre + "1.conv.bias"], 1, 1, "none", "in", residual_follows=True)
        h = _store(t + h)  # networks.py:309
    return h


def mlp(P: Params, style):
    """MLP.forward (networks.py:280-292)."""
    h = style.reshape(style.shape[0], -1)
    h = F.relu(F.linear(h, P["mlp.model.0.fc.weight"], P["mlp.model.0.fc.bias"]))
    h = F.relu(F.linear(h, P["mlp.model.1.fc.weight"], P["mlp.model.1.fc.bias"]))
    return F.linear(h, P["mlp.model.2.fc.weight"], P["mlp.model.2.fc.bias"])


def decode(P: Params, content, style, g: dict):
    """AdaINGen.decode (networks.py:147-163) + Decoder.forward (networks.py:247-264).
    AdaIN layer j takes columns [2Cj, 2Cj+C) of the MLP output as BIAS and
    [2Cj+C, 2Cj+2C) as WEIGHT (assign_adain_params, networks.py:154-163)."""
    nd, nr = g["n_downsample"], g["n_res"]
    ap = mlp(P, style)
    C = content.shape[1]
    h = content
    j = 0
    for r in range(nr):
        pre = "dec.model.0.model.%d.model." % r
        b0, w0 = ap[:, 2 * C * j: 2 * C * j + C], ap[:, 2 * C * j + C: 2 * C * (j + 1)]
        j += 1
        b1, w1 = ap[:, 2 * C * j: 2 * C * j + C], ap[:, 2 * C * j + C: 2 * C * (j + 1)]
        j += 1
        t = conv_block(h, P[pre + "0.conv.weight"], P[pre + "0.conv.bias"], 1, 1, "relu", "adain", (w0, b0))
        t = conv_block(t, P[pre + "1.conv.weight"], P[pre + "1.conv.bias"], 1, 1, "none", "adain", (w1, b1), residual_follows=True)
        h = _store(t + h, g16=(r + 1 < nr))      # the last block feeds a sub-pixel layer, whose input-gradient kernels write fp32
    idx = 2
    for i in range(nd):
        pre = "dec.model.%d." % idx
        # (the last of these feeds the fp32 7x7 output layer: stored fp32; the others feed the next sub-pixel layer: fp32 gradient)
        h = conv_block(h, P[pre + "conv.weight"], P[pre + "conv.bias"], 1, 2, "relu", "ln",
                       (P[pre + "norm.gamma"], P[pre + "norm.beta"]), upsample=True, out16=(i + 1 < nd), g16=False)
        idx += 2
    pre = "dec.model.%d." % (idx - 1)
    return conv_block(h, P[pre + "conv.weight"], P[pre + "conv.bias"], 1, 3, "tanh")


def gen_encode(P: Params, x, g: dict):
    """AdaINGen.encode (networks.py:141-145) -> (content, style)."""
    return content_encode(P, x, g), style_encode(P, x, g)


# --------------------------------------------------------------------------------------
# discriminator (networks.py:21-106), lsgan branch only
# --------------------------------------------------------------------------------------

def avgpool3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False) (networks.py:33)."""
    return F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)


def dis_forward(P: Params, x, dcfg: dict) -> List[torch.Tensor]:
    """MsImageDis.forward (networks.py:50-57): per scale 4x(4x4 s2 reflect, lrelu) + 1x1."""
    outs = []
    nl = dcfg["n_layer"]
    for s in range(dcfg["num_scales"]):
        h = x
        for i in range(nl):
            h = conv_block(h, P["cnns.%d.%d.conv.weight" % (s, i)], P["cnns.%d.%d.conv.bias" % (s, i)], 2, 1, "lrelu", out16=(i + 1 < nl))
        outs.append(F.conv2d(h, P["cnns.%d.%d.weight" % (s, nl)], P["cnns.%d.%d.bias" % (s, nl)]))
        x = avgpool3s2(x)
    return outs


def lsgan(outs, target: float):
    """sum over scales of mean((out - target)^2) (networks.py:67,83,98)."""
    loss = 0
    for o in outs:
        loss = loss + torch.mean((o - target) ** 2)
    return loss


# --------------------------------------------------------------------------------------
# trainer-level pieces (trainer.py)
# --------------------------------------------------------------------------------------

def focus_translation(fg, bg, focus):
    """trainer.py:85-88."""
    m = ((focus + 1) / 2).repeat(1, 3, 1, 1)
    return fg * m + bg * (1 - m)


def focus_losses(focus, hp):
    """trainer.py:146-158 for one mask: returns (size, digit)."""
    m = (focus + 1) / 2
    size = (F.relu(torch.sum(m - hp["focus_upper"])) ** 2) * hp["focus_delta"] + \
           (F.relu(torch.sum(hp["focus_lower"] - m)) ** 2) * hp["focus_delta"]
    digit = torch.sum(1 / (_abs(m - 0.5) + hp["focus_epsilon"]))
    return size, digit


def generator_forward(G_AB: Params, G_BA: Params, x_a, x_b, z, hp, with_recon: bool):
    """The shared generator forward of gen_update / dis_update (trainer.py:103-133 /
    258-277): the focus branch (focus_loss > 0: decoder output = image + focus mask, trainer.py:107-116,126-128) or the non-focus
    branch (focus_loss == 0, gen.output_dim 3: the decoder output is the image, trainer.py:117-121,129-130).
    ``z`` = (z_1, z_2, z_3), each (B, style_dim, 1, 1)."""
    g = hp["gen"]
    z1, z2, z3 = z
    alpha = hp["alpha"]
    focus = hp["focus_loss"] > 0
    c1 = content_encode(G_AB, x_a, g)
    c2, s2 = gen_encode(G_BA, x_a, g)
    out = {}
    fB = fA = fA2 = None
    if focus:
        xB, fB = decode(G_AB, c1, z1, g).split(3, 1)
        xA, fA = decode(G_BA, c2, alpha * z2, g).split(3, 1)
        xB = focus_translation(xB, x_a, fB)
        xA = focus_translation(xA, x_a, fA)
    else:
        xB = decode(G_AB, c1, z1, g)
        xA = decode(G_BA, c2, alpha * z2, g)
    if with_recon:
        c4, s4 = gen_encode(G_AB, x_b, g)
        out["x_A_recon"] = decode(G_BA, c2, s2, g)[:, :3]
        out["x_B_recon"] = decode(G_AB, c4, s4, g)[:, :3]
    c3 = content_encode(G_BA, xB, g)
    if focus:
        xA2, fA2 = decode(G_BA, c3, z3, g).split(3, 1)
        xA2 = focus_translation(xA2, xB, fA2)
    else:
        xA2 = decode(G_BA, c3, z3, g)
    out.update(x_B_fake=xB, x_A_fake=xA, x_A2_fake=xA2, f_B=fB, f_A=fA, f_A2=fA2,
               pair_A1=torch.cat((x_a, xA), 1), pair_A2=torch.cat((x_a, xA2), 1),
               c_1=c1, c_2=c2, c_3=c3, s_2=s2)
    return out


def gen_losses(nets: Dict[str, Params], x_a, x_b, z, hp):
    """gen_update's loss graph (trainer.py:103-165).  Returns (total, dict of the 12 loss
    scalars the reference exposes as ``loss_*`` attributes, intermediates)."""
    d = hp["dis"]
    fw = generator_forward(nets["gen_AB"], nets["gen_BA"], x_a, x_b, z, hp, with_recon=True)
    L = OrderedDict()
    L["loss_gen_adv_A"] = (lsgan(dis_forward(nets["dis_A"], fw["x_A_fake"], d), 1.0) +
                           lsgan(dis_forward(nets["dis_A"], fw["x_A2_fake"], d), 1.0)) * 0.5
    L["loss_gen_adv_B"] = lsgan(dis_forward(nets["dis_B"], fw["x_B_fake"], d), 1.0)
    # calc_gen_d2_loss(pair_A1, pair_A2): pair_A1 -> 1, pair_A2 -> 0 (networks.py:91-98)
    L["loss_gen_adv_2"] = lsgan(dis_forward(nets["dis_2"], fw["pair_A1"], d), 1.0) + \
                          lsgan(dis_forward(nets["dis_2"], fw["pair_A2"], d), 0.0)
    total = hp["gan_w"] * L["loss_gen_adv_A"] + hp["gan_w"] * L["loss_gen_adv_B"] + \
            hp["gan_cw"] * L["loss_gen_adv_2"]
    if hp["focus_loss"] > 0:      # trainer.py:145-161 (the non-focus branch sets none of the six focus attributes)
        sB, dB = focus_losses(fw["f_B"], hp)
        sA, dA = focus_losses(fw["f_A"], hp)
        sA2, dA2 = focus_losses(fw["f_A2"], hp)
        L["loss_gen_focus_B_size"], L["loss_gen_focus_B_digit"] = sB, dB
        L["loss_gen_focus_A_size"], L["loss_gen_focus_A_digit"] = sA, dA
        L["loss_gen_focus_A2_size"], L["loss_gen_focus_A2_digit"] = sA2, dA2
        B, _, H, W = x_a.shape
        total = total + hp["focus_loss"] * (sB + dB + sA + dA + sA2 + dA2) / H / W / B / 3
    L["loss_idt_A"] = torch.mean(_abs(fw["x_A_recon"] - x_a))
    L["loss_idt_B"] = torch.mean(_abs(fw["x_B_recon"] - x_b))
    total = total + hp["recon_x_w"] * L["loss_idt_A"] + hp["recon_x_w"] * L["loss_idt_B"]
    L["loss_gen_total"] = total
    return total, L, fw


def dis_losses(nets: Dict[str, Params], x_a, x_b, z, hp):
    """dis_update's loss graph (trainer.py:258-290).  calc_dis_loss(fake, real) puts fake->0,
    real->1 (networks.py:60-67); for dis_2 'fake' = pair_A1 and 'real' = pair_A2."""
    d = hp["dis"]
    fw = generator_forward(nets["gen_AB"], nets["gen_BA"], x_a, x_b, z, hp, with_recon=False)
    L = OrderedDict()

    def dl(D, fake, real):
        return lsgan(dis_forward(D, fake, d), 0.0) + lsgan(dis_forward(D, real, d), 1.0)

    L["loss_dis_A"] = (dl(nets["dis_A"], fw["x_A_fake"], x_a) + dl(nets["dis_A"], fw["x_A2_fake"], x_a)) * 0.5
    L["loss_dis_B"] = dl(nets["dis_B"], fw["x_B_fake"], x_b)
    L["loss_dis_2"] = dl(nets["dis_2"], fw["pair_A1"], fw["pair_A2"])
    total = hp["gan_w"] * L["loss_dis_A"] + hp["gan_w"] * L["loss_dis_B"] + hp["gan_cw"] * L["loss_dis_2"]
    L["loss_dis_total"] = total
    return total, L, fw


class AdamState:
    """torch.optim.Adam semantics as configured at trainer.py:39-42: classic L2 weight decay
    added to the gradient, bias-corrected moments, eps outside the sqrt."""

    def __init__(self, params: List[torch.Tensor], lr, beta1, beta2, weight_decay, eps=1e-8):
        self.params = params
        self.lr, self.b1, self.b2, self.wd, self.eps = lr, beta1, beta2, weight_decay, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads: List[torch.Tensor], lr: Optional[float] = None):
        lr = self.lr if lr is None else lr
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            g = g + self.wd * p
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-lr / bc1)


def step_lr(base_lr: float, gamma: float, step_size: int, n_calls: int) -> float:
    """StepLR (utils.py:263-271): lr after ``n_calls`` scheduler.step() calls."""
    return base_lr * gamma ** (n_calls // step_size)


class OracleTrainer:
    """Explicit-z CPU counterpart of aclgan_Trainer (trainer.py:14-331) built from the
    functions above.  Used by tests as the checker and by bench.py's cpu_baseline leg."""

    NETS = ("gen_AB", "gen_BA", "dis_A", "dis_B", "dis_2")

    def __init__(self, hp: dict, nets: Optional[Dict[str, Params]] = None, seed: int = 0):
        self.hp = hp
        g = torch.Generator().manual_seed(seed)
        if nets is None:
            gs = gen_param_shapes(hp["input_dim_a"], hp["gen"])
            nets = {
                "gen_AB": init_params(gs, hp.get("init", "kaiming"), g),
                "gen_BA": init_params(gs, hp.get("init", "kaiming"), g),
                "dis_A": init_params(dis_param_shapes(hp["input_dim_a"], hp["dis"]), "gaussian", g),
                "dis_B": init_params(dis_param_shapes(hp["input_dim_a"], hp["dis"]), "gaussian", g),
                "dis_2": init_params(dis_param_shapes(hp["input_dim_b"], hp["dis"]), "gaussian", g),
            }
        self.nets = {k: OrderedDict((n, t.detach().clone().requires_grad_(True)) for n, t in v.items())
                     for k, v in nets.items()}
        gp = list(self.nets["gen_AB"].values()) + list(self.nets["gen_BA"].values())
        dp = list(self.nets["dis_A"].values()) + list(self.nets["dis_B"].values()) + list(self.nets["dis_2"].values())
        self.gen_opt = AdamState(gp, hp["lr"], hp["beta1"], hp["beta2"], hp["weight_decay"])
        self.dis_opt = AdamState(dp, hp["lr"], hp["beta1"], hp["beta2"], hp["weight_decay"])
        self.sched_calls = 0
        self.losses: Dict[str, float] = {}

    def _lr(self):
        if self.hp.get("lr_policy", "constant") == "step":
            return step_lr(self.hp["lr"], self.hp["gamma"], self.hp["step_size"], self.sched_calls)
        return self.hp["lr"]

    def _zero(self):
        for n in self.NETS:
            for t in self.nets[n].values():
                t.grad = None

    def gen_update(self, x_a, x_b, z, apply: bool = True):
        self._zero()
        total, L, fw = gen_losses(self.nets, x_a, x_b, z, self.hp)
        total.backward()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.gen_opt.params]
        if apply:
            self.gen_opt.step(grads, self._lr())
        self.losses.update({k: float(v.detach()) for k, v in L.items()})
        return L, fw, grads

    def dis_update(self, x_a, x_b, z, apply: bool = True):
        self._zero()
        total, L, fw = dis_losses(self.nets, x_a, x_b, z, self.hp)
        total.backward()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.dis_opt.params]
        if apply:
            self.dis_opt.step(grads, self._lr())
        self.losses.update({k: float(v.detach()) for k, v in L.items()})
        return L, fw, grads

    def update_learning_rate(self):
        self.sched_calls += 1
