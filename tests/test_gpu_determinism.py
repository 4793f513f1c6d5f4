"""Deterministic mode (aclgan_set_deterministic / aclgan_Trainer(deterministic=True)): every reduction that the default plan
combines with fp32 atomics takes an ordered path, so results are reproducible BIT FOR BIT run to run -- and still the same
numbers (oracle parity within the usual tolerances).

The reference has no such mode (train.py:29 sets cudnn.benchmark = True and never calls torch.use_deterministic_algorithms);
this is the SURVEY section 7 item "split-K + deterministic reduction" taken to the whole step.

  * operator level: input and weight gradients of every kernel family that uses atomics by default (reflection halo, small-grid
    split-K, the sub-pixel ring, thin 7x7 layers, 3- / 6-channel first layers, the Cout = 1 head, reduced widths), fp32 and
    the 16-bit dgrad: oracle parity + two calls agree bitwise;
  * step level: dis_update + gen_update of two trainers built from the same state agree bitwise in all 16 losses and every
    gradient tensor (fp32 and bf16), and agree with the default mode to the tolerance of the full-size parity tests.
"""
import ctypes as C

import pytest
import torch

from oracle import aclgan_oracle as O

pytestmark = pytest.mark.gpu

TOL = 2e-4


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib
    return _lib


@pytest.fixture()
def det(L):
    """deterministic mode for one test; the process-wide switch is restored afterwards"""
    prev = L.lib.aclgan_get_deterministic()      # (1 when the whole suite runs under ACLGAN_DETERMINISTIC=1)
    L.check(L.lib.aclgan_set_deterministic(1))
    assert L.lib.aclgan_get_deterministic() == 1
    yield
    L.check(L.lib.aclgan_set_deterministic(prev))


# (B, Hi, Wi, Ci, Co, k, s, p, up): one per default-mode atomics site
DET_CASES = [
    (2, 8, 8, 256, 256, 3, 1, 1, 0),      # ResBlock conv: Winograd interior + halo launch  -> padded grid + fold
    (2, 10, 12, 64, 64, 3, 1, 1, 0),      # direct interior + halo
    (2, 8, 8, 256, 128, 5, 1, 2, 1),      # sub-pixel layer: ring contributions via atomics  -> plain upsample+5x5 dgrad
    (2, 9, 13, 32, 48, 5, 1, 2, 1),       # sub-pixel layer without the ordered-slice wgrad kernel (Cin 32, Cout 48)
    (2, 8, 8, 256, 512, 4, 2, 1, 0),      # late discriminator conv: small grid, split-K dgrad
    (2, 32, 32, 6, 64, 4, 2, 1, 0),       # first discriminator layer, 6 channels: general wgrad kernel, column-sum bias gradient
    (2, 32, 32, 3, 64, 4, 2, 1, 0),       # same, 3 channels
    (2, 4, 4, 512, 1, 1, 1, 0, 0),        # discriminator head, Cout 1
    (2, 16, 20, 3, 64, 7, 1, 3, 0),       # thin 7x7 encoder layer (4x4x1 MFMA kernel, per-workgroup partials)
    (2, 16, 16, 64, 4, 7, 1, 3, 0),       # thin 7x7 output layer, Cout 4
    (3, 12, 20, 16, 8, 4, 2, 1, 0),       # reduced width: atomics weight-gradient kernel with pixel slices
    (5, 9, 7, 12, 20, 3, 1, 1, 0),        # nothing a multiple of anything
]


def _tensors(case, seed):
    B, Hi, Wi, Ci, Co, k, s, p, up = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, Hi, Wi, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    b = torch.randn(Co, generator=g) * 0.1
    return x, w, b


@pytest.mark.parametrize("case", DET_CASES)
def test_conv_backward_reproducible(L, det, case):
    from gpu_util import conv_desc, gpu_conv_dgrad, gpu_conv_wgrad, nhwc, nchw, ohwi, rel_err
    B, Hi, Wi, Ci, Co, k, s, p, up = case
    x, w, b = _tensors(case, 3)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y = O.conv_block(x, w, b, s, p, "none", upsample=bool(up))
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(4))
    y.backward(dy)
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, "none")
    xg, wg, dyg = nhwc(x.detach()).cuda(), ohwi(w.detach()).cuda(), nhwc(dy).cuda()
    dx1 = gpu_conv_dgrad(L, d, dyg, wg)
    dx2 = gpu_conv_dgrad(L, d, dyg, wg)
    assert rel_err(nchw(dx1), x.grad) < TOL
    assert torch.equal(dx1, dx2)
    dw1, db1 = gpu_conv_wgrad(L, d, xg, dyg)
    dw2, db2 = gpu_conv_wgrad(L, d, xg, dyg)
    assert rel_err(dw1, ohwi(w.grad)) < TOL and rel_err(db1, b.grad) < TOL
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    # accumulate mode of the input gradient goes through the same ordered path
    base = torch.randn(B, Hi, Wi, Ci, generator=torch.Generator().manual_seed(6))
    acc = gpu_conv_dgrad(L, d, dyg, wg, accumulate_into=base.clone().cuda())
    assert rel_err(nchw(acc).cpu() - nchw(base), x.grad) < 5 * TOL


def test_scratchless_calls_refuse_where_they_would_need_atomics(L, det):
    from gpu_util import conv_desc, nhwc
    d = conv_desc(L, 2, 32, 32, 6, 64, 4, 2, 1, 0, "none")
    x = torch.randn(2, 32, 32, 6, device="cuda"); dy = torch.randn(2, 16, 16, 64, device="cuda")
    dw = torch.zeros(64, 4, 4, 6, device="cuda"); db = torch.zeros(64, device="cuda")
    rc = L.lib.aclgan_conv2d_wgrad(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.stream_ptr())
    assert rc != 0 and b"deterministic" in L.lib.aclgan_last_error()


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 8, 8, 256, 256, 3, 1, 1, 0), (2, 8, 8, 256, 128, 5, 1, 2, 1), (2, 8, 8, 256, 512, 4, 2, 1, 0)])
def test_conv_dgrad16_reproducible(L, det, case, dt):
    from gpu_util import conv_desc, nhwc, nchw, ohwi, rel_err
    B, Hi, Wi, Ci, Co, k, s, p, up = case
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dt]
    x, w, b = _tensors(case, 8)
    xr = x.double().requires_grad_(True)
    assert L.lib.aclgan_conv16_eligible(C.byref(conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, "none")), 1) == 1
    y = O.conv_block(xr, w.to(tdt).double(), b.double(), s, p, "none", upsample=bool(up))
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(dy.to(tdt).double())
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, "none")
    wg, dyg = ohwi(w).cuda(), nhwc(dy).cuda()
    w16t = torch.empty(wg.numel(), dtype=torch.int16, device="cuda")
    L.check(L.lib.aclgan_pack_weights16(L.ptr(wg), None, L.ptr(w16t), Co, k * k, Ci, L.DTYPE[dt], L.stream_ptr()), "pack_weights16")
    nb = L.lib.aclgan_conv2d_dgrad16_scratch_bytes(C.byref(d))
    assert nb >= 4 * B * ((Hi << up) + 2 * p) * ((Wi << up) + 2 * p) * Ci     # the padded-grid gradient
    scr = torch.empty(nb // 4 + 64, device="cuda")
    outs = []
    for _ in range(2):
        dx = torch.full((B, Hi, Wi, Ci), float("nan"), device="cuda")
        L.check(L.lib.aclgan_conv2d_dgrad16(C.byref(d), L.DTYPE[dt], L.ptr(dyg), L.ptr(wg), L.ptr(w16t), L.ptr(dx), 0, L.ptr(scr),
                                            L.stream_ptr()), "conv2d_dgrad16")
        outs.append(dx)
    # exact-arithmetic check (operands rounded on both sides): the plain upsample+5x5 gradient merges no filters, so it holds
    # for the sub-pixel layers too
    assert rel_err(nchw(outs[0]), xr.grad) < TOL
    assert torch.equal(outs[0], outs[1])


def _step(T, cfg, nets, x_a, x_b, z, dtype, deterministic):
    out = {}
    for which in ("dis", "gen"):
        tr = T.aclgan_Trainer(cfg, compute_dtype=dtype, deterministic=deterministic)
        assert tr.deterministic == bool(deterministic)
        for name in O.OracleTrainer.NETS:
            getattr(tr, name).load_state_dict(nets[name], strict=False)
        if which == "dis":
            tr.dis_update(x_a, x_b, cfg, z=z[:3]); names = ("dis_A", "dis_B", "dis_2")
        else:
            tr.gen_update(x_a, x_b, cfg, z=z[3:]); names = ("gen_AB", "gen_BA")
        torch.cuda.synchronize()
        from aclgan_amd import _lib
        lnames = _lib.LOSS_NAMES[12:16] if which == "dis" else _lib.LOSS_NAMES[0:12]
        out[which] = ({n: float(getattr(tr, n)) for n in lnames},
                      {(n, k): g.contiguous().clone() for n in names for k, g in getattr(tr, n).named_grads()})
    return out


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_step_reproducible_bit_for_bit(L, dtype):
    from aclgan_amd import trainer as T
    cfg = O.default_config()
    cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5
    nets = O.test_nets(cfg, 0)
    g = torch.Generator().manual_seed(21)
    B, S = 2, 128
    x_a = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    x_b = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    z = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(6)]
    prev = L.lib.aclgan_get_deterministic()
    try:
        a = _step(T, cfg, nets, x_a, x_b, z, dtype, True)
        b = _step(T, cfg, nets, x_a, x_b, z, dtype, True)
        ref = _step(T, cfg, nets, x_a, x_b, z, dtype, False)
    finally:
        L.check(L.lib.aclgan_set_deterministic(prev))
    for which in ("dis", "gen"):
        la, ga = a[which]; lb, gb = b[which]; lr, gr = ref[which]
        assert la and ga
        assert la == lb, (which, {k: (la[k], lb[k]) for k in la if la[k] != lb[k]})
        diff = [k for k in ga if not torch.equal(ga[k], gb[k])]
        assert not diff, (which, diff[:8])
        # same numbers as the default plan (different summation order only; ReLU mask flips bound the tail like in
        # tests/test_gpu_fullsize.py)
        gmax = max(float(t.double().norm()) for t in gr.values())
        worst = max(((ga[k].double() - gr[k].double()).norm().item() / (gr[k].double().norm().item() + 1e-5 * gmax / 1e-2), k) for k in ga)
        assert worst[0] <= (1e-2 if dtype == "fp32" else 0.3), worst
        for k in la:
            assert abs(la[k] - lr[k]) <= (5e-3 if k.endswith("_size") else 1e-4 if dtype == "fp32" else 2e-3) * max(1e-3, abs(lr[k])), (k, la[k], lr[k])


_ORACLE_CACHE = {}


def chained_report(seed, steps=3):
    """three chained (dis_update, gen_update, update_learning_rate) iterations of the deterministic HIP trainer and of the fp32 oracle
    from one state: per network (max |dp| / lr, share of elements within 0.05 lr, relative L2 error of the update p - p0), plus the trainer"""
    from aclgan_amd.trainer import aclgan_Trainer
    cfg = O.default_config()
    cfg["gen"].update(dim=16, mlp_dim=32, n_res=2); cfg["dis"].update(dim=16)
    cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5
    nets = O.test_nets(cfg, seed)
    g = torch.Generator().manual_seed(71 + seed)
    x_a = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    x_b = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    zs = [[torch.randn(2, 8, 1, 1, generator=g) for _ in range(6)] for _ in range(steps)]

    def run():
        tr = aclgan_Trainer(cfg, deterministic=True)
        for name in O.OracleTrainer.NETS:
            getattr(tr, name).load_state_dict(nets[name], strict=False)
        for it in range(steps):
            tr.dis_update(x_a, x_b, cfg, z=zs[it][:3]); tr.gen_update(x_a, x_b, cfg, z=zs[it][3:]); tr.update_learning_rate()
        torch.cuda.synchronize()
        return tr
    tr = run()
    p0 = {n: {k: v.clone() for k, v in nets[n].items()} for n in O.OracleTrainer.NETS}
    if (seed, steps) not in _ORACLE_CACHE:      # (the oracle's three CPU steps are the slow part: shared by the convolution paths)
        orc = O.OracleTrainer(cfg, nets=nets)
        for it in range(steps):
            orc.dis_update(x_a, x_b, zs[it][:3]); orc.gen_update(x_a, x_b, zs[it][3:]); orc.update_learning_rate()
        _ORACLE_CACHE[(seed, steps)] = orc
    orc = _ORACLE_CACHE[(seed, steps)]
    lr = cfg["lr"]
    report = {}
    for n in O.OracleTrainer.NETS:
        num = den = 0.0
        worst, tight, total = 0.0, 0, 0
        for k, p in getattr(tr, n).named_parameters():
            got = p.detach().cpu().double(); ref = orc.nets[n][k].detach().double(); start = p0[n][k].double()
            d = (got - ref).abs()
            worst = max(worst, d.max().item())
            tight += int((d <= 0.05 * lr).sum()); total += d.numel()
            num += ((got - start) - (ref - start)).pow(2).sum().item(); den += (ref - start).pow(2).sum().item()
        report[n] = (worst / lr, tight / total, (num / max(den, 1e-300)) ** 0.5)
    return report, tr, run


# the three convolution paths of the 3x3 layers (tuning switch "wino_fused": 0 = three-launch Winograd pipeline, 2 = the one-launch kernel
# wherever eligible; the direct kernels are ACLGAN_NOWINO, an environment switch)
CHAIN_SEEDS = (6, 16, 26)


@pytest.mark.parametrize("fused", [0, 2])
def test_three_chained_steps_parameters_match_oracle_elementwise(L, det, fused):
    """Deterministic mode, three chained (dis_update, gen_update, update_learning_rate) iterations from the same state as the fp32
    oracle: every parameter compared ELEMENTWISE after the third Adam step (not by norm).

    What can and cannot agree: Adam's update is lr * m_hat / (sqrt(v_hat) + eps), i.e. ~ lr * sign(g) on the first step -- an
    element whose gradient sits within fp32 summation noise of zero can come out with the other sign (a 2 lr difference per
    step).  So: (i) hard bound: no element differs by more than 2 lr x 3 steps; (ii) the update as a whole (p - p0) agrees in
    relative L2 per network; (iii) the bulk of the elements agrees tightly.

    Round 5 (advisor, round 4): the bounds of (ii) and (iii) are held PER CONVOLUTION PATH and over SEVERAL fixtures -- one 64 x 64
    fixture and one seed cannot tell a systematic error of a kernel from the lottery of which near-zero pre-activations flip.  The
    three-launch pipeline keeps the round-3 bounds on the round-3 fixture; both paths must hold the mean over CHAIN_SEEDS, and the
    fused path may not be systematically worse than the pipeline (mean over the seeds, same fixtures)."""
    prev = C.c_int(); prev_s2 = C.c_int()
    L.check(L.lib.aclgan_tuning(b"wino_fused", fused, C.byref(prev)), "tuning")
    # Round 6: mode 2 would also force the 4x4 stride-2 layers into the fused kernel (a THIRD path on this 64 x 64 fixture, measured: gen_BA share
    # 0.787 over the three seeds against 0.930 / 0.992 for the two paths this test compares -- more Winograd-level forward noise, more flipped
    # masks).  This test keeps comparing the two 3x3 paths of round 5; that the stride-2 path's kernels are exact is shown where the lottery
    # is taken out: tests/test_gpu_maskfrozen.py runs it forced, masks frozen, every gradient tensor to 1e-3.
    L.check(L.lib.aclgan_tuning(b"wino_s2k4", 0, C.byref(prev_s2)), "tuning")
    try:
        reports = {}
        for seed in CHAIN_SEEDS:
            report, tr, run = chained_report(seed)
            reports[seed] = report
            print("wino_fused=%d seed %d: 3 chained deterministic steps vs the fp32 oracle, per network: (max |dp| / lr, share of elements within 0.05 lr, "
                  "relative L2 of the update)" % (fused, seed), {n: ("%.2f" % a, "%.4f" % b, "%.2e" % c) for n, (a, b, c) in report.items()})
            for n, (a, b, c) in report.items():
                assert a <= 6.05, (seed, n, a)           # (i): 2 lr per step
            if seed == CHAIN_SEEDS[0]:
                # and bit-reproducible: a second trainer from the same state lands on the same bits
                tr2 = run()
                assert torch.equal(tr._param[0], tr2._param[0]) and torch.equal(tr._param[1], tr2._param[1])
        mean = {n: tuple(sum(reports[s][n][i] for s in CHAIN_SEEDS) / len(CHAIN_SEEDS) for i in range(3)) for n in reports[CHAIN_SEEDS[0]]}
        print("wino_fused=%d mean over seeds %s:" % (fused, CHAIN_SEEDS), {n: ("%.2f" % a, "%.4f" % b, "%.2e" % c) for n, (a, b, c) in mean.items()})
        for n, (a, b, c) in mean.items():
            gen = n.startswith("gen")
            assert b >= (CHAIN_BOUNDS["gen_share"] if gen else CHAIN_BOUNDS["dis_share"]), (fused, n, b)
            assert c <= (CHAIN_BOUNDS["gen_update"] if gen else CHAIN_BOUNDS["dis_update"]), (fused, n, c)
        if fused == 0:
            # the round-3 gate on the round-3 fixture (seed 6), unchanged, for the path it was measured on
            for n, (a, b, c) in reports[6].items():
                assert b >= (0.80 if n.startswith("gen") else 0.999), (n, b)
                assert c <= (0.1 if n.startswith("gen") else 2e-3), (n, c)
        CHAIN_MEANS[fused] = mean
        if 0 in CHAIN_MEANS and 2 in CHAIN_MEANS:
            # the one-launch kernel against the pipeline on the same fixtures: not systematically worse (a ratio of means; measured values
            # in the test output / profiles/r05_gpu_tests.log)
            for n in mean:
                share0, share2 = CHAIN_MEANS[0][n][1], CHAIN_MEANS[2][n][1]
                err0, err2 = CHAIN_MEANS[0][n][2], CHAIN_MEANS[2][n][2]
                assert share2 >= share0 - CHAIN_BOUNDS["share_gap"], (n, share0, share2)
                assert err2 <= max(CHAIN_BOUNDS["err_ratio"] * err0, CHAIN_BOUNDS["err_floor"]), (n, err0, err2)
    finally:
        L.check(L.lib.aclgan_tuning(b"wino_fused", prev.value, None), "tuning")
        L.check(L.lib.aclgan_tuning(b"wino_s2k4", prev_s2.value, None), "tuning")


# Measured on the MI355X (round 5, profiles/r05_experiments.md section 1): mean over the three fixtures, pipeline / fused --
#   share within 0.05 lr: gen_AB 0.881 / 0.860, gen_BA 0.930 / 0.992, discriminators >= 0.9998 / 0.9998;
#   update error: gen_AB 2.5e-2 / 3.3e-2, gen_BA 1.6e-2 / 5.8e-3, dis_2 7.3e-4 / 1.2e-3, dis_A / dis_B <= 1e-4.
# Per fixture the two paths trade places (seed 6: gen_AB 0.863 / 0.924; seed 26: 0.862 / 0.742): which near-zero pre-activations flip is
# drawn by the rounding pattern, not by the path.  Bounds = those means with ~7 % (shares) / ~2x (errors) of room.
CHAIN_BOUNDS = {"gen_share": 0.80, "dis_share": 0.999, "gen_update": 0.06, "dis_update": 2.5e-3, "share_gap": 0.06, "err_ratio": 2.0, "err_floor": 2e-3}
CHAIN_MEANS = {}
