"""Parity of the HIP path against the CPU oracle AT THE SIZES THAT ARE BENCHMARKED (BASELINE.json
configs[1]: 256x256 B=8 and configs[3]: 512x512 B=4, full-width male2female architecture).

The small-size tests (tests/test_gpu_step.py) never reach the launch shapes that dominate the bench: the
512-workgroup 128x128-tile forward, the 36x42 split wgrad, the XCD tile remap, the sub-pixel interior at
128^2 / 256^2 / 512^2.  Here the same weights / inputs / z go through the HIP library and through the fp32
oracle (oracle/aclgan_oracle.py, pinned to the reference by tests/golden) and are compared directly:

  * forward tensors (contents, styles, decoder outputs after the focus blend, the consistency pass,
    discriminator maps)                                       <= 1e-4 rel (north-star tolerance: 1e-3)
  * the 16 reported losses                                    <= 1e-4 rel ('size' losses 2e-2)
  * one dis_update + one gen_update: every gradient tensor    <= 1e-2 relative L2 (smooth fixture:
    focus_epsilon 0.5, see tests/golden/make_golden.py for why the default 0.01 is ill-conditioned)

Oracle cost on the GPU box's host cores: ~4 s per 256^2 sample-step; each benchmarked shape runs its oracle update ONCE for both of its
tests (round 4), so the whole file is ~2.5 minutes.
"""
import os

import pytest
import torch

from oracle import aclgan_oracle as O

pytestmark = pytest.mark.gpu

GTOL = 1e-2          # per-tensor relative L2 of the gradients; measured worst 2.8e-3 with the direct 3x3 kernels (ACLGAN_NOWINO=1),
                     # 4.7e-3 with the Winograd F(4x4,3x3) ResBlock path (its 1.3e-5 per-conv noise flips a few more ReLU masks)
FTOL = 1e-4          # forward tensors, max-abs relative (north star: 1e-3; measured worst 1.6e-5)
LTOL = 1e-4          # losses (measured worst 4e-7)


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    import aclgan_amd  # noqa: F401
    from aclgan_amd import trainer
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    return trainer


def _make(T, cfg, nets):
    tr = T.aclgan_Trainer(cfg)
    for name in O.OracleTrainer.NETS:
        getattr(tr, name).load_state_dict(nets[name], strict=False)
    return tr


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _inputs(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    x_a = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    x_b = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    z = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(6)]
    return x_a, x_b, z


_ORACLE = {}


def _oracle_step(B, S):
    """ONE fp32 oracle dis_update + gen_update (autograd, no Adam) per benchmarked shape, shared by the forward / loss test and the gradient
    test of that shape (round 3 ran the 256x256 B=8 and 512x512 B=4 oracle passes twice: most of this file's wall time on the GPU box).
    Smooth fixture (focus_epsilon 0.5, see tests/golden/make_golden.py) for the gradients; the forward tensors do not depend on it and
    the loss VALUES at the default epsilon are recomputed from the same forward by `_default_eps_losses`."""
    key = (B, S)
    if key not in _ORACLE:
        cfg = O.default_config()
        cfg["display_size"] = 1
        cfg["focus_epsilon"] = 0.5
        nets = O.test_nets(cfg, 0)
        x_a, x_b, z = _inputs(B, S, 12)
        od = O.OracleTrainer(cfg, nets=nets); od.dis_update(x_a, x_b, z[:3], apply=False)
        og = O.OracleTrainer(cfg, nets=nets); _, fw, _ = og.gen_update(x_a, x_b, z[3:], apply=False)
        fw = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in fw.items()}
        with torch.no_grad():
            dA = [t.detach() for t in O.dis_forward(nets["dis_A"], fw["x_A_fake"], cfg["dis"])]
            d2 = [t.detach() for t in O.dis_forward(nets["dis_2"], fw["pair_A2"], cfg["dis"])]
        _ORACLE.clear()      # one shape resident at a time (the 512x512 graph is tens of GB of host memory)
        _ORACLE[key] = dict(cfg=cfg, nets=nets, x_a=x_a, x_b=x_b, z=z, od=od, og=og, fw=fw, dA=dA, d2=d2)
    return _ORACLE[key]


def _default_eps_losses(o):
    """the generator losses of the same forward at the shipped focus_epsilon 0.01 (trainer.py:151: only the three digit losses and the
    total depend on it)"""
    cfg = dict(o["cfg"]); cfg["focus_epsilon"] = O.default_config()["focus_epsilon"]
    fw, og = o["fw"], o["og"]
    L = dict(og.losses)
    B, _, H, W = o["x_a"].shape
    with torch.no_grad():
        tot = cfg["gan_w"] * L["loss_gen_adv_A"] + cfg["gan_w"] * L["loss_gen_adv_B"] + cfg["gan_cw"] * L["loss_gen_adv_2"]
        fsum = 0.0
        for nm, key in (("B", "f_B"), ("A", "f_A"), ("A2", "f_A2")):
            sz, dg = O.focus_losses(fw[key], cfg)
            L["loss_gen_focus_%s_size" % nm], L["loss_gen_focus_%s_digit" % nm] = float(sz), float(dg)
            fsum += float(sz) + float(dg)
        tot += cfg["focus_loss"] * fsum / H / W / B / 3 + cfg["recon_x_w"] * (L["loss_idt_A"] + L["loss_idt_B"])
    L["loss_gen_total"] = tot
    return cfg, L


def _forward_and_losses(T, B, S):
    o = _oracle_step(B, S)
    cfg, nets, x_a, x_b, z, fw, dA, d2 = o["cfg"], o["nets"], o["x_a"], o["x_b"], o["z"], o["fw"], o["dA"], o["d2"]
    tr = _make(T, cfg, nets)
    # ---- HIP forward through the public encode / decode / discriminator surface ----
    xa = x_a.cuda()
    zz = [t.cuda() for t in z[3:]]
    worst = {}

    def chk(name, got, want):
        worst[name] = _rel(got, want)

    c1, _ = tr.gen_AB.encode(xa); chk("c_1", c1, fw["c_1"])
    c2, s2 = tr.gen_BA.encode(xa); chk("c_2", c2, fw["c_2"]); chk("s_2", s2, fw["s_2"])
    xB4 = tr.gen_AB.decode(c1, zz[0])
    xA4 = tr.gen_BA.decode(c2, cfg["alpha"] * zz[1])
    chk("f_B", xB4[:, 3:], fw["f_B"]); chk("f_A", xA4[:, 3:], fw["f_A"])
    xB = tr.focus_translation(xB4[:, :3], xa, xB4[:, 3:]); chk("x_B_fake", xB, fw["x_B_fake"])
    xA = tr.focus_translation(xA4[:, :3], xa, xA4[:, 3:]); chk("x_A_fake", xA, fw["x_A_fake"])
    rec = tr.gen_BA.decode(c2, s2); chk("x_A_recon", rec[:, :3], fw["x_A_recon"])
    c3, _ = tr.gen_BA.encode(xB); chk("c_3", c3, fw["c_3"])
    xA24 = tr.gen_BA.decode(c3, zz[2])
    xA2 = tr.focus_translation(xA24[:, :3], xB, xA24[:, 3:]); chk("x_A2_fake", xA2, fw["x_A2_fake"])
    for s, (g_, w_) in enumerate(zip(tr.dis_A(xA), dA)):
        chk("dis_A_xA_s%d" % s, g_, w_)
    for s, (g_, w_) in enumerate(zip(tr.dis_2(torch.cat((xa, xA2), 1)), d2)):
        chk("dis_2_pA2_s%d" % s, g_, w_)
    print("forward max-abs rel errors @%dx%d B=%d:" % (S, S, B), {k: "%.2e" % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v < FTOL}
    assert not bad, bad
    # ---- the 16 losses from the two update calls (the step's own fused loss kernels), at the SHIPPED focus_epsilon ----
    cfg0, Lg = _default_eps_losses(o)
    Ld = dict(o["od"].losses)
    tr.dis_update(x_a, x_b, cfg0, z=z[:3])
    ld = {n: float(getattr(tr, n)) for n in Ld}
    tr2 = _make(T, cfg0, nets)
    tr2.gen_update(x_a, x_b, cfg0, z=z[3:])
    lg = {n: float(getattr(tr2, n)) for n in Lg}
    errs = {}
    for n, v in list(Ld.items()) + list(Lg.items()):
        v = float(v)
        got = ld[n] if n in ld else lg[n]
        tol = 5e-3 if n.endswith("_size") else LTOL
        errs[n] = (abs(got - v) / max(1e-3, abs(v)), tol, got, v)
    print("loss rel errors:", {k: "%.2e" % e[0] for k, e in errs.items()})
    bad = {k: e for k, e in errs.items() if not e[0] <= e[1]}
    assert not bad, bad


def _host_mem_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


def _require_host_memory(B, S):
    """the oracle's autograd graph of one update holds ~3.5 GB per 256x256 sample in host memory.  A host that
    cannot hold the benchmarked batch FAILS the test (round 3 shrank B silently: the log could not show which batch had been checked)."""
    need = B * 3.5 * (S / 256.0) ** 2 + 16
    avail = _host_mem_gb()
    assert not avail or need <= avail, "oracle step at %dx%d B=%d needs ~%.0f GB of host memory, %.0f GB available" % (S, S, B, need, avail)


FLOOR = 1e-3         # tensors whose reference gradient norm is below FLOOR * (largest gradient norm of the update) are reported separately


def _step_gradients(T, B, S):
    _require_host_memory(B, S)
    o = _oracle_step(B, S)
    cfg, nets, x_a, x_b, z, od, og = o["cfg"], o["nets"], o["x_a"], o["x_b"], o["z"], o["od"], o["og"]
    trd = _make(T, cfg, nets); trd.dis_update(x_a, x_b, cfg, z=z[:3])
    trg = _make(T, cfg, nets); trg.gen_update(x_a, x_b, cfg, z=z[3:])
    for n, v in list(od.losses.items()) + list(og.losses.items()):
        got = float(getattr(trd if n.startswith("loss_dis") else trg, n))
        assert abs(got - v) <= (5e-3 if n.endswith("_size") else LTOL) * max(1e-3, abs(v)), (n, got, v)
    # every tensor: PURE relative L2 error against its own reference norm.  Tensors whose reference norm is below FLOOR x the
    # update's largest gradient norm are listed by name and held to the same absolute error a FLOOR-sized tensor would be allowed.
    worst, small = [], []
    for tr, orc, nets_ in ((trd, od, ("dis_A", "dis_B", "dis_2")), (trg, og, ("gen_AB", "gen_BA"))):
        gmax = max(float(t.grad.norm()) for n in nets_ for t in orc.nets[n].values())
        for n in nets_:
            for k, gr in getattr(tr, n).named_grads():
                ref = orc.nets[n][k].grad.double()
                err = (gr.cpu().double() - ref).norm().item()
                rn = ref.norm().item()
                if rn >= FLOOR * gmax:
                    worst.append((err / rn, n, k))
                else:
                    small.append((err / (FLOOR * gmax), rn / gmax, n, k))
    worst.sort(reverse=True); small.sort(reverse=True)
    print("worst gradient tensors @%dx%d B=%d (relative L2, %d tensors):" % (S, S, B, len(worst)), [("%.2e" % e, n, k) for e, n, k in worst[:6]])
    print("tensors under the floor (|g_ref| < %.0e x largest; error relative to the floor, own norm relative to the largest): %d" % (FLOOR, len(small)),
          [("%.2e" % e, "%.1e" % r, n, k) for e, r, n, k in small[:8]])
    assert worst[0][0] <= GTOL, worst[:6]
    assert not small or small[0][0] <= GTOL, small[:6]


def test_forward_and_losses_256_b8(T):
    """BASELINE configs[1] at its real size (shares its oracle pass with test_step_gradients_256_b8 right below)."""
    _forward_and_losses(T, 8, 256)


def test_step_gradients_256_b8(T):
    """gradients AT THE BENCHMARKED BATCH (BASELINE configs[1]): the weight-gradient split plans (ordered pixel slices, Winograd
    A^T B with K = 2048 tiles) differ from the B=2 case below"""
    _step_gradients(T, 8, 256)


def test_forward_and_losses_512_b4(T):
    """BASELINE configs[3] (glasses-removal 512x512 fp32, batch 4) at its real size."""
    _forward_and_losses(T, 4, 512)


def test_step_gradients_512_b4(T):
    """BASELINE configs[3] at its batch"""
    _step_gradients(T, 4, 512)


def test_step_gradients_256_b2(T):
    _step_gradients(T, 2, 256)


def test_step_gradients_512_b1(T):
    _step_gradients(T, 1, 512)
