"""Step-level parity of the HIP path (aclgan_Trainer on libaclgan_hip) against the golden
fixtures (fp64 reference truth) and the fp32 CPU oracle.  Run on the GPU box: pytest -m gpu"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import aclgan_oracle as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    import aclgan_amd  # noqa: F401
    from aclgan_amd import trainer
    return trainer


def _load(name):
    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    return meta, data


def _make(T, cfg, nets):
    tr = T.aclgan_Trainer(cfg)
    for name in O.OracleTrainer.NETS:
        getattr(tr, name).load_state_dict(nets[name], strict=False)
    return tr


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_state_dict_surface(T):
    cfg = O.default_config()
    tr = T.aclgan_Trainer(cfg)
    want = {}
    for line in open(os.path.join(GOLDEN, "state_dict_keys.txt")):
        net, key, shp = line.split()
        want.setdefault(net, []).append((key, tuple(int(s) for s in shp.split("x"))))
    for net in O.OracleTrainer.NETS:
        sd = getattr(tr, net).state_dict()
        assert [(k, tuple(v.shape)) for k, v in sd.items()] == want[net], net
    # load/save round trip through the reference layout
    nets = O.test_nets(cfg, 3)
    tr.gen_AB.load_state_dict(nets["gen_AB"], strict=False)
    for k, v in tr.gen_AB.state_dict().items():
        if k in nets["gen_AB"]:
            assert torch.equal(v.cpu(), nets["gen_AB"][k]), k
    # trainer-level nn.Module surface: state_dict() / load_state_dict() / parameters() with the reference's 254 keys
    full = tr.state_dict()
    assert [(k.split(".", 1)[0], k.split(".", 1)[1], tuple(v.shape)) for k, v in full.items()] == \
           [(net, key, shp) for net in O.OracleTrainer.NETS for key, shp in want[net]]
    assert len(full) == 254 and len(tr.parameters()) == 222 and sum(p.numel() for p in tr.parameters()) == 30058648 + 24822729
    tr2 = T.aclgan_Trainer(cfg)
    tr2.load_state_dict(full)
    assert torch.equal(tr2._param[0], tr._param[0]) and torch.equal(tr2._param[1], tr._param[1])
    with pytest.raises(T.L.AclganError):
        tr2.load_state_dict({"bogus.weight": torch.zeros(1)})
    # the other init kinds of utils.py:279-288
    for kind, check in (("orthogonal", lambda w: (w.reshape(w.shape[0], -1) @ w.reshape(w.shape[0], -1).T - 2 * torch.eye(w.shape[0], device=w.device)).abs().max() < 1e-3),
                        ("default", lambda w: abs(float(w.abs().max()) - (1.0 / (w.shape[1] * 16)) ** 0.5) < 1e-3),
                        ("gaussian", lambda w: abs(float(w.std()) - 0.02) < 1e-3)):
        c2 = dict(cfg); c2["init"] = kind
        tk = T.aclgan_Trainer(c2)
        assert check(dict(tk.gen_BA.named_parameters())["enc_content.model.2.conv.weight"].contiguous()), kind
        assert float(dict(tk.gen_BA.named_parameters())["enc_content.model.2.conv.bias"].abs().max()) == 0.0
    # init statistics (utils.py:274-294, trainer.py:49-52)
    w = dict(tr.gen_BA.named_parameters())["enc_content.model.2.conv.weight"]
    assert abs(float(w.std()) - (2.0 / (128 * 16)) ** 0.5) < 2e-3
    wd = dict(tr.dis_2.named_parameters())["cnns.1.2.conv.weight"]
    assert abs(float(wd.std()) - 0.02) < 1e-3


@pytest.mark.parametrize("fix", ["step_reduced_64", "step_full_64", "step_reduced_64_plain"])
def test_forward_matches_reference(T, fix):
    """encode / decode / discriminator forward vs the fp64 reference tensors (north star: 1e-3 rel)."""
    meta, data = _load(fix)
    cfg = meta["config"]
    tr = _make(T, cfg, O.test_nets(cfg, 0))
    x_a = torch.from_numpy(data["x_a"]).cuda()
    z = [torch.from_numpy(data["z%d" % i]).cuda() for i in range(3)]
    stats = meta["fwd_stats"]

    def chk(name, t):
        if "fw_" + name in data.files:
            assert _rel(t, torch.from_numpy(data["fw_" + name])) < 1e-3, name
        s, nrm, mx = stats[name]
        assert abs(float(t.double().norm()) - nrm) <= 1e-3 * nrm, name

    c1, _ = tr.gen_AB.encode(x_a); chk("c_1", c1)
    c2, s2 = tr.gen_BA.encode(x_a); chk("c_2", c2); chk("s_2", s2)
    xB4 = tr.gen_AB.decode(c1, z[0]); chk("dec_AB_c1_z1", xB4)
    xA4 = tr.gen_BA.decode(c2, cfg["alpha"] * z[1]); chk("dec_BA_c2_z2", xA4)
    focus = cfg["focus_loss"] > 0      # the _plain fixture: non-focus configuration, the 3-channel decoder output is the image
    xB = tr.focus_translation(xB4[:, :3], x_a, xB4[:, 3:]) if focus else xB4; chk("x_B_fake", xB)
    xA = tr.focus_translation(xA4[:, :3], x_a, xA4[:, 3:]) if focus else xA4; chk("x_A_fake", xA)
    c3, _ = tr.gen_BA.encode(xB); chk("c_3", c3)
    xA24 = tr.gen_BA.decode(c3, z[2])
    xA2 = tr.focus_translation(xA24[:, :3], xB, xA24[:, 3:]) if focus else xA24; chk("x_A2_fake", xA2)
    dA = tr.dis_A(xA)
    for s in range(3):
        chk("dis_A_xA_s%d" % s, dA[s])
    d2 = tr.dis_2(torch.cat((x_a, xA2), 1))
    for s in range(3):
        chk("dis_2_pA2_s%d" % s, d2[s])


# per-tensor relative L2 of the HIP gradients against the fp32 oracle at 64x64, B <= 2 (measured worst, final round-3 build, recorded
# in profiles/r03_gpu_tests.log): one ReLU mask flip moves an upstream tensor by ~1 / sqrt(#elements of the layer)
ETOL_SMOOTH, ETOL_DEFAULT = 3e-2, 3e-2      # measured worst 9.6e-3 (reduced width, smooth) / 4.7e-3 (full width, default epsilon)


def _grads_by_name(tr, nets):
    out = {}
    for n in nets:
        for k, g in getattr(tr, n).named_grads():
            out[(n, k)] = g.contiguous().clone()
    return out


@pytest.mark.parametrize("fix,ltol,gtol", [("step_reduced_64_smooth", 1e-3, 1e-2), ("step_full_64_smooth", 1e-3, 1e-2),
                                            ("step_reduced_64", 1e-3, 1e-1), ("step_full_64", 1e-3, 1e-1),
                                            ("step_reduced_64_plain", 1e-3, 1e-2)])      # non-focus configuration (trainer.py:117-121,266-276)
def test_update_steps_match_reference(T, fix, ltol, gtol):
    """dis_update and gen_update (each from the fixture's initial weights) vs the fp64 reference:
    the 16 losses (1e-3 rel; 'size' losses 2e-2, a 200x-cancelling sum squared), every gradient
    tensor's L2 norm (1e-2 on the smooth fixtures; default focus_epsilon=0.01 fixtures 1e-1,
    see tests/golden/make_golden.py), and elementwise against the fp32 oracle."""
    meta, data = _load(fix)
    cfg = meta["config"]
    nets = O.test_nets(cfg, 0)
    x_a, x_b = torch.from_numpy(data["x_a"]), torch.from_numpy(data["x_b"])
    z = [torch.from_numpy(data["z%d" % i]) for i in range(6)]

    trd = _make(T, cfg, nets)
    trd.dis_update(x_a, x_b, cfg, z=z[:3])
    gd = _grads_by_name(trd, ("dis_A", "dis_B", "dis_2"))
    trg = _make(T, cfg, nets)
    trg.gen_update(x_a, x_b, cfg, z=z[3:6])
    gg = _grads_by_name(trg, ("gen_AB", "gen_BA"))
    torch.cuda.synchronize()

    for n, v in meta["losses"].items():
        got = float(getattr(trd if n.startswith("loss_dis") else trg, n))
        tol = 5e-3 if n.endswith("_size") else ltol    # centred, ordered summation (misc.hip: focus_sums_kernel)
        assert abs(got - v) <= tol * max(1e-3, abs(v)), (n, got, v)

    gmax = max(v[1] for v in meta["grad_stats"].values())
    for key, (s, nrm, mx) in meta["grad_stats"].items():
        upd, net, k = key.split("/", 2)
        g = (gd if upd == "dis_update" else gg)[(net, k)]
        assert abs(float(g.double().norm()) - nrm) <= gtol * nrm + 1e-5 * gmax, (key, float(g.double().norm()), nrm)

    # per-tensor relative L2 error vs the fp32 oracle on the same inputs (catches permutations a
    # norm cannot see).  Bound 3e-2 (smooth) / 1e-1 (default eps): at B<=2, 64x64 ONE ReLU /
    # LeakyReLU mask flip (a pre-activation within fp32 rounding of 0) perturbs every upstream
    # gradient by ~1/sqrt(#elements of that layer) ~ 2e-3..4e-3 in relative L2; measured: the fp32
    # CPU oracle itself sits 1e-3..3e-2 from the fp64 reference on these fixtures, and every HIP
    # backward kernel alone is exact to ~1e-7 on the same data (tests/test_gpu_ops.py,
    # scripts/diag_tail.py).
    # The forward and every loss value are bit-reproducible run to run (test_forward_is_bit_reproducible), so WHICH pre-activations
    # sit on the wrong side of zero relative to the oracle is fixed for a given build: the comparison is repeatable up to the
    # ~1e-7 atomics noise of the default backward (deterministic mode: exactly).  The bound covers the mask flips themselves
    # (they depend on the kernels' summation order, i.e. they change when a kernel's tiling changes, not between runs).
    etol = ETOL_SMOOTH if fix.endswith(("smooth", "plain")) else ETOL_DEFAULT
    seen = []

    def l2ok(g, ref, key):
        err = (g.cpu().double() - ref.double()).norm().item()
        seen.append((err / (ref.double().norm().item() + 1e-5 * gmax / etol), key))
        assert err <= etol * ref.double().norm().item() + 1e-5 * gmax, (key, err, ref.norm().item())

    orc = O.OracleTrainer(cfg, nets=nets)
    orc.dis_update(x_a, x_b, z[:3], apply=False)
    for (net, k), g in gd.items():
        l2ok(g, orc.nets[net][k].grad, (net, k))
    orc = O.OracleTrainer(cfg, nets=nets)
    orc.gen_update(x_a, x_b, z[3:6], apply=False)
    for (net, k), g in gg.items():
        l2ok(g, orc.nets[net][k].grad, (net, k))
    seen.sort(reverse=True)
    print("%s: worst per-tensor relative L2 vs the fp32 oracle:" % fix, [("%.2e" % e, k) for e, k in seen[:3]])

    # parameters after Adam: every element moved by at most ~lr on step 1, and the well-conditioned
    # ones agree with the reference's post-step statistics
    # Step 1 of Adam moves every element by ~lr * sign(g): an element whose gradient is within fp32 summation noise of zero may go
    # the other way (2 * lr off), which shifts the tensor norm by up to 2 * lr * |p_i| / ||p||.  The norm bound allows two such
    # elements of the largest magnitude on top of 1e-5 relative (seen: one element of a 64-entry LayerNorm beta, 1.8e-5).
    def norm_tol(nrm, mx):
        return 1e-5 * max(1.0, nrm) + 2 * (2 * cfg["lr"] * mx / max(nrm, 1e-12))

    for key, (s, nrm, mx) in meta["param_stats_after_gen"].items():
        net, k = key.split("/", 1)
        p = dict(getattr(trg, net).named_parameters())[k]
        assert abs(float(p.double().norm()) - nrm) <= norm_tol(nrm, mx), key
        assert (p.cpu() - nets[net][k]).abs().max().item() <= 1.01 * cfg["lr"] + 1e-9, key
    for key, (s, nrm, mx) in meta["param_stats_after_dis"].items():
        net, k = key.split("/", 1)
        p = dict(getattr(trd, net).named_parameters())[k]
        assert abs(float(p.double().norm()) - nrm) <= norm_tol(nrm, mx), key


def test_chained_steps_and_lr_schedule(T):
    """train.py:71-74,101 order: dis_update, gen_update, update_learning_rate; chained losses."""
    meta, data = _load("step_reduced_64")
    cfg = dict(meta["config"]); cfg["step_size"] = 1; cfg["gamma"] = 0.5
    nets = O.test_nets(cfg, 0)
    x_a, x_b = torch.from_numpy(data["x_a"]), torch.from_numpy(data["x_b"])
    z = [torch.from_numpy(data["z%d" % i]) for i in range(6)]
    tr = _make(T, cfg, nets)
    tr.dis_update(x_a, x_b, cfg, z=z[:3])
    tr.gen_update(x_a, x_b, cfg, z=z[3:6])
    for n, v in meta["seq_losses"].items():
        tol = 5e-3 if n.endswith("_size") else 2e-3
        assert abs(float(getattr(tr, n)) - v) <= tol * max(1e-3, abs(v)), n
    assert tr._current_lr(cfg) == cfg["lr"]
    tr.update_learning_rate()
    assert abs(tr._current_lr(cfg) - 0.5 * cfg["lr"]) < 1e-12
    # default path: z drawn from the CPU generator in the reference's order
    torch.manual_seed(5)
    tr.dis_update(x_a, x_b, cfg)
    assert np.isfinite(float(tr.loss_dis_total))


def test_checkpoint_roundtrip(T, tmp_path):
    meta, data = _load("step_reduced_64")
    cfg = meta["config"]
    tr = _make(T, cfg, O.test_nets(cfg, 0))
    x_a, x_b = torch.from_numpy(data["x_a"]), torch.from_numpy(data["x_b"])
    z = [torch.from_numpy(data["z%d" % i]) for i in range(6)]
    tr.dis_update(x_a, x_b, cfg, z=z[:3]); tr.gen_update(x_a, x_b, cfg, z=z[3:6])
    tr.save(str(tmp_path), 41)
    assert sorted(os.listdir(tmp_path)) == ["dis_00000042.pt", "gen_00000042.pt", "optimizer.pt"]
    sd = torch.load(os.path.join(tmp_path, "gen_00000042.pt"))
    assert set(sd.keys()) == {"AB", "BA"}
    tr2 = T.aclgan_Trainer(cfg)
    assert tr2.resume(str(tmp_path), cfg) == 42
    for n in O.OracleTrainer.NETS:
        for (k, a), (_, b) in zip(getattr(tr, n).named_parameters(), getattr(tr2, n).named_parameters()):
            assert torch.equal(a, b), (n, k)
    # identical continuation
    tr.dis_update(x_a, x_b, cfg, z=z[:3]); tr2.dis_update(x_a, x_b, cfg, z=z[:3])
    assert abs(float(tr.loss_dis_total) - float(tr2.loss_dis_total)) < 1e-5
    pa = dict(tr.dis_A.named_parameters())["cnns.0.1.conv.weight"]; pb = dict(tr2.dis_A.named_parameters())["cnns.0.1.conv.weight"]
    assert (pa - pb).abs().max().item() < 1e-7


def test_sample_returns_reference_tuple(T):
    cfg = O.default_config(); cfg["display_size"] = 2
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1); cfg["dis"].update(dim=8)
    tr = T.aclgan_Trainer(cfg)
    x = torch.rand(2, 3, 64, 64) * 2 - 1
    out = tr.sample(x, x)
    assert len(out) == 9
    # (x_A, x_A_fake, mask_A, x_B_fake, mask_B, x_A2_fake, mask_A2, x_A_recon, mask_recon)  trainer.py:237
    assert [o.shape[1] for o in out] == [3, 3, 1, 3, 1, 3, 1, 3, 1]
    assert all(tuple(o.shape[2:]) == (64, 64) and o.shape[0] == 2 for o in out)


def test_full_size_step_properties(T):
    """BASELINE.json configs[1] shape (256x256, B=8, full width): size-independent properties.
    (a) losses finite and in range; (b) gen_update leaves the discriminators untouched and
    dis_update the generators (trainer.py:39-42: separate optimizers); (c) Adam's first step
    moves every parameter by at most lr; (d) the L1 identity losses equal a torch recomputation
    from the HIP forward (encode -> decode round trip through the public API)."""
    cfg = O.default_config()
    tr = T.aclgan_Trainer(cfg)
    g = torch.Generator().manual_seed(1)
    x_a = torch.rand(8, 3, 256, 256, generator=g) * 2 - 1
    x_b = torch.rand(8, 3, 256, 256, generator=g) * 2 - 1
    z = [torch.randn(8, 8, 1, 1, generator=g) for _ in range(3)]
    gen0 = tr._param[0].clone(); dis0 = tr._param[1].clone()
    c2, s2 = tr.gen_BA.encode(x_a)
    rec = tr.gen_BA.decode(c2, s2)[:, :3]
    idt = (rec - x_a.cuda()).abs().mean().item()
    tr.gen_update(x_a, x_b, cfg, z=z)
    torch.cuda.synchronize()
    assert abs(float(tr.loss_idt_A) - idt) <= 1e-4 * idt
    for n in ["loss_gen_adv_A", "loss_gen_adv_B", "loss_gen_adv_2", "loss_idt_A", "loss_idt_B", "loss_gen_total"]:
        assert np.isfinite(float(getattr(tr, n))) and 0 <= float(getattr(tr, n)) < 1e3, n
    assert torch.equal(tr._param[1], dis0)
    d = (tr._param[0] - gen0).abs().max().item()
    assert 0 < d <= 1.01 * cfg["lr"]
    gen1 = tr._param[0].clone()
    tr.dis_update(x_a, x_b, cfg, z=z)
    torch.cuda.synchronize()
    assert torch.equal(tr._param[0], gen1)
    d = (tr._param[1] - dis0).abs().max().item()
    assert 0 < d <= 1.01 * cfg["lr"]
    assert np.isfinite(float(tr.loss_dis_total))


def test_non_square_odd_batch_step_matches_oracle(T):
    """B=3 (the reference's default batch_size), 64x96 images, reduced width: losses and gradients of both
    updates against the fp32 oracle (smooth focus_epsilon so the gradient comparison is meaningful)."""
    cfg = O.default_config()
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=2)
    cfg["dis"].update(dim=8)
    cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5
    nets = O.test_nets(cfg, 4)
    g = torch.Generator().manual_seed(9)
    x_a = torch.rand(3, 3, 64, 96, generator=g) * 2 - 1
    x_b = torch.rand(3, 3, 64, 96, generator=g) * 2 - 1
    z = [torch.randn(3, 8, 1, 1, generator=g) for _ in range(6)]
    trd = _make(T, cfg, nets); trd.dis_update(x_a, x_b, cfg, z=z[:3])
    trg = _make(T, cfg, nets); trg.gen_update(x_a, x_b, cfg, z=z[3:])
    od = O.OracleTrainer(cfg, nets=nets); od.dis_update(x_a, x_b, z[:3], apply=False)
    og = O.OracleTrainer(cfg, nets=nets); og.gen_update(x_a, x_b, z[3:], apply=False)
    for n, v in list(od.losses.items()) + list(og.losses.items()):
        got = float(getattr(trd if n.startswith("loss_dis") else trg, n))
        assert abs(got - v) <= (5e-3 if n.endswith("_size") else 1e-3) * max(1e-3, abs(v)), (n, got, v)
    for tr, orc, nets_ in ((trd, od, ("dis_A", "dis_B", "dis_2")), (trg, og, ("gen_AB", "gen_BA"))):
        gmax = max(float(t.grad.norm()) for n in nets_ for t in orc.nets[n].values())
        for n in nets_:
            for k, gr in getattr(tr, n).named_grads():
                ref = orc.nets[n][k].grad
                err = (gr.cpu().double() - ref.double()).norm().item()
                assert err <= 3e-2 * ref.double().norm().item() + 1e-5 * gmax, (n, k, err)


def test_step_at_sizes_divisible_by_4_only(T):
    """The reference accepts any H, W >= 64 divisible by 2^n_downsample = 4 (odd intermediate maps in the discriminator
    pyramid and the style encoder); so does the build: 72x88, B=2, both updates against the fp32 oracle."""
    cfg = O.default_config()
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1)
    cfg["dis"].update(dim=8)
    cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5
    nets = O.test_nets(cfg, 5)
    g = torch.Generator().manual_seed(19)
    x_a = torch.rand(2, 3, 72, 88, generator=g) * 2 - 1
    x_b = torch.rand(2, 3, 72, 88, generator=g) * 2 - 1
    z = [torch.randn(2, 8, 1, 1, generator=g) for _ in range(6)]
    trd = _make(T, cfg, nets); trd.dis_update(x_a, x_b, cfg, z=z[:3])
    trg = _make(T, cfg, nets); trg.gen_update(x_a, x_b, cfg, z=z[3:])
    od = O.OracleTrainer(cfg, nets=nets); od.dis_update(x_a, x_b, z[:3], apply=False)
    og = O.OracleTrainer(cfg, nets=nets); og.gen_update(x_a, x_b, z[3:], apply=False)
    for n, v in list(od.losses.items()) + list(og.losses.items()):
        got = float(getattr(trd if n.startswith("loss_dis") else trg, n))
        assert abs(got - v) <= (5e-3 if n.endswith("_size") else 1e-3) * max(1e-3, abs(v)), (n, got, v)
    for tr, orc, nets_ in ((trd, od, ("dis_A", "dis_B", "dis_2")), (trg, og, ("gen_AB", "gen_BA"))):
        gmax = max(float(t.grad.norm()) for n in nets_ for t in orc.nets[n].values())
        for n in nets_:
            for k, gr in getattr(tr, n).named_grads():
                ref = orc.nets[n][k].grad
                err = (gr.cpu().double() - ref.double()).norm().item()
                assert err <= 3e-2 * ref.double().norm().item() + 1e-5 * gmax, (n, k, err)


def test_inference_shapes_and_sample_values(T):
    """(a) forward-only calls take the shapes test.py produces (Resize(256) of a non-square photo: 64x84 here), with a
    forward-only workspace; (b) sample() (trainer.py:179-245) VALUES against the oracle, all 9 outputs."""
    cfg = O.default_config()
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1)
    cfg["dis"].update(dim=8)
    cfg["display_size"] = 2
    nets = O.test_nets(cfg, 6)
    tr = _make(T, cfg, nets)
    g = torch.Generator().manual_seed(23)
    x = torch.rand(1, 3, 64, 84, generator=g) * 2 - 1
    c, s_ = tr.gen_AB.encode(x)
    img = tr.gen_AB.decode(c, s_)
    P, gc = nets["gen_AB"], cfg["gen"]
    assert _rel(c, O.content_encode(P, x, gc)) < 1e-3
    assert _rel(img, O.decode(P, O.content_encode(P, x, gc), O.style_encode(P, x, gc), gc)) < 1e-3
    assert tr._ws_shape[3] is False     # no training arena was allocated for it
    x_a = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    x_b = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    out = tr.sample(x_a, x_b)
    z1, z2, z3 = tr.z_1.cpu(), tr.z_2.cpu(), tr.z_3.cpu()
    AB, BA = nets["gen_AB"], nets["gen_BA"]
    want = [[] for _ in range(9)]
    for i in range(2):
        xa = x_a[i:i + 1]
        c1, s1 = O.content_encode(BA, xa, gc), O.style_encode(BA, xa, gc)
        o = O.decode(BA, c1, z1[i:i + 1], gc); a_fake = O.focus_translation(o[:, :3], xa, o[:, 3:]); m_a = o[:, 3:]
        o = O.decode(BA, c1, s1, gc); a_rec, m_rec = o[:, :3], o[:, 3:]
        o = O.decode(AB, O.content_encode(AB, xa, gc), z2[i:i + 1], gc); b_fake = O.focus_translation(o[:, :3], xa, o[:, 3:]); m_b = o[:, 3:]
        o = O.decode(BA, O.content_encode(BA, b_fake, gc), z3[i:i + 1], gc); a2 = O.focus_translation(o[:, :3], b_fake, o[:, 3:]); m_a2 = o[:, 3:]
        for lst, t in zip(want, (xa, a_fake, m_a, b_fake, m_b, a2, m_a2, a_rec, m_rec)):
            lst.append(t)
    for k, (got, w) in enumerate(zip(out, want)):
        assert _rel(got, torch.cat(w)) < 1e-3, k


def test_resume_from_reference_written_checkpoint(T, tmp_path):
    """f-2: a checkpoint written by the REFERENCE's trainer.save (tests/golden/ckpt_reference_reduced, produced by
    make_golden.py --checkpoint-only) loads into the build: same file names / keys / layouts / Adam state; the next
    dis_update reproduces the reference's next loss."""
    import shutil
    src = os.path.join(GOLDEN, "ckpt_reference_reduced")
    exp = json.load(open(os.path.join(src, "expect.json")))
    for f in exp["files"]:
        shutil.copy(os.path.join(src, f), tmp_path)
    meta, data = _load("step_reduced_64")
    cfg = meta["config"]
    tr = T.aclgan_Trainer(cfg)
    assert tr.resume(str(tmp_path), cfg) == 7
    ref_gen = torch.load(os.path.join(src, "gen_00000007.pt"), map_location="cpu")
    for k, v in tr.gen_BA.state_dict().items():
        assert torch.equal(v.cpu(), ref_gen["BA"][k]), k
    assert tr._opt[0]["steps"] == 1 and tr._opt[1]["steps"] == 1
    ref_opt = torch.load(os.path.join(src, "optimizer.pt"), map_location="cpu")
    m0 = dict(tr.dis_A.named_parameters())   # exp_avg of the first dis tensor through the flat buffer
    e = tr.dis_A._entries[0]
    assert torch.equal(tr.dis_A._view(e, tr._m[1]).cpu(), ref_opt["dis"]["state"][0]["exp_avg"])
    x_a, x_b = torch.from_numpy(data["x_a"]), torch.from_numpy(data["x_b"])
    z = [torch.from_numpy(data["z%d" % i]) for i in range(3)]
    tr.dis_update(x_a, x_b, cfg, z=z)
    assert abs(float(tr.loss_dis_total) - exp["loss_dis_total_after_resume"]) <= 2e-3 * exp["loss_dis_total_after_resume"]
    # and the build's own save() has the reference's on-disk structure
    out = os.path.join(tmp_path, "mine"); os.makedirs(out)
    tr.save(out, 7)
    mine = torch.load(os.path.join(out, "optimizer.pt"), map_location="cpu")
    assert set(mine.keys()) == set(ref_opt.keys()) == {"gen", "dis"}
    assert set(mine["dis"]["state"][0].keys()) == set(ref_opt["dis"]["state"][0].keys())
    assert mine["dis"]["param_groups"][0]["params"] == ref_opt["dis"]["param_groups"][0]["params"]
    mg = torch.load(os.path.join(out, "gen_00000008.pt"), map_location="cpu")
    assert list(mg["AB"].keys()) == list(ref_gen["AB"].keys())


def test_sample_non_focus_configuration(T):
    """sample() with focus_loss 0 / gen.output_dim 3 (trainer.py:216-230,238-245): the reference's 7-tuple, values against the
    reference's own outputs -- including its x_B_recon of B x B images (it encodes the whole batch x_b inside the per-image loop)."""
    meta, data = _load("step_reduced_64_plain")
    ref = np.load(os.path.join(GOLDEN, "sample_reduced_64_plain.npz"))
    cfg = meta["config"]
    tr = _make(T, cfg, O.test_nets(cfg, 0))
    x_a, x_b = torch.from_numpy(data["x_a"]), torch.from_numpy(data["x_b"])
    tr.z_1, tr.z_2, tr.z_3 = (torch.from_numpy(data["z%d" % i]).cuda() for i in range(3))
    outs = tr.sample(x_a, x_b)
    names = ["x_A", "x_A_fake", "x_B_fake", "x_A2_fake", "x_A_recon", "x_B", "x_B_recon"]
    assert len(outs) == 7
    for n, o in zip(names, outs):
        r = torch.from_numpy(ref[n])
        assert tuple(o.shape) == tuple(r.shape), (n, o.shape, r.shape)
        assert _rel(o, r) < 1e-3, n


def test_errors_surface_as_exceptions_not_aborts(T):
    """reference behaviour at the boundary: bad shapes / unsupported branches raise (networks.py:74,325 'assert 0');
    here they come back as codes + messages from the C ABI and become AclganError."""
    cfg = O.default_config()
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1); cfg["dis"].update(dim=8); cfg["display_size"] = 1
    tr = T.aclgan_Trainer(cfg)
    x = torch.zeros(1, 3, 70, 64)
    with pytest.raises(T.L.AclganError, match="multiples of 4"):
        tr.gen_update(x, x, cfg)
    with pytest.raises(T.L.AclganError, match="H,W>=64"):
        tr.dis_update(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32), cfg)
    with pytest.raises(T.L.AclganError, match=r"\(B,3,H,W\)"):
        tr.gen_update(torch.zeros(1, 4, 64, 64), torch.zeros(1, 4, 64, 64), cfg)
    nofocus = dict(cfg); nofocus["focus_loss"] = 0
    with pytest.raises(T.L.AclganError, match="focus_loss"):
        tr.gen_update(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64), nofocus)
    # the trainer is still usable afterwards
    tr.dis_update(torch.rand(1, 3, 64, 64) * 2 - 1, torch.rand(1, 3, 64, 64) * 2 - 1, cfg)
    assert np.isfinite(float(tr.loss_dis_total))


def test_forward_is_bit_reproducible(T):
    """encode / decode / discriminator forward give identical bits on repeated calls (split-K layers reduce ordered
    partials, not atomics); full width so that the split-K and sub-pixel ring launches are on the path."""
    meta, data = _load("step_full_64")
    cfg = meta["config"]
    tr = _make(T, cfg, O.test_nets(cfg, 0))
    x = torch.from_numpy(data["x_a"]).cuda()
    z = torch.from_numpy(data["z0"]).cuda()
    outs = []
    for _ in range(3):
        c, s_ = tr.gen_AB.encode(x)
        img = tr.gen_AB.decode(c, z)
        d = tr.dis_2(torch.cat((x, img[:, :3]), 1))
        outs.append([c.clone(), s_.clone(), img.clone()] + [t.clone() for t in d])
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))


def test_inference_script_matches_oracle(T, tmp_path):
    """test.py counterpart (reference test.py:88-131): translate() against the fp32 oracle, then the CLI end to end
    on a checkpoint written by save() and a PNG read through the PIL Resize(new_size) path."""
    import subprocess
    import sys
    import yaml
    from PIL import Image
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("aclgan_test_script", os.path.join(ROOT, "test.py"))
    script = importlib.util.module_from_spec(spec); spec.loader.exec_module(script)

    meta, data = _load("step_reduced_64")
    cfg = meta["config"]
    nets = O.test_nets(cfg, 0)
    tr = _make(T, cfg, nets)
    x = torch.from_numpy(data["x_a"])[:1]
    styles = torch.randn(3, cfg["gen"]["style_dim"], 1, 1, generator=torch.Generator().manual_seed(3))
    for a2b, net in ((True, "gen_AB"), (False, "gen_BA")):
        got = script.translate(tr, x.cuda(), styles.cuda(), a2b=a2b)
        P, g = nets[net], cfg["gen"]
        c = O.content_encode(P, x, g)
        for j in range(3):
            o4 = O.decode(P, c, styles[j:j + 1], g)
            img, mask = o4[:, :3], o4[:, 3:]
            m = ((mask + 1) / 2).repeat(1, 3, 1, 1)
            want = (((img + 1) / 2) * m + ((x + 1) / 2) * (1 - m)) * 2 - 1     # test.py:73-76
            out, out_mask, out_img = got[j]
            assert _rel(out, (want + 1) / 2) < 1e-3
            assert _rel(out_img, img) < 1e-3 and _rel(out_mask, mask.expand(-1, 3, -1, -1)) < 1e-3
    # style-image branch (test.py:100-101): the style code comes from the encoder
    got = script.translate(tr, x.cuda(), None, a2b=True, style_image=x.cuda())
    s = O.style_encode(nets["gen_AB"], x, cfg["gen"])
    o4 = O.decode(nets["gen_AB"], O.content_encode(nets["gen_AB"], x, cfg["gen"]), s, cfg["gen"])
    assert len(got) == 1 and _rel(got[0][2], o4[:, :3]) < 1e-3

    # CLI: same flags as the reference script
    tr.save(str(tmp_path), 6)
    cfg_path = os.path.join(tmp_path, "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)).save(os.path.join(tmp_path, "in.png"))
    out_dir = os.path.join(tmp_path, "out")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "test.py"), "--config", cfg_path, "--input", os.path.join(tmp_path, "in.png"),
                        "--output_folder", out_dir, "--checkpoint", os.path.join(tmp_path, "gen_00000007.pt"), "--num_style", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = sorted(os.listdir(out_dir))
    assert names == ["input.jpg", "output000.jpg", "output000_img.jpg", "output000_mask.jpg",
                     "output001.jpg", "output001_img.jpg", "output001_mask.jpg"]
    new_size = cfg["new_size"]
    w, h = Image.open(os.path.join(out_dir, "output000.jpg")).size
    assert (h, w) == (new_size, int(new_size * 96 / 64))   # Resize(int): smaller edge -> new_size (256 x 384)
