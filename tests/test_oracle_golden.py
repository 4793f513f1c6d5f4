"""The oracle (oracle/aclgan_oracle.py) replayed against the golden vectors captured from the
reference implementation (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import aclgan_oracle as O

from conftest import GOLDEN


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def opv():
    return np.load(os.path.join(GOLDEN, "op_vectors.npz"))


def test_conv_block_combos(opv):
    combos = json.loads(str(opv["cb_combos"]))
    for i, (ci, co, k, s, p, norm, act, H) in enumerate(combos):
        for B in (1, 2):
            key = "cb%d_B%d" % (i, B)
            args = None
            if norm == "adain":
                args = (T(opv[key + "_adain_w"]).view(B, co), T(opv[key + "_adain_b"]).view(B, co))
            if norm == "ln":
                args = (T(opv[key + "_gamma"]), T(opv[key + "_beta"]))
            y = O.conv_block(T(opv[key + "_x"]), T(opv[key + "_w"]), T(opv[key + "_b"]), s, p, act, norm, args)
            ref = T(opv[key + "_y"])
            assert y.shape == ref.shape
            assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), key


def test_upsample_ln_block(opv):
    y = O.conv_block(T(opv["up_x"]), T(opv["up_w"]), T(opv["up_b"]), 1, 2, "relu", "ln",
                     (T(opv["up_gamma"]), T(opv["up_beta"])), upsample=True)
    assert (y - T(opv["up_y"])).abs().max().item() < 2e-5


def test_avgpool_pyramid(opv):
    x = T(opv["pool_x"])
    assert torch.equal(O.avgpool3s2(x), T(opv["pool_y1"]))
    assert (O.avgpool3s2(O.avgpool3s2(x)) - T(opv["pool_y2"])).abs().max().item() < 1e-6
    assert (O.avgpool3s2(T(opv["pool_odd_x"])) - T(opv["pool_odd_y"])).abs().max().item() < 1e-6


def test_lsgan_target_conventions(opv):
    dcfg = dict(dim=4, norm="none", activ="lrelu", n_layer=4, gan_type="lsgan", num_scales=3, pad_type="reflect")
    P = {k[len("lsgan_D_"):]: T(opv[k]) for k in opv.files if k.startswith("lsgan_D_")}
    xf, xr = T(opv["lsgan_x_fake"]), T(opv["lsgan_x_real"])
    dis = O.lsgan(O.dis_forward(P, xf, dcfg), 0.0) + O.lsgan(O.dis_forward(P, xr, dcfg), 1.0)
    gen = O.lsgan(O.dis_forward(P, xf, dcfg), 1.0)
    d2 = O.lsgan(O.dis_forward(P, xf, dcfg), 1.0) + O.lsgan(O.dis_forward(P, xr, dcfg), 0.0)
    assert abs(float(dis) - float(opv["lsgan_dis_loss"])) < 1e-5
    assert abs(float(gen) - float(opv["lsgan_gen_loss"])) < 1e-5
    assert abs(float(d2) - float(opv["lsgan_gen_d2_loss"])) < 1e-5


def test_focus_translation(opv):
    y = O.focus_translation(T(opv["ft_fg"]), T(opv["ft_bg"]), T(opv["ft_focus"]))
    assert (y - T(opv["ft_y"])).abs().max().item() < 1e-6


def test_state_dict_keys_match_reference():
    hp = O.default_config()
    want = {}
    for line in open(os.path.join(GOLDEN, "state_dict_keys.txt")):
        net, key, shp = line.split()
        want.setdefault(net, {})[key] = tuple(int(s) for s in shp.split("x"))
    gs = dict(O.gen_param_shapes(3, hp["gen"]))
    gs.update(O.gen_buffer_shapes(hp["gen"]))
    for net in ("gen_AB", "gen_BA"):
        assert want[net] == gs
    assert want["dis_A"] == dict(O.dis_param_shapes(3, hp["dis"]))
    assert want["dis_2"] == dict(O.dis_param_shapes(6, hp["dis"]))
    order = json.load(open(os.path.join(GOLDEN, "param_order.json")))
    mine = ["gen_AB/" + k for k in O.gen_param_shapes(3, hp["gen"])] + ["gen_BA/" + k for k in O.gen_param_shapes(3, hp["gen"])]
    assert order["gen"] == mine


def _load(name):
    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    return meta, data


def _run(meta, data, dtype):
    cfg = meta["config"]
    nets = {k: {n: t.to(dtype) for n, t in v.items()} for k, v in O.test_nets(cfg, 0).items()}
    x_a, x_b = T(data["x_a"]).to(dtype), T(data["x_b"]).to(dtype)
    z = [T(data["z%d" % i]).to(dtype) for i in range(6)]
    orc_d = O.OracleTrainer(cfg, nets=nets)
    orc_d.dis_update(x_a, x_b, z[:3])
    orc_g = O.OracleTrainer(cfg, nets=nets)
    _, fw, _ = orc_g.gen_update(x_a, x_b, z[3:6])
    return orc_d, orc_g, nets


def test_seeded_weights_are_the_fixture_weights():
    meta, _ = _load("step_reduced_64")
    nets = O.test_nets(meta["config"], 0)
    for key, (s, n, m) in meta["param_stats_initial"].items():
        net, k = key.split("/", 1)
        t = nets[net][k].double()
        assert abs(float(t.norm()) - n) <= 1e-9 * max(1.0, n), key
        assert abs(float(t.sum()) - s) <= 1e-7 * max(1.0, n), key


@pytest.mark.parametrize("name", ["step_reduced_64", "step_reduced_64_smooth", "step_reduced_64_plain"])      # _plain: the non-focus configuration
def test_step_fp64_reproduces_reference(name):
    """In float64 the oracle must be the same function as the reference: 1e-9."""
    meta, data = _load(name)
    orc_d, orc_g, _ = _run(meta, data, torch.float64)
    losses = dict(orc_d.losses)
    losses.update(orc_g.losses)
    for n, v in meta["losses"].items():
        assert abs(losses[n] - v) <= 1e-9 * max(1.0, abs(v)), n
    for key, (s, nrm, mx) in meta["grad_stats"].items():
        upd, net, k = key.split("/", 2)
        g = (orc_d if upd == "dis_update" else orc_g).nets[net][k].grad
        assert abs(float(g.norm()) - nrm) <= 1e-8 * max(1e-12, nrm) + 1e-14, key
    for key, (s, nrm, mx) in meta["param_stats_after_gen"].items():
        net, k = key.split("/", 1)
        p = orc_g.nets[net][k].detach()
        assert abs(float(p.sum()) - s) <= 1e-9 * max(1.0, nrm), key
    for key, (s, nrm, mx) in meta["param_stats_after_dis"].items():
        net, k = key.split("/", 1)
        p = orc_d.nets[net][k].detach()
        assert abs(float(p.sum()) - s) <= 1e-9 * max(1.0, nrm), key


@pytest.mark.parametrize("name", ["step_reduced_64_smooth", "step_full_64_smooth", "step_full_64", "step_reduced_64_plain"])
def test_step_fp32_within_fp32_noise(name):
    """The oracle as it is used on the GPU box (fp32) against the fp64 reference truth.
    Tolerances: losses 1e-4 rel (focus 'size' losses 1e-2: a 200x-cancelling sum squared);
    gradient norms 3e-3 rel on the smooth fixtures, 5e-2 on the default one (sign
    discontinuity of the digit loss, see make_golden.run_step_fixture)."""
    meta, data = _load(name)
    orc_d, orc_g, _ = _run(meta, data, torch.float32)
    losses = dict(orc_d.losses)
    losses.update(orc_g.losses)
    for n, v in meta["losses"].items():
        tol = 1e-2 if n.endswith("_size") else 1e-4
        assert abs(losses[n] - v) <= tol * max(1e-3, abs(v)), (n, losses[n], v)
    gtol = 3e-3 if name.endswith(("smooth", "plain")) else 5e-2      # (no digit loss in the non-focus configuration either)
    gmax = max(v[1] for v in meta["grad_stats"].values())
    for key, (s, nrm, mx) in meta["grad_stats"].items():
        upd, net, k = key.split("/", 2)
        g = (orc_d if upd == "dis_update" else orc_g).nets[net][k].grad
        assert abs(float(g.norm()) - nrm) <= gtol * nrm + 1e-6 * gmax, key


def test_forward_tensors_fp32():
    meta, data = _load("step_reduced_64")
    cfg = meta["config"]
    nets = O.test_nets(cfg, 0)
    x_a, x_b = T(data["x_a"]), T(data["x_b"])
    z = [T(data["z%d" % i]) for i in range(3)]
    with torch.no_grad():
        fw = O.generator_forward(nets["gen_AB"], nets["gen_BA"], x_a, x_b, z, cfg, with_recon=False)
        for k in ("c_1", "c_2", "s_2", "x_B_fake", "x_A_fake", "c_3", "x_A2_fake"):
            ref = T(data["fw_" + k])
            assert (fw[k] - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), k
        d = O.dis_forward(nets["dis_A"], fw["x_A_fake"], cfg["dis"])
        for s in range(3):
            ref = T(data["fw_dis_A_xA_s%d" % s])
            assert (d[s] - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
