#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE on CPU.

Run only in the build container (needs /root/reference):
    python tests/golden/make_golden.py

The reference (hyperplane-lab/ACL-GAN) ships no tests or golden vectors of its own, so the
oracle (oracle/aclgan_oracle.py) and the HIP path are pinned against outputs of the
reference implementation itself, captured here as *data only* (inputs, seeded weights identified by
per-tensor checksums, expected outputs).  No reference source travels.

Shims needed to import the reference on this CPU-only, torchvision-less box:
  * empty stub modules for torchvision (utils.py imports it at module level),
  * torch.Tensor.cuda -> identity (trainer.py hard-codes .cuda()),
  * YAML loaded with yaml.safe_load instead of utils.get_config.
Style noise is injected by temporarily replacing torch.randn with a queue of fixed tensors
(the reference draws z_1, z_2, z_3 in that order per update; trainer.py:99-101, 254-256).
"""
import copy
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

for m in ["torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.models",
          "torchvision.datasets"]:
    sys.modules[m] = types.ModuleType(m)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self
torch.set_num_threads(8)

import trainer as ref_trainer  # noqa: E402  (the reference)
import networks as ref_networks  # noqa: E402

from oracle import aclgan_oracle as O  # noqa: E402


def base_config():
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/male2female.yaml")))
    cfg["display_size"] = 2
    return cfg


def reduced_config():
    cfg = base_config()
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1)
    cfg["dis"].update(dim=8)
    return cfg


def seeded_inputs(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x_a = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    x_b = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    z = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(6)]
    return x_a, x_b, z


def fill_reference(tr, cfg):
    """Load the seeded test weights (oracle.test_nets) into the reference trainer; returns them."""
    nets = O.test_nets(cfg, seed=cfg.get("_fill_seed", 0))
    for name in O.OracleTrainer.NETS:
        mod = getattr(tr, name)
        sd = mod.state_dict()
        for k, v in nets[name].items():
            assert sd[k].shape == v.shape, (name, k, sd[k].shape, v.shape)
            sd[k] = v.clone()
        mod.load_state_dict(sd)
    return nets


class RandnQueue:
    def __init__(self, items):
        self.items = list(items)
        self._orig = torch.randn

    def __enter__(self):
        def fake(*a, **k):
            return self.items.pop(0).clone()
        torch.randn = fake
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig


LOSS_NAMES_DIS = ["loss_dis_A", "loss_dis_B", "loss_dis_2", "loss_dis_total"]
LOSS_NAMES_GEN = ["loss_gen_adv_A", "loss_gen_adv_B", "loss_gen_adv_2",
                  "loss_gen_focus_B_size", "loss_gen_focus_B_digit", "loss_gen_focus_A_size",
                  "loss_gen_focus_A_digit", "loss_gen_focus_A2_size", "loss_gen_focus_A2_digit",
                  "loss_idt_A", "loss_idt_B", "loss_gen_total"]


def tstats(t):
    t = t.detach().double()
    return [float(t.sum()), float(t.norm()), float(t.abs().max())]


def run_step_fixture(cfg, B, H, W, seed, fname, store_tensors):
    """One dis_update and one gen_update of the REFERENCE, each from the same initial weights,
    evaluated in float64 (``trainer.double()``), so the stored numbers are free of fp32
    rounding noise.  Why fp64: the reference's own fp32 gradients deviate from this fp64
    truth by up to 9e-3 rel (measured at full width; the focus 'digit' loss
    sum(1/(|m-0.5|+eps)), trainer.py:151, has a gradient sign discontinuity of 1/eps^2 = 1e4
    at m = 0.5, and Adam's first step turns gradient noise into O(lr) parameter noise), which
    would otherwise force loose tolerances.  fp32 implementations (the oracle in fp32, the
    HIP path) are compared against this truth with explicit fp32 tolerances in the tests;
    the '_smooth' fixtures (focus_epsilon 0.5, else default) allow tight gradient checks.
    The chained order of train.py:71-74 (dis then gen on updated weights) is recorded as
    'seq_losses'."""
    dd = torch.float64
    x_a, x_b, z = seeded_inputs(B, H, W, seed)
    xa, xb, zd = x_a.to(dd), x_b.to(dd), [t.to(dd) for t in z]
    out = {"x_a": x_a.numpy(), "x_b": x_b.numpy()}
    for i, t in enumerate(z):
        out["z%d" % i] = t.numpy()
    meta = {"config": cfg, "B": B, "H": H, "W": W, "seed": seed, "dtype": "float64 reference",
            "losses": {}, "seq_losses": {}, "grad_stats": {}, "param_stats_initial": {},
            "param_stats_after_dis": {}, "param_stats_after_gen": {}, "fwd_stats": {}}

    def fresh():
        tr = ref_trainer.aclgan_Trainer(cfg)
        nets = fill_reference(tr, cfg)
        return tr.double(), nets

    tr, nets0 = fresh()
    for name in O.OracleTrainer.NETS:
        for k, p in getattr(tr, name).named_parameters():
            meta["param_stats_initial"]["%s/%s" % (name, k)] = tstats(p)

    # forward intermediates of the shared generator pass, from the reference modules
    with torch.no_grad():
        c1, _ = tr.gen_AB.encode(xa)
        c2, s2 = tr.gen_BA.encode(xa)
        xB4 = tr.gen_AB.decode(c1, zd[0])
        xA4 = tr.gen_BA.decode(c2, cfg["alpha"] * zd[1])
        focus = cfg["focus_loss"] > 0      # trainer.py:107-121: focus branch (image + mask) or the decoder output itself
        xB = tr.focus_translation(xB4[:, :3], xa, xB4[:, 3:]) if focus else xB4
        xA = tr.focus_translation(xA4[:, :3], xa, xA4[:, 3:]) if focus else xA4
        c3, _ = tr.gen_BA.encode(xB)
        xA24 = tr.gen_BA.decode(c3, zd[2])
        xA2 = tr.focus_translation(xA24[:, :3], xB, xA24[:, 3:]) if focus else xA24
        dA = tr.dis_A(xA)
        d2 = tr.dis_2(torch.cat((xa, xA2), 1))
        fw = {"c_1": c1, "c_2": c2, "s_2": s2, "dec_AB_c1_z1": xB4, "dec_BA_c2_z2": xA4,
              "x_B_fake": xB, "x_A_fake": xA, "c_3": c3, "x_A2_fake": xA2,
              "dis_A_xA_s0": dA[0], "dis_A_xA_s1": dA[1], "dis_A_xA_s2": dA[2],
              "dis_2_pA2_s0": d2[0], "dis_2_pA2_s1": d2[1], "dis_2_pA2_s2": d2[2]}
        for k, v in fw.items():
            meta["fwd_stats"][k] = tstats(v)
            if store_tensors or v.numel() <= 16384:
                out["fw_" + k] = v.float().numpy()

    # dis_update from the initial weights (z_1..z_3 = z[0:3])
    with RandnQueue(zd[:3]):
        tr.dis_update(xa, xb, cfg)
    for n in LOSS_NAMES_DIS:
        meta["losses"][n] = float(getattr(tr, n).detach())
    ref_grads = {}
    for name in ("dis_A", "dis_B", "dis_2"):
        for k, p in getattr(tr, name).named_parameters():
            meta["grad_stats"]["dis_update/%s/%s" % (name, k)] = tstats(p.grad)
            meta["param_stats_after_dis"]["%s/%s" % (name, k)] = tstats(p)
            ref_grads[("dis", name, k)] = p.grad.clone()
            if store_tensors and p.numel() <= 2048:
                out["gd_%s/%s" % (name, k)] = p.grad.float().numpy().copy()
    # chained gen_update on the updated discriminators (train.py:71-74 order)
    with RandnQueue(zd[3:6]):
        tr.gen_update(xa, xb, cfg)
    meta["seq_losses"] = {n: float(getattr(tr, n).detach()) for n in LOSS_NAMES_GEN if hasattr(tr, n)}
    # gen_update from the initial weights (z_1..z_3 = z[3:6])
    tr, _ = fresh()
    with RandnQueue(zd[3:6]):
        tr.gen_update(xa, xb, cfg)
    for n in LOSS_NAMES_GEN:
        if hasattr(tr, n):      # (the non-focus branch sets none of the six focus attributes, trainer.py:145)
            meta["losses"][n] = float(getattr(tr, n).detach())
    for name in ("gen_AB", "gen_BA"):
        for k, p in getattr(tr, name).named_parameters():
            meta["grad_stats"]["gen_update/%s/%s" % (name, k)] = tstats(p.grad)
            meta["param_stats_after_gen"]["%s/%s" % (name, k)] = tstats(p)
            ref_grads[("gen", name, k)] = p.grad.clone()
            if store_tensors and p.numel() <= 2048:
                out["pg_%s/%s" % (name, k)] = p.detach().float().numpy().copy()
                out["gg_%s/%s" % (name, k)] = p.grad.float().numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname + ".npz"), **out)
    json.dump(meta, open(os.path.join(HERE, fname + ".json"), "w"), indent=1, sort_keys=True)

    # pin the oracle: evaluated in float64 it must reproduce the float64 reference to ~1e-9
    n64 = {k: {n: t.to(dd) for n, t in v.items()} for k, v in nets0.items()}
    orc = O.OracleTrainer(cfg, nets=n64)
    orc.dis_update(xa, xb, zd[:3])
    losses = dict(orc.losses)
    g_dis = {(n, k): t.grad for n in ("dis_A", "dis_B", "dis_2") for k, t in orc.nets[n].items()}
    orc = O.OracleTrainer(cfg, nets=n64)
    orc.gen_update(xa, xb, zd[3:6])
    losses.update(orc.losses)
    worst = max(abs(losses[n] - v) / max(1e-6, abs(v)) for n, v in meta["losses"].items())
    gworst = 0.0
    for (kind, name, k), g in ref_grads.items():
        go = g_dis[(name, k)] if kind == "dis" else orc.nets[name][k].grad
        if g.abs().max() > 1e-14:
            gworst = max(gworst, ((go - g).abs().max() / g.abs().max()).item())
    pworst = 0.0
    for name in ("gen_AB", "gen_BA"):
        for k, p in getattr(tr, name).named_parameters():
            pworst = max(pworst, (orc.nets[name][k].detach() - p.detach()).abs().max().item())
    print("%s: fp64 oracle vs fp64 reference: losses %.2e  grads %.2e  params-after-Adam %.2e"
          % (fname, worst, gworst, pworst))
    assert worst < 1e-9 and gworst < 1e-8 and pworst < 1e-9, (worst, gworst, pworst)


def op_vectors():
    """Op-level vectors from the reference's own modules."""
    g = torch.Generator().manual_seed(1234)
    out = {}
    # Conv2dBlock combos: (cin, cout, k, stride, pad, norm, act, H)
    combos = [
        (3, 8, 7, 1, 3, "none", "relu", 12), (3, 8, 7, 1, 3, "in", "relu", 12),
        (8, 16, 4, 2, 1, "in", "relu", 12), (8, 16, 4, 2, 1, "none", "lrelu", 12),
        (16, 16, 3, 1, 1, "in", "none", 8), (16, 16, 3, 1, 1, "adain", "relu", 8),
        (16, 8, 5, 1, 2, "ln", "relu", 8), (8, 4, 7, 1, 3, "none", "tanh", 8),
        (6, 8, 4, 2, 1, "none", "lrelu", 10),
    ]
    for i, (ci, co, k, s, p, norm, act, H) in enumerate(combos):
        for B in (1, 2):
            blk = ref_networks.Conv2dBlock(ci, co, k, s, p, norm=norm, activation=act, pad_type="reflect")
            w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
            b = torch.randn(co, generator=g) * 0.1
            blk.conv.weight.data.copy_(w)
            blk.conv.bias.data.copy_(b)
            x = torch.randn(B, ci, H, H + 2, generator=g)
            key = "cb%d_B%d" % (i, B)
            extra = {}
            if norm == "adain":
                aw = torch.randn(B * co, generator=g)
                ab = torch.randn(B * co, generator=g)
                blk.norm.weight, blk.norm.bias = aw, ab
                extra = {"adain_w": aw.numpy(), "adain_b": ab.numpy()}
            if norm == "ln":
                gam = torch.rand(co, generator=g)
                bet = torch.randn(co, generator=g) * 0.1
                blk.norm.gamma.data.copy_(gam)
                blk.norm.beta.data.copy_(bet)
                extra = {"gamma": gam.numpy(), "beta": bet.numpy()}
            with torch.no_grad():
                y = blk(x)
            out[key + "_x"] = x.numpy(); out[key + "_w"] = w.numpy(); out[key + "_b"] = b.numpy()
            out[key + "_y"] = y.numpy()
            for ek, ev in extra.items():
                out[key + "_" + ek] = ev
    out["cb_combos"] = np.array(json.dumps(combos))
    # upsample + 5x5 LN block as the decoder composes it (networks.py:256-257)
    up = torch.nn.Upsample(scale_factor=2)
    blk = ref_networks.Conv2dBlock(8, 4, 5, 1, 2, norm="ln", activation="relu", pad_type="reflect")
    x = torch.randn(2, 8, 6, 6, generator=g)
    with torch.no_grad():
        y = blk(up(x))
    out["up_x"] = x.numpy(); out["up_w"] = blk.conv.weight.detach().numpy(); out["up_b"] = blk.conv.bias.detach().numpy()
    out["up_gamma"] = blk.norm.gamma.detach().numpy(); out["up_beta"] = blk.norm.beta.detach().numpy()
    out["up_y"] = y.numpy()
    # avg-pool pyramid of the discriminator (networks.py:33)
    dcfg = dict(dim=4, norm="none", activ="lrelu", n_layer=4, gan_type="lsgan", num_scales=3, pad_type="reflect")
    D = ref_networks.MsImageDis(6, dcfg)
    x = torch.randn(2, 6, 64, 64, generator=g)
    with torch.no_grad():
        out["pool_x"] = x.numpy()
        out["pool_y1"] = D.downsample(x).numpy()
        out["pool_y2"] = D.downsample(D.downsample(x)).numpy()
        xo = torch.randn(1, 3, 7, 9, generator=g)
        out["pool_odd_x"] = xo.numpy()
        out["pool_odd_y"] = D.downsample(xo).numpy()
        # LSGAN target conventions of the three loss functions (networks.py:60-106)
        x2 = torch.randn(2, 6, 64, 64, generator=g)
        out["lsgan_x_fake"] = x.numpy(); out["lsgan_x_real"] = x2.numpy()
        for k, v in D.state_dict().items():
            out["lsgan_D_" + k] = v.numpy()
        out["lsgan_dis_loss"] = np.array(float(D.calc_dis_loss(x, x2)))
        out["lsgan_gen_loss"] = np.array(float(D.calc_gen_loss(x)))
        out["lsgan_gen_d2_loss"] = np.array(float(D.calc_gen_d2_loss(x, x2)))
    # focus translation and the focus losses (trainer.py:85-88, 146-158)
    cfg = reduced_config()
    tr = ref_trainer.aclgan_Trainer(cfg)
    fg = torch.randn(2, 3, 16, 16, generator=g); bg = torch.randn(2, 3, 16, 16, generator=g)
    fo = torch.tanh(torch.randn(2, 1, 16, 16, generator=g))
    out["ft_fg"] = fg.numpy(); out["ft_bg"] = bg.numpy(); out["ft_focus"] = fo.numpy()
    out["ft_y"] = tr.focus_translation(fg, bg, fo).numpy()
    np.savez_compressed(os.path.join(HERE, "op_vectors.npz"), **out)


def key_list():
    cfg = base_config()
    tr = ref_trainer.aclgan_Trainer(cfg)
    lines = []
    for name in O.OracleTrainer.NETS:
        for k, v in getattr(tr, name).state_dict().items():
            lines.append("%s %s %s" % (name, k, "x".join(str(s) for s in v.shape)))
    open(os.path.join(HERE, "state_dict_keys.txt"), "w").write("\n".join(lines) + "\n")
    # parameters() order per optimizer (Adam state indices; trainer.py:37-42)
    order = {"gen": [], "dis": []}
    for name in ("gen_AB", "gen_BA"):
        order["gen"] += ["%s/%s" % (name, k) for k, _ in getattr(tr, name).named_parameters()]
    for name in ("dis_A", "dis_B", "dis_2"):
        order["dis"] += ["%s/%s" % (name, k) for k, _ in getattr(tr, name).named_parameters()]
    json.dump(order, open(os.path.join(HERE, "param_order.json"), "w"), indent=0)


if __name__ == "__main__" and "--checkpoint-only" not in sys.argv and "--loop-only" not in sys.argv and "--plain-only" not in sys.argv:
    torch.manual_seed(0)
    op_vectors()
    key_list()
    run_step_fixture(reduced_config(), 2, 64, 64, 1, "step_reduced_64", True)
    smooth = reduced_config(); smooth["focus_epsilon"] = 0.5
    run_step_fixture(smooth, 2, 64, 64, 1, "step_reduced_64_smooth", True)
    run_step_fixture(base_config(), 1, 64, 64, 2, "step_full_64", False)
    smooth = base_config(); smooth["focus_epsilon"] = 0.5
    run_step_fixture(smooth, 1, 64, 64, 2, "step_full_64_smooth", False)
    print("golden fixtures written to", HERE)


def plain_config():
    """the non-focus configuration (trainer.py:117-121,129-130,266-276: the paper's ablation): focus_loss 0 and a 3-channel decoder"""
    cfg = reduced_config()
    cfg["focus_loss"] = 0
    cfg["gen"]["output_dim"] = 3
    return cfg


def plain_fixture():
    run_step_fixture(plain_config(), 2, 64, 64, 3, "step_reduced_64_plain", True)
    # sample() of the non-focus branch (trainer.py:216-230,238-245): 7 outputs, float32 reference
    cfg = plain_config()
    tr = ref_trainer.aclgan_Trainer(cfg)
    fill_reference(tr, cfg)
    x_a, x_b, z = seeded_inputs(2, 64, 64, 3)
    tr.z_1, tr.z_2, tr.z_3 = z[0], z[1], z[2]
    with torch.no_grad():
        outs = tr.sample(x_a, x_b)
    names = ["x_A", "x_A_fake", "x_B_fake", "x_A2_fake", "x_A_recon", "x_B", "x_B_recon"]
    assert len(outs) == 7
    np.savez_compressed(os.path.join(HERE, "sample_reduced_64_plain.npz"), **{n: o.numpy() for n, o in zip(names, outs)})
    print("plain sample fixture:", {n: tuple(o.shape) for n, o in zip(names, outs)})


def checkpoint_fixture():
    """A checkpoint WRITTEN BY THE REFERENCE (trainer.save, trainer.py:324-331) at the reduced width, after one
    dis_update + gen_update: the build's resume() must load it (tests/test_gpu_step.py) -- file names, dict
    keys, state_dict keys, OIHW layout, torch.optim.Adam state format."""
    cfg = reduced_config()
    tr = ref_trainer.aclgan_Trainer(cfg)
    fill_reference(tr, cfg)
    x_a, x_b, z = seeded_inputs(2, 64, 64, 1)
    with RandnQueue(z[:3]):
        tr.dis_update(x_a, x_b, cfg)
    with RandnQueue(z[3:6]):
        tr.gen_update(x_a, x_b, cfg)
    d = os.path.join(HERE, "ckpt_reference_reduced")
    os.makedirs(d, exist_ok=True)
    tr.save(d, 6)   # -> gen_00000007.pt, dis_00000007.pt, optimizer.pt
    # what a third step from this checkpoint must produce (float32 reference on CPU)
    with RandnQueue(z[:3]):
        tr.dis_update(x_a, x_b, cfg)
    meta = {"loss_dis_total_after_resume": float(tr.loss_dis_total.detach()),
            "files": sorted(os.listdir(d))}
    json.dump(meta, open(os.path.join(d, "expect.json"), "w"), indent=1)
    print("checkpoint fixture:", meta)


def loop_fixture():
    """Iterations of the reference's own training LOOP, not only of its two update methods: the loop body is taken from the
    reference's source text AT GENERATION TIME (train.py lines 65-104: D / G cadence on the per-epoch index `it`,
    update_learning_rate() every iteration, snapshot cadence, exit at max_iter) and executed around the imported reference trainer,
    with the data loaders replaced by fixed batches and the TensorBoard / image writers by recorders.  Nothing of that text is
    stored: the fixture holds the batches, the style noise in draw order, and per iteration which updates ran, the 16 loss
    attributes, both learning rates and the snapshot calls.  Hyper-parameters chosen so that every branch of the cadence is crossed in
    6 iterations: D_update 1, G_update 2, epochs of 3 batches (`it` restarts at iteration 3: the generator update there tells the
    per-epoch index from the global one), StepLR step_size 2 (two decays),
    snapshot every 4."""
    import contextlib
    import textwrap
    cfg = reduced_config()
    cfg.update(focus_epsilon=0.5, D_update=1, G_update=2, step_size=2, gamma=0.5, max_iter=6, log_iter=1,
               image_save_iter=10 ** 9, image_display_iter=10 ** 9, snapshot_save_iter=4)
    dd = torch.float64
    tr = ref_trainer.aclgan_Trainer(cfg)
    fill_reference(tr, cfg)
    tr = tr.double()
    g = torch.Generator().manual_seed(7)
    B, H, W, NB = 2, 64, 64, 3
    batches_a = [(torch.rand(B, 3, H, W, generator=g) * 2 - 1) for _ in range(NB)]
    batches_b = [(torch.rand(B, 3, H, W, generator=g) * 2 - 1) for _ in range(NB)]
    zs = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(3 * 12)]
    records, calls, saves = [], [], []
    orig_dis, orig_gen = tr.dis_update, tr.gen_update
    tr.dis_update = lambda *a, **k: (calls.append("dis"), orig_dis(*a, **k))[1]
    tr.gen_update = lambda *a, **k: (calls.append("gen"), orig_gen(*a, **k))[1]
    tr.save = lambda d, it: saves.append(int(it))

    def write_loss(iterations, trainer, train_writer):      # utils.py:174-178 logs every attribute that starts with loss_
        rec = {"iterations": int(iterations), "calls": list(calls),
               "lr_gen": float(trainer.gen_opt.param_groups[0]["lr"]), "lr_dis": float(trainer.dis_opt.param_groups[0]["lr"]),
               "losses": {n: float(getattr(trainer, n).detach()) for n in LOSS_NAMES_DIS + LOSS_NAMES_GEN if hasattr(trainer, n)}}
        del calls[:]
        records.append(rec)

    @contextlib.contextmanager
    def Timer(msg):
        yield

    src = open(os.path.join(REF, "train.py")).read().splitlines()
    assert src[64].startswith("while True:") and "sys.exit('Finish training')" in src[103], "reference train.py moved"
    body = textwrap.dedent("\n".join(src[64:104]))
    env = {"trainer": tr, "config": cfg, "iterations": 0, "max_iter": cfg["max_iter"], "torch": torch, "sys": sys, "Timer": Timer,
           "write_loss": write_loss, "train_writer": None, "checkpoint_directory": "unused",
           "train_loader_a": [t.to(dd) for t in batches_a], "train_loader_b": [t.to(dd) for t in batches_b]}
    sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with RandnQueue([z.to(dd) for z in zs]) as q:
            try:
                exec(compile(body, "<reference train.py:65-104>", "exec"), env)
            except SystemExit as e:
                assert "Finish training" in str(e)
            used = len(zs) - len(q.items)
    finally:
        torch.cuda.synchronize = sync
    out = {}
    for i in range(NB):
        out["x_a%d" % i] = batches_a[i].numpy(); out["x_b%d" % i] = batches_b[i].numpy()
    for i in range(used):
        out["z%d" % i] = zs[i].numpy()
    final = {}
    for name in O.OracleTrainer.NETS:
        for k, p in getattr(tr, name).named_parameters():
            final["%s/%s" % (name, k)] = tstats(p)
    meta = {"config": cfg, "B": B, "H": H, "W": W, "batches_per_epoch": NB, "n_z": used, "records": records, "saves": saves,
            "final_iterations": int(env["iterations"]), "param_stats_final": final, "dtype": "float64 reference"}
    np.savez_compressed(os.path.join(HERE, "loop_reduced_64.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "loop_reduced_64.json"), "w"), indent=1, sort_keys=True)
    print("loop fixture: %d iterations, updates per iteration %s, lr_gen %s, snapshots at %s, %d noise tensors"
          % (len(records), [r["calls"] for r in records], [r["lr_gen"] for r in records], saves, used))

    # pin the oracle + the build's loop function: the fp64 oracle driven by acl-gan_amd/train_loop.py reproduces the reference's loop
    sys.path.insert(0, os.path.join(ROOT, "acl-gan_amd"))
    import train_loop
    nets = O.test_nets(cfg, seed=cfg.get("_fill_seed", 0))
    orc = O.OracleTrainer(cfg, nets={k: {n: t.to(dd) for n, t in v.items()} for k, v in nets.items()})
    zq = [z.to(dd) for z in zs]

    class Adapter:
        def dis_update(self, a, b, hp, z=None): orc.dis_update(a, b, z)
        def gen_update(self, a, b, hp, z=None): orc.gen_update(a, b, z)
        def update_learning_rate(self): orc.update_learning_rate()
    seen = []
    train_loop.run_epochs(Adapter(), lambda: zip([t.to(dd) for t in batches_a], [t.to(dd) for t in batches_b]), cfg,
                          z_source=lambda kind: [zq.pop(0), zq.pop(0), zq.pop(0)],
                          on_iteration=lambda info: seen.append((info["ran_dis"], info["ran_gen"], dict(orc.losses), orc._lr())))
    worst = 0.0
    for rec, (rd, rg, lo, lr) in zip(records, seen):
        assert rec["calls"] == (["dis"] if rd else []) + (["gen"] if rg else []), (rec["calls"], rd, rg)
        assert abs(lr - rec["lr_gen"]) < 1e-18, (lr, rec["lr_gen"])
        for n, v in rec["losses"].items():
            worst = max(worst, abs(lo[n] - v) / max(1e-6, abs(v)))
    print("loop fixture: fp64 oracle through train_loop.run_epochs vs the reference loop: losses %.2e" % worst)
    assert len(seen) == len(records) and worst < 1e-8, worst


if __name__ == "__main__" and "--checkpoint-only" in sys.argv:
    checkpoint_fixture()
if __name__ == "__main__" and "--loop-only" in sys.argv:
    loop_fixture()
if __name__ == "__main__" and "--plain-only" in sys.argv:
    plain_fixture()
