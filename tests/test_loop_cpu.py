"""f-1: the training LOOP -- which update runs on which iteration, when the learning rate steps, when a snapshot
is due, when training stops -- against iterations of the reference's own loop (tests/golden/loop_reduced_64.*: reference
train.py:65-104 executed around the reference trainer by tests/golden/make_golden.py --loop-only).  CPU: the loop function of the
build (acl-gan_amd/train_loop.py, the body of train.py) drives the fp64 oracle; the HIP trainer takes the same path in
tests/test_gpu_loop.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import aclgan_oracle as O

from conftest import GOLDEN


def load_loop():
    meta = json.load(open(os.path.join(GOLDEN, "loop_reduced_64.json")))
    arr = np.load(os.path.join(GOLDEN, "loop_reduced_64.npz"))
    nb = meta["batches_per_epoch"]
    xa = [torch.from_numpy(arr["x_a%d" % i]) for i in range(nb)]
    xb = [torch.from_numpy(arr["x_b%d" % i]) for i in range(nb)]
    zs = [torch.from_numpy(arr["z%d" % i]) for i in range(meta["n_z"])]
    return meta, xa, xb, zs


class OracleAdapter:
    """the oracle behind the reference's trainer surface (explicit z)"""
    def __init__(self, cfg, dtype):
        nets = O.test_nets(cfg, seed=cfg.get("_fill_seed", 0))
        self.orc = O.OracleTrainer(cfg, nets={k: {n: t.to(dtype) for n, t in v.items()} for k, v in nets.items()})

    def dis_update(self, a, b, hp, z=None): self.orc.dis_update(a, b, z)
    def gen_update(self, a, b, hp, z=None): self.orc.gen_update(a, b, z)
    def update_learning_rate(self): self.orc.update_learning_rate()


def drive(cfg, xa, xb, zs, dtype=torch.float64):
    import aclgan_amd  # noqa: F401  (import shim for the hyphenated package directory)
    from aclgan_amd.train_loop import run_epochs, snapshot_due
    ad = OracleAdapter(cfg, dtype)
    zq = [z.to(dtype) for z in zs]
    seen, saves = [], []

    def on_iteration(info):
        seen.append({"calls": (["dis"] if info["ran_dis"] else []) + (["gen"] if info["ran_gen"] else []), "lr": ad.orc._lr(),
                     "losses": dict(ad.orc.losses), "it": info["it"]})
        if snapshot_due(info["iterations"], cfg):
            saves.append(info["iterations"])
    n = run_epochs(ad, lambda: zip([t.to(dtype) for t in xa], [t.to(dtype) for t in xb]), cfg,
                   z_source=lambda kind: [zq.pop(0) for _ in range(3)] if len(zq) >= 3 else None, on_iteration=on_iteration)
    return seen, saves, n, len(zq)


def test_loop_matches_reference_iterations():
    meta, xa, xb, zs = load_loop()
    cfg = meta["config"]
    seen, saves, n, left = drive(cfg, xa, xb, zs)
    assert n == meta["final_iterations"] == cfg["max_iter"] and left == 0
    assert saves == meta["saves"]
    assert len(seen) == len(meta["records"])
    for got, rec in zip(seen, meta["records"]):
        assert got["calls"] == rec["calls"], (rec["iterations"], got["calls"], rec["calls"])
        assert abs(got["lr"] - rec["lr_gen"]) < 1e-18 and abs(got["lr"] - rec["lr_dis"]) < 1e-18
        for k, v in rec["losses"].items():
            assert abs(got["losses"][k] - v) <= 1e-9 * max(1.0, abs(v)), (rec["iterations"], k, got["losses"][k], v)
    # the fixture crosses every branch: an iteration without gen_update, the per-epoch restart of `it` (a gen_update on the ODD global
    # iteration 3), two learning-rate decays, one snapshot
    assert [r["calls"] for r in meta["records"]] == [["dis", "gen"], ["dis"], ["dis", "gen"], ["dis", "gen"], ["dis"], ["dis", "gen"]]
    assert [r["lr_gen"] for r in meta["records"]] == [1e-4, 1e-4, 5e-5, 5e-5, 2.5e-5, 2.5e-5]


def test_loop_fixture_detects_a_swapped_cadence():
    """the test above has teeth: with D_update / G_update exchanged the call pattern no longer matches the reference's"""
    meta, xa, xb, zs = load_loop()
    cfg = dict(meta["config"]); cfg["D_update"], cfg["G_update"] = cfg["G_update"], cfg["D_update"]
    seen, _, _, _ = drive(cfg, xa, xb, zs + zs, dtype=torch.float32)
    assert [s["calls"] for s in seen] != [r["calls"] for r in meta["records"]]
    # ... and a cadence on the GLOBAL iteration index differs from the reference on iteration 3
    assert meta["records"][3]["calls"] == ["dis", "gen"] and 3 % meta["config"]["G_update"] != 0


def test_run_epochs_counts_passes_and_refuses_an_empty_pass():
    """round 5 (advisor): the pass index is reported (train.py records it next to the checkpoints so that a restart continues the loaders'
    per-epoch permutations), and a pass that yields no batch raises instead of spinning forever."""
    import pytest
    import aclgan_amd  # noqa: F401
    from aclgan_amd.train_loop import run_epochs

    class Dummy:
        def __init__(self): self.calls = []
        def dis_update(self, a, b, hp, z=None): self.calls.append("d")
        def gen_update(self, a, b, hp, z=None): self.calls.append("g")
        def update_learning_rate(self): self.calls.append("lr")
    cfg = {"D_update": 1, "G_update": 2, "max_iter": 7}
    seen = []
    tr = Dummy()
    n = run_epochs(tr, lambda: iter([(0, 0)] * 3), cfg, iterations=0, on_iteration=lambda info: seen.append((info["iterations"], info["it"], info["epoch"])), epoch0=5)
    assert n == 7
    assert seen == [(0, 0, 5), (1, 1, 5), (2, 2, 5), (3, 0, 6), (4, 1, 6), (5, 2, 6), (6, 0, 7)]
    with pytest.raises(RuntimeError, match="no batch"):
        run_epochs(Dummy(), lambda: iter([]), cfg)
