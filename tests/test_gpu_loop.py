"""f-1 on the HIP trainer: acl-gan_amd/train_loop.py (the body of train.py) drives aclgan_Trainer over the batches and the style noise
of the reference-loop fixture (tests/golden/loop_reduced_64.*: reference train.py:65-104 around the reference trainer, float64):
per iteration the same updates run, the same learning rate is in force, and the 16 losses agree at the step-test tolerances."""
import pytest
import torch

from test_loop_cpu import load_loop

pytestmark = pytest.mark.gpu

LTOL = 1e-4          # losses, relative (6 chained fp32 Adam steps against the float64 reference; measured worst 2.2e-6)
LTOL_SIZE = 5e-3     # the *_size losses: a cancelling sum, squared


def run(cfg, xa, xb, zs):
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib as L
    from aclgan_amd.trainer import aclgan_Trainer
    from aclgan_amd.train_loop import run_epochs, snapshot_due
    from oracle import aclgan_oracle as O
    tr = aclgan_Trainer(cfg)
    nets = O.test_nets(cfg, seed=cfg.get("_fill_seed", 0))
    for name in O.OracleTrainer.NETS:
        getattr(tr, name).load_state_dict(nets[name], strict=False)
    zq = list(zs)
    seen, saves = [], []

    def on_iteration(info):
        vals = tr._losses.cpu().tolist()
        seen.append({"calls": (["dis"] if info["ran_dis"] else []) + (["gen"] if info["ran_gen"] else []), "lr": tr._current_lr(cfg),
                     "losses": {n: vals[i] for i, n in enumerate(L.LOSS_NAMES)}})
        if snapshot_due(info["iterations"], cfg):
            saves.append(info["iterations"])
    n = run_epochs(tr, lambda: zip([t.cuda() for t in xa], [t.cuda() for t in xb]), cfg,
                   z_source=lambda kind: [zq.pop(0) for _ in range(3)] if len(zq) >= 3 else None, on_iteration=on_iteration)
    return seen, saves, n, len(zq)


def test_train_loop_against_reference_iterations():
    meta, xa, xb, zs = load_loop()
    cfg = meta["config"]
    seen, saves, n, left = run(cfg, xa, xb, zs)
    assert n == meta["final_iterations"] and left == 0 and saves == meta["saves"]
    worst = 0.0
    for got, rec in zip(seen, meta["records"]):
        assert got["calls"] == rec["calls"], (rec["iterations"], got["calls"], rec["calls"])
        assert abs(got["lr"] - rec["lr_gen"]) <= 1e-12
        for k, v in rec["losses"].items():
            tol = LTOL_SIZE if k.endswith("_size") else LTOL
            err = abs(got["losses"][k] - v) / max(1e-3, abs(v))
            worst = max(worst, err if not k.endswith("_size") else 0.0)
            assert err <= tol, (rec["iterations"], k, got["losses"][k], v)
    print("train loop vs the reference's loop, 6 iterations: worst relative loss error %.2e (bound %.0e)" % (worst, LTOL))


def test_swapped_cadence_is_detected():
    meta, xa, xb, zs = load_loop()
    cfg = dict(meta["config"]); cfg["D_update"], cfg["G_update"] = cfg["G_update"], cfg["D_update"]
    seen, _, _, _ = run(cfg, xa, xb, zs + zs)
    assert [s["calls"] for s in seen] != [r["calls"] for r in meta["records"]]
    mism = [abs(s["losses"]["loss_dis_total"] - r["losses"]["loss_dis_total"]) / max(1e-3, abs(r["losses"]["loss_dis_total"])) for s, r in zip(seen, meta["records"])]
    assert max(mism) > 10 * LTOL, mism
