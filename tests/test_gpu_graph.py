"""HIP-graph replay of the update launch sequence (aclgan_Trainer(hip_graph=True), SURVEY section 7 step 7 "static launch
sequence"): the same kernels in the same order, so a graph-replayed run must reproduce the eager run -- bitwise in deterministic
mode, to summation-order accuracy otherwise.  The reference has no counterpart (eager PyTorch)."""
import pytest
import torch

from oracle import aclgan_oracle as O

pytestmark = pytest.mark.gpu


def _run(T, cfg, nets, batches, **kw):
    tr = T.aclgan_Trainer(cfg, **kw)
    for name in O.OracleTrainer.NETS:
        getattr(tr, name).load_state_dict(nets[name], strict=False)
    losses = []
    for x_a, x_b, z in batches:
        tr.dis_update(x_a, x_b, cfg, z=z[:3])
        tr.gen_update(x_a, x_b, cfg, z=z[3:])
        tr.update_learning_rate()
        losses.append((float(tr.loss_dis_total), float(tr.loss_gen_total)))
    torch.cuda.synchronize()
    params = {k: v.detach().clone() for k, v in tr.named_parameters()}
    return tr, losses, params


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_graph_replay_matches_eager(dtype):
    assert torch.cuda.is_available()
    import aclgan_amd  # noqa: F401
    from aclgan_amd import trainer as T, _lib as L
    cfg = O.default_config()
    cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5
    nets = O.test_nets(cfg, 0)
    g = torch.Generator().manual_seed(31)
    B, S = 2, 64
    batches = [(torch.rand(B, 3, S, S, generator=g) * 2 - 1, torch.rand(B, 3, S, S, generator=g) * 2 - 1,
                [torch.randn(B, 8, 1, 1, generator=g) for _ in range(6)]) for _ in range(4)]     # step 1 eager, 2 captures, 3-4 replay
    prev = L.lib.aclgan_get_deterministic()
    try:
        te, le, pe = _run(T, cfg, nets, batches, compute_dtype=dtype, deterministic=True)
        tg, lg, pg = _run(T, cfg, nets, batches, compute_dtype=dtype, deterministic=True, hip_graph=True)
    finally:
        L.check(L.lib.aclgan_set_deterministic(prev))
    assert tg.hip_graph, "capture fell back to eager execution"
    assert tg._graphs["gen"]["graph"] is not None and tg._graphs["dis"]["graph"] is not None
    assert le == lg, (le, lg)
    bad = [k for k in pe if not torch.equal(pe[k], pg[k])]
    assert not bad, bad[:6]
    # a different batch size re-captures instead of replaying a stale graph
    x_a, x_b, z = batches[0]
    tg.dis_update(x_a[:1], x_b[:1], cfg, z=[t[:1] for t in z[:3]])
    assert torch.isfinite(tg.loss_dis_total).item()
