"""Round 5: the branch-parallel schedule of the step (csrc/engine.hip "Lanes") and the batched small launches.

The reference's step (trainer.py:99-169, 254-292) is a graph with wide independent branches -- two translation directions, the
reconstruction decodes, three discriminators with three scales each.  The engine spreads them over 1 .. 4 HIP streams ("lanes") but
builds and replays everything in ONE host order, with the parameter gradients on one ordered stream: the number of lanes must not
change a single bit of any loss, gradient or parameter.  A missing cross-lane dependency shows up here as a bitwise difference.

  * deterministic mode: losses and every gradient tensor of dis_update + gen_update with 2, 3, 4 lanes == 1 lane, bitwise (fp32, bf16);
  * batched Winograd filter transforms (one launch per network part) == per-filter transforms, bitwise, and they remove the launches;
  * the batched LSGAN launch == the per-term operator calls, bitwise (loss values and loss gradients);
  * default mode: the same comparison within the tolerance of the atomics' summation order;
  * error path: an injected failure in the middle of the backward replay leaves nothing in flight -- the same trainer then produces
    the bits of a fresh one.
(The data-parallel bucket callback under 1 / 2 / 4 lanes: tests/test_gpu_ddp.py::test_bucket_callback_fires_after_the_last_writer.)
"""
import ctypes as C

import pytest
import torch

from oracle import aclgan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib
    return _lib


def _tune(L, key, value):
    prev = C.c_int()
    L.check(L.lib.aclgan_tuning(key, value, C.byref(prev)), "aclgan_tuning")
    return prev.value


def _fixture(S=128, B=2, seed=21, narrow=False):
    cfg = O.default_config()
    cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5
    if narrow:
        cfg["gen"].update(dim=16, mlp_dim=32, n_res=2); cfg["dis"].update(dim=16)
    nets = O.test_nets(cfg, 0)
    g = torch.Generator().manual_seed(seed)
    x_a = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    x_b = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    z = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(6)]
    return cfg, nets, x_a, x_b, z


def _same(a, b):
    for which in ("dis", "gen"):
        la, ga = a[which]; lb, gb = b[which]
        assert la and ga
        assert la == lb, (which, {k: (la[k], lb[k]) for k in la if la[k] != lb[k]})
        diff = [k for k in ga if not torch.equal(ga[k], gb[k])]
        assert not diff, (which, len(diff), diff[:8])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_lane_count_does_not_change_a_bit(L, dtype):
    from aclgan_amd import trainer as T
    from test_gpu_determinism import _step
    cfg, nets, x_a, x_b, z = _fixture()
    prev_det = L.lib.aclgan_get_deterministic()
    prev = _tune(L, b"lanes", 1)
    try:
        one = _step(T, cfg, nets, x_a, x_b, z, dtype, True)
        for lanes in (2, 3, 4, 3):        # (the default, 3, twice: a race is not obliged to show on the first try)
            _tune(L, b"lanes", lanes)
            _same(one, _step(T, cfg, nets, x_a, x_b, z, dtype, True))
    finally:
        _tune(L, b"lanes", prev)
        L.check(L.lib.aclgan_set_deterministic(prev_det))


def test_lane_count_default_mode_within_atomics_noise(L):
    """default mode (fp32 atomics in the halo / split-K paths): not bitwise, but the same numbers"""
    from aclgan_amd import trainer as T
    from test_gpu_determinism import _step
    cfg, nets, x_a, x_b, z = _fixture(S=64, narrow=True)
    prev = _tune(L, b"lanes", 1)
    try:
        one = _step(T, cfg, nets, x_a, x_b, z, "fp32", False)
        _tune(L, b"lanes", 2)
        two = _step(T, cfg, nets, x_a, x_b, z, "fp32", False)
    finally:
        _tune(L, b"lanes", prev)
    for which in ("dis", "gen"):
        la, ga = one[which]; lb, gb = two[which]
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-5 * max(1e-3, abs(la[k])), (k, la[k], lb[k])
        gmax = max(float(t.double().norm()) for t in ga.values())
        for k in ga:
            err = (ga[k].double() - gb[k].double()).norm().item() / (ga[k].double().norm().item() + 1e-3 * gmax)
            assert err <= 1e-4, (which, k, err)


def test_batched_filter_transforms_same_bits_fewer_launches(L):
    from aclgan_amd import trainer as T
    from test_gpu_determinism import _step
    cfg, nets, x_a, x_b, z = _fixture()
    prev_det = L.lib.aclgan_get_deterministic()
    prev = _tune(L, b"u_batch", 0)
    try:
        n0 = L.lib.aclgan_launch_count()
        per_filter = _step(T, cfg, nets, x_a, x_b, z, "fp32", True)
        n1 = L.lib.aclgan_launch_count()
        _tune(L, b"u_batch", 1)
        batched = _step(T, cfg, nets, x_a, x_b, z, "fp32", True)
        n2 = L.lib.aclgan_launch_count()
    finally:
        _tune(L, b"u_batch", prev)
        L.check(L.lib.aclgan_set_deterministic(prev_det))
    _same(per_filter, batched)
    # 4 ResBlocks x 2 convolutions per encoder / decoder: forward transforms in both updates, input-gradient transforms in gen_update --
    # 8 filters per batched launch
    print("launches of (dis_update + gen_update): per-filter transforms %d, batched %d" % (n1 - n0, n2 - n1))
    assert (n1 - n0) - (n2 - n1) >= 60, (n1 - n0, n2 - n1)


def test_fused_mlp_forward_same_bits_fewer_launches(L):
    """round 6: the generator's MLP forward as one launch (aclgan_mlp3_fwd) against three aclgan_linear_fwd launches: the same bits in every loss,
    gradient and parameter of a step, 2 launches fewer per decode"""
    from aclgan_amd import trainer as T
    from test_gpu_determinism import _step
    cfg, nets, x_a, x_b, z = _fixture()
    prev_det = L.lib.aclgan_get_deterministic()
    prev = _tune(L, b"mlp_fused", 0)
    try:
        n0 = L.lib.aclgan_launch_count()
        three = _step(T, cfg, nets, x_a, x_b, z, "fp32", True)
        n1 = L.lib.aclgan_launch_count()
        _tune(L, b"mlp_fused", 1)
        one = _step(T, cfg, nets, x_a, x_b, z, "fp32", True)
        n2 = L.lib.aclgan_launch_count()
    finally:
        _tune(L, b"mlp_fused", prev)
        L.check(L.lib.aclgan_set_deterministic(prev_det))
    _same(three, one)
    print("launches of (dis_update + gen_update): three-launch MLP %d, fused %d" % (n1 - n0, n2 - n1))
    assert (n1 - n0) - (n2 - n1) >= 16, (n1 - n0, n2 - n1)      # (4 decodes in dis_update + 5 in gen_update, 2 launches each)


def test_lsgan_batch_equals_the_operator(L):
    """One launch for all (scale, segment) terms of a discriminator call (aclgan_lsgan_loss_multi = the engine's lsgan_loss_batch;
    networks.py:64-67,81-83,96-98) against the per-term operator: loss slots and loss gradients bit for bit, including several terms
    adding into one slot (the joint batches: dis_A on x_A_fake | x_A2_fake | x_a) and more terms than one launch holds."""
    g = torch.Generator().manual_seed(5)
    shapes = [3 * 16 * 16, 3 * 16 * 16, 8 * 8, 8 * 8, 4 * 4, 17, 1, 2 * 16 * 16, 100, 100, 5, 7, 300, 33]      # 14 terms > LSGAN_MAX_TERMS (12)
    maps = [torch.randn(n, generator=g).cuda() for n in shapes]
    targets = [float(i % 2) for i in range(len(shapes))]
    weights = [0.5 if i % 3 == 0 else 1.0 for i in range(len(shapes))]
    gscales = [1.0 if i % 4 else 0.2 for i in range(len(shapes))]
    slot_of = [i % 3 for i in range(len(shapes))]
    slots_a = torch.zeros(3, device="cuda"); slots_b = torch.zeros(3, device="cuda")
    grads_a = [torch.full_like(m, float("nan")) for m in maps]; grads_b = [torch.full_like(m, float("nan")) for m in maps]
    st = L.stream_ptr()
    for i, m in enumerate(maps):
        L.check(L.lib.aclgan_lsgan_loss(L.ptr(m), m.numel(), targets[i], weights[i], C.c_void_p(slots_a.data_ptr() + 4 * slot_of[i]),
                                        L.ptr(grads_a[i]), gscales[i], st), "lsgan_loss")
    n = len(maps)
    arr = lambda vals: (C.c_void_p * n)(*vals)      # noqa: E731
    L.check(L.lib.aclgan_lsgan_loss_multi(arr([m.data_ptr() for m in maps]), (C.c_int * n)(*[m.numel() for m in maps]),
                                          (C.c_float * n)(*targets), (C.c_float * n)(*weights),
                                          arr([slots_b.data_ptr() + 4 * slot_of[i] for i in range(n)]),
                                          arr([gb.data_ptr() for gb in grads_b]), (C.c_float * n)(*gscales), n, st), "lsgan_loss_multi")
    torch.cuda.synchronize()
    assert torch.equal(slots_a, slots_b), (slots_a, slots_b)
    for ga, gb in zip(grads_a, grads_b):
        assert torch.equal(ga, gb)
    ref = [0.0, 0.0, 0.0]
    for i, m in enumerate(maps):
        ref[slot_of[i]] += weights[i] * float(((m.double() - targets[i]) ** 2).mean())
    for k in range(3):
        assert abs(float(slots_b[k]) - ref[k]) <= 1e-5 * abs(ref[k])


def test_injected_backward_fault_leaves_nothing_in_flight(L):
    """aclgan_tuning("fault_at", k): the replay of the backward fails after closure k -- with lanes and the parameter-gradient stream
    holding half an update.  The call must return the error only after those streams have drained (csrc/engine.hip run_tape ->
    lanes_quiesce): the SAME trainer, reset to the same state, then reproduces a fresh trainer bit for bit."""
    from aclgan_amd import trainer as T
    cfg, nets, x_a, x_b, z = _fixture(S=64, narrow=True)
    prev_det = L.lib.aclgan_get_deterministic()

    def fresh():
        tr = T.aclgan_Trainer(cfg, deterministic=True)
        for name in O.OracleTrainer.NETS:
            getattr(tr, name).load_state_dict(nets[name], strict=False)
        return tr
    try:
        ref = fresh()
        ref.dis_update(x_a, x_b, cfg, z=z[:3]); ref.gen_update(x_a, x_b, cfg, z=z[3:])
        torch.cuda.synchronize()
        tr = fresh()
        for k in (0, 7, 40, 100):      # (this narrow gen_update replays ~146 closures)
            _tune(L, b"fault_at", k)
            with pytest.raises(L.AclganError, match="injected fault"):
                tr.gen_update(x_a, x_b, cfg, z=z[3:])
        _tune(L, b"fault_at", -1)
        torch.cuda.synchronize()
        tr2 = fresh()          # (the failed updates touched gradients and possibly Adam state of `tr`: compare a clean replay instead, sharing the process)
        tr2.dis_update(x_a, x_b, cfg, z=z[:3]); tr2.gen_update(x_a, x_b, cfg, z=z[3:])
        torch.cuda.synchronize()
        assert torch.equal(ref._param[0], tr2._param[0]) and torch.equal(ref._param[1], tr2._param[1])
        # and the trainer that saw the faults still works: its next update runs and gives finite losses
        tr.dis_update(x_a, x_b, cfg, z=z[:3])
        torch.cuda.synchronize()
        assert torch.isfinite(torch.tensor(float(tr.loss_dis_total)))
    finally:
        _tune(L, b"fault_at", -1)
        L.check(L.lib.aclgan_set_deterministic(prev_det))


def test_forward_only_call_after_an_update_is_not_refused(L):
    """advisor, round 5: the arena check of a forward-only call (encode / decode / sample: no side stack, an arena sized by
    aclgan_forward_workspace_bytes) compared against the side-stream high-water mark the PREVIOUS update had left behind, so a trained
    context refused an inference call at a new shape with a spurious ACLGAN_ENOMEM.  fwd_begin and the dry runs now reset that mark."""
    from aclgan_amd.trainer import aclgan_Trainer
    cfg, nets, x_a, x_b, z = _fixture(S=128, B=2)
    tr = aclgan_Trainer(cfg)
    for name in O.OracleTrainer.NETS:
        getattr(tr, name).load_state_dict(nets[name], strict=False)
    tr.dis_update(x_a, x_b, cfg, z=z[:3]); tr.gen_update(x_a, x_b, cfg, z=z[3:])
    torch.cuda.synchronize()
    # a NEW shape: the trainer binds an arena of exactly the forward-only size (+ nothing for a side stack)
    tr._ws = None; tr._ws_shape = None
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(1, 3, 192, 160, generator=g) * 2 - 1).cuda()
    c, s = tr.gen_AB.encode(x)
    y = tr.gen_AB.decode(c, s)
    outs = tr.dis_A(x)
    torch.cuda.synchronize()
    assert y.shape == (1, 4, 192, 160) and len(outs) == 3 and torch.isfinite(y).all()
    with torch.no_grad():      # (the two updates have moved the parameters: the oracle gets the trainer's current ones)
        now = {k: v.detach().cpu() for k, v in tr.gen_AB.state_dict().items()}
        c_ref, s_ref = O.gen_encode(now, x.cpu(), cfg["gen"])
    assert ((c.cpu() - c_ref).abs().max() / c_ref.abs().max()).item() < 1e-4


def test_trainer_sizes_its_arena_again_after_a_tuning_change(L):
    """advisor, round 5: the trainer cached its arena by shape only; after aclgan_tuning raised the need of an update (more lanes) every
    update failed with ACLGAN_ENOMEM.  The cache is now keyed by the tuning epoch (aclgan_tuning_get "epoch")."""
    from aclgan_amd.trainer import aclgan_Trainer
    cfg, nets, x_a, x_b, z = _fixture(S=64, B=2, narrow=True)
    old = _tune(L, b"lanes", 1)
    try:
        tr = aclgan_Trainer(cfg)
        for name in O.OracleTrainer.NETS:
            getattr(tr, name).load_state_dict(nets[name], strict=False)
        tr.dis_update(x_a, x_b, cfg, z=z[:3])
        n1 = tr._ws.numel()
        v = C.c_longlong(); L.check(L.lib.aclgan_tuning_get(b"lanes", C.byref(v)), "tuning_get"); assert v.value == 1
        _tune(L, b"lanes", 3)
        tr.dis_update(x_a, x_b, cfg, z=z[:3]); tr.gen_update(x_a, x_b, cfg, z=z[3:])      # (ACLGAN_ENOMEM before the fix)
        torch.cuda.synchronize()
        assert tr._ws.numel() >= n1
        L.check(L.lib.aclgan_tuning_get(b"lanes", C.byref(v)), "tuning_get"); assert v.value == 3
        assert L.lib.aclgan_tuning_get(b"no_such_key", C.byref(v)) != 0
    finally:
        _tune(L, b"lanes", old)
