"""The overlapped gradient all-reduce without a GPU: the engine's bucket schedule comes from a launch-free dry run
of the SAME scheduler the GPU step uses (aclgan_bucket_schedule), so order, coverage of the flat buffer and the
callback wiring of ddp.BucketReducer are testable on CPU with gloo, world size 2."""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

B, H, W = 2, 64, 64
BUCKET = 4096


def _ctx(L, reduced=True):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female.yaml")))
    if reduced:
        cfg["gen"].update(dim=8, mlp_dim=16, n_res=2)
        cfg["dis"].update(dim=8)
    from aclgan_amd.trainer import arch_from_config
    arch = arch_from_config(cfg)
    ctx = C.c_void_p()
    L.check(L.lib.aclgan_ctx_create(C.byref(arch), C.byref(ctx)))
    grads = {}
    for grp in (0, 1):
        n = L.lib.aclgan_group_numel(ctx, grp)
        grads[grp] = torch.zeros(n)
        # parameters are never dereferenced by a dry run: any non-null pointer will do
        L.check(L.lib.aclgan_bind_params(ctx, grp, L.ptr(grads[grp]), L.ptr(grads[grp]), None, None))
    return ctx, grads


def _tensor_offsets(L, ctx, grp):
    out = {}
    name = C.create_string_buffer(256)
    off = C.c_int64(); shp = (C.c_int * 4)(); nd = C.c_int()
    for i in range(L.lib.aclgan_tensor_count(ctx, grp)):
        L.check(L.lib.aclgan_tensor_info(ctx, grp, i, name, 256, C.byref(off), shp, C.byref(nd)))
        out[name.value.decode()] = off.value
    return out


def _schedule(L, ctx, grp, fire=0):
    order = (C.c_int * 4096)()
    cnt = C.c_int()
    L.check(L.lib.aclgan_bucket_schedule(ctx, grp, B, H, W, fire, order, 4096, C.byref(cnt)), "bucket_schedule")
    return [order[i] for i in range(cnt.value)]


def test_bucket_schedule_order_and_coverage():
    sys.path.insert(0, ROOT)
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib as L
    ctx, grads = _ctx(L)
    fn = C.cast(None, L.BUCKET_FN)
    L.check(L.lib.aclgan_set_grad_buckets(ctx, BUCKET, fn, None))
    for grp in (0, 1):
        n = grads[grp].numel()
        nb = (n + BUCKET - 1) // BUCKET
        order = _schedule(L, ctx, grp)
        assert sorted(order) == list(range(nb)), "every bucket exactly once"
        assert order == _schedule(L, ctx, grp), "the schedule is a function of the graph only"
        offs = _tensor_offsets(L, ctx, grp)
        pos = {b: i for i, b in enumerate(order)}
        if grp == 0:
            # reverse-backward readiness: gen_AB's content encoder runs FIRST in the forward (trainer.py:103), so the
            # bucket holding its first conv completes LAST; both decoders and MLPs (first used at trainer.py:108-109,
            # after all of x_a's encoders) complete before any bucket of that encoder
            last = offs["gen_AB/enc_content.model.0.conv.weight"] // BUCKET
            assert last in order[-2:]        # (the tensor may straddle two buckets; both complete with that closure)
            for net in ("gen_AB", "gen_BA"):
                dec = offs[net + "/dec.model.0.model.1.model.1.conv.weight"] // BUCKET
                mlp = offs[net + "/mlp.model.1.fc.weight"] // BUCKET
                enc = offs[net + "/enc_content.model.1.conv.weight"] // BUCKET
                assert pos[dec] < pos[enc] and pos[mlp] < pos[enc], (net, pos[dec], pos[mlp], pos[enc])
        else:
            # dis_update (round 5 order): dis_B runs first in the forward (it needs only x_B_fake and fills the second lane while
            # the chain to x_A2_fake is still running), then dis_A, then dis_2 -> dis_2's buckets complete first, dis_B's last.
            # Compared on layers from the middle of each network (a bucket at a network boundary holds tensors of two of them).
            mid = {net: offs[net + "/cnns.1.2.conv.weight"] // BUCKET for net in ("dis_A", "dis_B", "dis_2")}
            assert pos[mid["dis_2"]] < pos[mid["dis_A"]] < pos[mid["dis_B"]], (mid, pos)
            assert offs["dis_B/cnns.0.0.conv.weight"] // BUCKET in order[-2:]
    L.lib.aclgan_ctx_destroy(ctx)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib as L
    from aclgan_amd import ddp
    ctx, grads = _ctx(L)
    ok = True
    red = ddp.BucketReducer(ctx, lambda g: grads[g], world, bucket_elems=BUCKET)
    orders = []
    for grp in (1, 0):
        g = torch.Generator().manual_seed(10 * grp + rank)
        grads[grp].copy_(torch.randn(grads[grp].numel(), generator=g))
        want = sum(torch.randn(grads[grp].numel(), generator=torch.Generator().manual_seed(10 * grp + r)) for r in range(world)) / world
        red.begin(grp)
        # the dry run fires the callback exactly where the GPU step would: the reducer starts one async all-reduce per bucket
        order = _schedule(L, ctx, grp, fire=1)
        ok = ok and red.order == order and len(red.works) == len(order)
        red.finish(grp)
        ok = ok and torch.allclose(grads[grp], want, atol=1e-6)
        orders.append(order)
    # identical order on every rank (otherwise the collectives would not match up)
    t = torch.tensor([hash(tuple(o)) % (2 ** 31) for o in orders])
    lst = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(lst, t)
    ok = ok and all(torch.equal(lst[0], x) for x in lst)
    # a step that skips a bucket must not reach Adam silently
    red.begin(0)
    try:
        red.finish(0)
        ok = False
    except RuntimeError:
        pass
    red.close()
    # broadcast_flat: replicas start from rank 0's values
    p = torch.full((1000,), float(rank))
    ddp.broadcast_flat(p, 0, bucket_elems=300)
    ok = ok and bool((p == 0).all())
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()
    L.lib.aclgan_ctx_destroy(ctx)


def _run_world(world, port_base):
    port = port_base + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_bucket_reducer_gloo_world2():
    _run_world(2, 31500)


def test_bucket_reducer_gloo_world8():
    """the driver's 8-GPU shape: 8 ranks, the same dry-run scheduler on each, identical bucket order on all 8 (the collectives
    match up), every bucket reduced exactly once, the result is the 8-rank average"""
    _run_world(8, 34500)


def test_loader_shards_are_disjoint_and_equal_sized(tmp_path):
    """torchrun train.py: the per-rank loader shards (acl-gan_amd/data.py GpuImageLoader rank / world_size): one shared permutation
    per epoch, rank r takes slice r of every global batch -- disjoint, equal batch counts, a new permutation every epoch"""
    sys.path.insert(0, ROOT)
    import aclgan_amd  # noqa: F401
    from aclgan_amd.data import GpuImageLoader
    src = list(range(103))
    world, bs = 4, 3
    loaders = [GpuImageLoader(src, bs, True, 64, 64, 64, num_workers=1, device="cpu", rank=r, world_size=world, shard_seed=7) for r in range(world)]
    assert all(len(l) == 103 // (bs * world) for l in loaders)
    for epoch in range(2):
        per_rank = [l.batch_indices() for l in loaders]
        assert all(len(b) == len(loaders[0]) and all(len(x) == bs for x in b) for b in per_rank)
        flat = [i for b in per_rank for x in b for i in x]
        assert len(flat) == len(set(flat)) == len(loaders[0]) * bs * world        # disjoint shards, nobody sees a sample twice
        if epoch == 0:
            first = flat
        else:
            assert flat != first                                                  # reshuffled
    # world 1 keeps the reference behaviour (default generator permutation, batch_size samples per batch)
    one = GpuImageLoader(src, bs, True, 64, 64, 64, num_workers=1, device="cpu")
    assert len(one) == 103 // bs and len(one.batch_indices()) == 103 // bs
