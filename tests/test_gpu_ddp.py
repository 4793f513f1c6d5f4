"""RCCL path of the data-parallel helper on one GPU (world_size 1): the same calls the N>1 bench makes."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_allreduce_flat_rccl_world1():
    import torch.distributed as dist
    import aclgan_amd  # noqa: F401
    from aclgan_amd import ddp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = torch.randn(3 * ddp.BUCKET_ELEMS // 8 + 5, device="cuda")
        ref = x.clone()
        ddp.allreduce_flat(x, 1, bucket_elems=ddp.BUCKET_ELEMS // 8)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
    finally:
        dist.destroy_process_group()


def test_trainer_data_parallel_world1():
    """The whole N>1 trainer path on one GPU (RCCL world size 1, ACLGAN_BENCH_FORCE_DIST=1): rank-0 broadcast of
    parameters / Adam state, the engine's bucket callback driving overlapped all-reduces, the forward sync point of the
    global-batch focus losses -- results must equal the plain single-GPU trainer's."""
    import torch.distributed as dist
    import aclgan_amd  # noqa: F401
    from aclgan_amd.trainer import aclgan_Trainer
    from oracle import aclgan_oracle as O
    cfg = O.default_config()
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1); cfg["dis"].update(dim=8); cfg["display_size"] = 1
    nets = O.test_nets(cfg, 0)
    g = torch.Generator().manual_seed(31)
    x_a = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    x_b = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    z = [torch.randn(2, 8, 1, 1, generator=g) for _ in range(6)]

    def run(tr):
        for n in O.OracleTrainer.NETS:
            getattr(tr, n).load_state_dict(nets[n], strict=False)
        tr.dis_update(x_a, x_b, cfg, z=z[:3])
        tr.gen_update(x_a, x_b, cfg, z=z[3:])
        torch.cuda.synchronize()
        # (gradients, not parameters: Adam's first step is lr * g / (|g| + eps), so an element whose gradient is ~0 may land
        #  on either side by 2 lr depending on fp32 summation order)
        return tr._grad[0].clone(), tr._grad[1].clone(), tr._losses.clone()
    ref = run(aclgan_Trainer(cfg))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29900 + os.getpid() % 300)
    os.environ["ACLGAN_BENCH_FORCE_DIST"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg2 = dict(cfg); cfg2["ddp_global_focus"] = True
        tr = aclgan_Trainer(cfg2)
        assert tr._reducer is not None and tr._zgen is not None
        got = run(tr)
        nb = [(tr._grad[g_].numel() + tr._reducer.bucket_elems - 1) // tr._reducer.bucket_elems for g_ in (0, 1)]
        assert sorted(tr._reducer.order) == list(range(nb[0]))          # the last update was gen_update: every bucket reduced once
        for a, b in zip(got, ref):
            assert (a - b).abs().max().item() <= 2e-5 * max(1e-6, b.abs().max().item())
        os.environ["ACLGAN_DDP_OVERLAP"] = "0"                          # the post-backward path gives the same result
        got2 = run(aclgan_Trainer(cfg2))
        for a, b in zip(got2, ref):
            assert (a - b).abs().max().item() <= 2e-5 * max(1e-6, b.abs().max().item())
    finally:
        os.environ.pop("ACLGAN_BENCH_FORCE_DIST", None); os.environ.pop("ACLGAN_DDP_OVERLAP", None)
        dist.destroy_process_group()


def test_bench_two_ranks_share_one_gpu():
    """bench.py's complete N=2 control flow on the 1-GPU box: two ranks launched by torch.distributed.run share GPU 0 and talk
    over gloo (RCCL refuses two ranks on one device): rank-0 broadcast, per-rank shards and z, the engine's bucket callback
    starting one async all-reduce per gradient bucket from inside the backward on BOTH ranks, barrier / max-over-ranks timing,
    one JSON line from rank 0 -- and after the steps the two replicas still hold bit-identical parameters."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.update(ACLGAN_DIST_BACKEND="gloo", ACLGAN_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "ACLGAN_BENCH_FORCE_DIST", "ACLGAN_DDP_OVERLAP"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64", "--batch", "2",
                        "--no-cpu-baseline", "--no-launch-floor"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["rccl_world_size"] == 2
    assert out["config"]["grad_allreduce"].startswith("overlapped") and out["config"]["dist_backend"] == "gloo"
    assert out["config"]["replicas_identical"] is True and out["config"]["losses_finite"] is True
    assert r.stdout.strip().splitlines()[-1].startswith("{")      # the JSON line is the last line of stdout


def _bench_shared(args, timeout=900):
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.update(ACLGAN_DIST_BACKEND="gloo", ACLGAN_BENCH_SHARE_GPU="1", ACLGAN_BENCH_TEST_WIDTH="16")      # (narrow networks: gloo reduces through the host)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "ACLGAN_BENCH_FORCE_DIST", "ACLGAN_DDP_OVERLAP"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.strip().splitlines()[-1].startswith("{")
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_eight_ranks_control_flow_and_overlap_fallback():
    """the driver's `bench.py --gpus 8` control flow on the 1-GPU box (8 ranks share GPU 0 over gloo, reduced width so that eight
    replicas fit comfortably): self-spawn, broadcast, 8 shards, bucket callbacks on every rank, max-over-ranks timing, one JSON
    line; and the --ddp-overlap 0 fallback (plain bucketed all-reduce after the backward) at 2 ranks."""
    out = _bench_shared(["--gpus", "8", "--steps", "2", "--warmup", "1", "--size", "64", "--batch", "1", "--no-cpu-baseline", "--no-launch-floor"])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 8 and out["config"]["rccl_world_size"] == 8
    assert out["config"]["grad_allreduce"].startswith("overlapped")
    assert out["config"]["replicas_identical"] is True and out["config"]["losses_finite"] is True
    assert out["config"]["ms_allreduce_exposed"] is not None and out["config"]["ms_allreduce_exposed"] >= 0.0
    assert out["roofline"]["frac"] <= 1.0
    out = _bench_shared(["--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64", "--batch", "2", "--no-cpu-baseline", "--no-launch-floor",
                         "--ddp-overlap", "0"])
    assert out["config"]["grad_allreduce"] == "after backward" and out["config"]["replicas_identical"] is True


@pytest.mark.parametrize("lanes", [1, 3])
def test_bucket_callback_fires_after_the_last_writer(lanes):
    """(round 5: under the branch-parallel replay as well -- `lanes` streams carry the backward closures, the parameter gradients are on
    their own stream; before a bucket is handed out, lane 0 = the caller's stream has joined all of them: csrc/engine.hip run_tape.)
    What the overlapped all-reduce relies on: when the engine calls back for a bucket, every kernel that writes that bucket has
    already been ENQUEUED -- a consumer ordered after the compute stream at callback time (ProcessGroupNCCL: the RCCL stream waits
    for an event recorded on the current stream when all_reduce(async_op=True) is called) reads the bucket's FINAL values.  Checked
    without a second GPU: the callback is replaced by exactly that ordering -- event on the compute stream, side stream waits for
    it, side stream snapshots the bucket -- while the backward keeps running; the gradient buffer is poisoned before the update.
    After the update every snapshot must equal the final buffer bit for bit."""
    import ctypes as C
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib as L
    from aclgan_amd.trainer import aclgan_Trainer
    from oracle import aclgan_oracle as O
    cfg = O.default_config()
    cfg["gen"].update(dim=16, mlp_dim=32, n_res=2); cfg["dis"].update(dim=16); cfg["display_size"] = 1
    g = torch.Generator().manual_seed(41)
    x_a = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    x_b = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    z = [torch.randn(2, 8, 1, 1, generator=g) for _ in range(3)]
    tr = aclgan_Trainer(cfg)
    side = torch.cuda.Stream()
    BUCKET = 32768
    fired = {"grp": None, "order": []}
    snap = {}

    def on_bucket(user, group, bucket, offset, numel):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        with torch.cuda.stream(side):
            snap[group][offset: offset + numel].copy_(tr._grad[group][offset: offset + numel], non_blocking=True)
        fired["order"].append(int(bucket))

    cb = L.BUCKET_FN(on_bucket)
    L.check(L.lib.aclgan_set_grad_buckets(tr._ctx, BUCKET, cb, None), "set_grad_buckets")
    prev_lanes = C.c_int()
    L.check(L.lib.aclgan_tuning(b"lanes", lanes, C.byref(prev_lanes)), "tuning lanes")
    try:
        for which, grp in (("dis", 1), ("gen", 0)):
            snap[grp] = torch.full_like(tr._grad[grp], float("nan"))
            tr._grad[grp].fill_(float("nan"))            # poison: zero_grad + every writer must have run before a bucket is read
            fired["order"] = []
            (tr.dis_update if which == "dis" else tr.gen_update)(x_a, x_b, cfg, z=z)
            torch.cuda.synchronize()
            nb = (tr._grad[grp].numel() + BUCKET - 1) // BUCKET
            assert sorted(fired["order"]) == list(range(nb)), (which, len(fired["order"]), nb)
            assert torch.isfinite(tr._grad[grp]).all()
            assert torch.equal(snap[grp], tr._grad[grp]), (which, int((snap[grp] != tr._grad[grp]).sum()))
    finally:
        L.lib.aclgan_set_grad_buckets(tr._ctx, 0, C.cast(None, L.BUCKET_FN), None)
        L.lib.aclgan_tuning(b"lanes", prev_lanes.value, None)


def _shard_trainers(cfg, nets, n):
    from aclgan_amd.trainer import aclgan_Trainer
    from oracle import aclgan_oracle as O
    trs = [aclgan_Trainer(cfg) for _ in range(n)]
    for tr in trs:
        for name in O.OracleTrainer.NETS:
            getattr(tr, name).load_state_dict(nets[name], strict=False)
    return trs


def test_shard_equivalence_on_one_gpu():
    """SURVEY.md 8e's correctness statement for the data-parallel step, checked without a second GPU: one trainer on a batch of 4
    versus two trainers on its halves whose gradient buffers are averaged by hand (what the RCCL AVG all-reduce does).
      * dis_update is linear in the batch: the averaged shard gradients equal the full-batch gradients.
      * gen_update with ddp_global_focus semantics (the forward sync point is fed the SUM of both shards' six mask totals -- what the
        6-float all-reduce produces): equal again, and every rank reports the global-batch focus losses.
      * gen_update without it (standard DDP semantics: each rank squares ITS OWN mask sum): visibly different.  To make that
        difference visible rather than a rounding-level effect the fixture picks focus_upper BETWEEN the two shards' mean mask
        values of f_A, so that the size term relu(sum(m - upper))^2 of that mask is active on the whole batch and on one shard but
        switched off by the relu on the other, and raises focus_delta so that the term carries weight in the total gradient.
    "Equal" = to the level at which two HIP runs at different batch sizes agree at all: the forward of a sample is not bit-identical
    between B=2 and B=4 (different split-K plans), so a few ReLU masks flip (measured 3.5e-4 relative L2 over the whole buffer)."""
    import ctypes as C
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib as L
    from oracle import aclgan_oracle as O
    cfg = O.default_config()
    cfg["gen"].update(dim=16, mlp_dim=32, n_res=2); cfg["dis"].update(dim=16); cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5        # smooth digit term
    nets = O.test_nets(cfg, 3)
    g = torch.Generator().manual_seed(43)
    x_a = torch.rand(4, 3, 128, 128, generator=g) * 2 - 1
    x_b = torch.rand(4, 3, 128, 128, generator=g) * 2 - 1
    x_a[2:] *= 0.25                   # the two shards see differently distributed images -> different mean mask values
    z = [torch.randn(4, 8, 1, 1, generator=g) for _ in range(6)]
    fulld, d0, d1, full, s0, s1, t0, t1 = _shard_trainers(cfg, nets, 8)      # (separate sets: an update also steps its optimizer)

    def sl(t, r):
        return t[2 * r: 2 * r + 2]

    # ---- dis_update: averaged shard gradients == full-batch gradients ----
    fulld.dis_update(x_a, x_b, cfg, z=z[:3])
    for r, tr in enumerate((d0, d1)):
        tr.dis_update(sl(x_a, r), sl(x_b, r), cfg, z=[sl(t, r) for t in z[:3]])
    torch.cuda.synchronize()
    avg = 0.5 * (d0._grad[1] + d1._grad[1])
    err_d = ((avg - fulld._grad[1]).double().norm() / fulld._grad[1].double().norm()).item()
    print("shard equivalence, dis_update: relative L2 of the averaged gradient buffer %.2e" % err_d)
    assert err_d <= 1e-4, err_d
    assert abs(0.5 * (float(d0.loss_dis_total) + float(d1.loss_dis_total)) - float(fulld.loss_dis_total)) <= 1e-5 * abs(float(fulld.loss_dis_total))
    del fulld, d0, d1

    # ---- the fixture: focus_upper between the shards' mean values of mask A ----
    c2, _ = full.gen_BA.encode(x_a)
    fA = full.gen_BA.decode(c2, cfg["alpha"] * z[4].cuda())[:, 3:]
    m = ((fA + 1) * 0.5).double()
    mean0, mean1 = m[:2].mean().item(), m[2:].mean().item()
    gap = abs(mean0 - mean1)
    assert gap > 1e-5, (mean0, mean1)
    cfg = dict(cfg)
    cfg["focus_upper"] = 0.5 * (mean0 + mean1) - 0.25 * gap      # global sum(m - upper) = +0.5 gap x pixels; one shard +, the other -
    cfg["focus_delta"] = 1.0 / max(gap, 1e-4)                     # weight of the size term ~ independent of how small the gap is
    cfg["focus_lower"] = 0.0                                      # masks are >= 0: the lower-bound half relu(sum(lower - m))^2 of the size loss stays off
    full.gen_update(x_a, x_b, cfg, z=z[3:])
    assert float(full.loss_gen_focus_A_size) > 0, "fixture: the size term of mask A must be active on the whole batch"

    # ---- with the global sums: pass 1 captures each shard's six local totals, pass 2 feeds every shard their SUM ----
    local, mode, cbs = {}, {"write": None}, []

    def make(rank, tr):
        def on_sync(user, ptr, n):
            off = int(ptr) - tr._ws.data_ptr()
            t = tr._ws[off: off + 4 * n].view(torch.float32)
            if mode["write"] is None:
                local[rank] = t.clone()
            else:
                t.copy_(mode["write"])
        return L.SYNC_FN(on_sync)
    for r, tr in enumerate((s0, s1)):
        cb = make(r, tr); cbs.append(cb)
        L.check(L.lib.aclgan_set_forward_sync(tr._ctx, cb, None, 2), "set_forward_sync")
        p0 = tr._param[0].clone(); m0 = tr._m[0].clone(); v0 = tr._v[0].clone(); st0 = tr._opt[0]["steps"]
        tr.gen_update(sl(x_a, r), sl(x_b, r), cfg, z=[sl(t, r) for t in z[3:]])
        torch.cuda.synchronize()
        tr._param[0].copy_(p0); tr._m[0].copy_(m0); tr._v[0].copy_(v0); tr._opt[0]["steps"] = st0      # undo pass 1's Adam step
    mode["write"] = local[0] + local[1]
    for r, tr in enumerate((s0, s1)):
        tr.gen_update(sl(x_a, r), sl(x_b, r), cfg, z=[sl(t, r) for t in z[3:]])
    torch.cuda.synchronize()
    for tr in (s0, s1):
        L.lib.aclgan_set_forward_sync(tr._ctx, C.cast(None, L.SYNC_FN), None, 1)
    ref = full._grad[0]
    err_g = ((0.5 * (s0._grad[0] + s1._grad[0]) - ref).double().norm() / ref.double().norm()).item()
    # ---- without: standard DDP semantics ----
    for r, tr in enumerate((t0, t1)):
        tr.gen_update(sl(x_a, r), sl(x_b, r), cfg, z=[sl(t, r) for t in z[3:]])
    torch.cuda.synchronize()
    err_l = ((0.5 * (t0._grad[0] + t1._grad[0]) - ref).double().norm() / ref.double().norm()).item()
    print("shard equivalence, gen_update: relative L2 of the averaged gradient buffer: %.2e with the global focus sums, %.2e with per-rank sums "
          "(mask A means %.6f / %.6f, focus_upper %.6f)" % (err_g, err_l, mean0, mean1, cfg["focus_upper"]))
    assert err_g <= 2e-3, err_g
    assert err_l >= 10 * err_g, (err_l, err_g)
    for n in ("loss_gen_focus_A_size", "loss_gen_focus_B_size", "loss_gen_focus_A2_size", "loss_gen_total"):      # global-batch values on every rank
        for tr in (s0, s1):
            assert abs(float(getattr(tr, n)) - float(getattr(full, n))) <= 2e-3 * max(1e-6, abs(float(getattr(full, n)))), (n, float(getattr(tr, n)), float(getattr(full, n)))
    # per-rank semantics: one shard's relu switched the term off
    assert min(float(t0.loss_gen_focus_A_size), float(t1.loss_gen_focus_A_size)) == 0.0
    # everything that is linear in the batch agrees in both modes: the reported batch-mean losses average exactly
    for n in ("loss_gen_adv_A", "loss_gen_adv_B", "loss_gen_adv_2", "loss_idt_A", "loss_idt_B"):
        want = float(getattr(full, n))
        got = 0.5 * (float(getattr(t0, n)) + float(getattr(t1, n)))
        assert abs(got - want) <= 2e-5 * max(1e-6, abs(want)), (n, got, want)


def test_train_py_two_ranks_share_one_gpu(tmp_path):
    """`torchrun train.py` end to end on the 1-GPU box (two ranks on GPU 0 over gloo): init_process_group, LOCAL_RANK device binding,
    rank-0-only directories / config copy / log lines / checkpoint, lockstep steps, clean exit on both ranks."""
    import subprocess
    import sys
    import yaml
    from conftest import ROOT
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female.yaml")))
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1); cfg["dis"].update(dim=8)
    cfg.update(batch_size=1, crop_image_height=64, crop_image_width=64, new_size=64, display_size=1, snapshot_save_iter=2, log_iter=1)
    cpath = os.path.join(tmp_path, "tiny.yaml")
    yaml.safe_dump(cfg, open(cpath, "w"))
    env = dict(os.environ)
    env.update(ACLGAN_DIST_BACKEND="gloo", ACLGAN_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "ACLGAN_BENCH_FORCE_DIST", "ACLGAN_DDP_OVERLAP"):
        env.pop(k, None)
    port = 23000 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "train.py"), "--config", cpath, "--output_path", str(tmp_path),
                        "--synthetic", "--max_iter", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ck = os.path.join(tmp_path, "outputs", "tiny", "checkpoints")
    # (loop_state.json, round 5: the run's seed and pass index next to the checkpoints, so that a restart continues the data order)
    assert sorted(os.listdir(ck)) == ["dis_00000002.pt", "dis_00000003.pt", "gen_00000002.pt", "gen_00000003.pt", "loop_state.json", "optimizer.pt"]
    import json
    with open(os.path.join(ck, "loop_state.json")) as f:
        state = json.load(f)
    assert state["iterations"] == 3 and state["seed"] == 0 and state["epoch"] == 0, state
    assert r.stdout.count("Iteration: 00000001/") == 1 and r.stdout.count("Finish training") == 1      # rank 0 alone prints
