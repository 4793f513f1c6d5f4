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
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["rccl_world_size"] == 2
    assert out["config"]["grad_allreduce"].startswith("overlapped") and out["config"]["dist_backend"] == "gloo"
    assert out["config"]["replicas_identical"] is True and out["config"]["losses_finite"] is True
    assert r.stdout.strip().splitlines()[-1].startswith("{")      # the JSON line is the last line of stdout
