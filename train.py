#!/usr/bin/env python3
"""train.py -- counterpart of the reference's training loop (reference train.py:22-104) on the
MI355X-native trainer.  Same flags (--config --output_path --resume --trainer), same D/G cadence on
the per-epoch index `it` (train.py:71-74), same snapshot cadence, lr stepped every iteration
(train.py:101), exit at max_iter.

Data: when the config's data_root (or the data_folder_* / data_list_* keys) points at existing image folders,
batches come from the device input pipeline (acl-gan_amd/data.py: host decode, ONE HIP kernel per batch for
flip/Resize/crop/ToTensor/Normalize, bit-identical to the reference's torchvision/PIL chain, utils.py:43-100)
and are zipped exactly like train.py:66; with --synthetic, synthetic U(-1,1) images of the configured crop size
are used instead.  A configured dataset that does not exist is an error (as in the reference), never a silent
fallback to noise.  Out of scope: the TensorBoard/HTML writers.  Losses are printed
every log_iter iterations with ONE device->host copy of the 16-entry loss array instead of 16 (.item() each).

Data parallel (not in the reference, which is single-GPU: train.py:42): launched as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train.py --config ...
every rank binds GPU LOCAL_RANK, joins an RCCL process group, builds a full replica (aclgan_Trainer broadcasts rank 0's weights and
Adam state and hooks the overlapped gradient all-reduce), reads ITS shard of every global batch (config batch_size is PER GPU; the
loaders share one seeded permutation per epoch and rank r takes slice r of each global batch) and steps in lockstep; rank 0 alone
creates directories, copies the config, prints and writes checkpoints."""
import argparse
import os
import shutil
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def get_config(path):
    with open(path) as f:
        return yaml.safe_load(f)   # utils.get_config (utils.py:103-105) with a safe loader


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="configs/male2female.yaml", help="Path to the config file.")
    ap.add_argument("--output_path", type=str, default=".", help="outputs path")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--trainer", type=str, default="aclgan", help="aclgan")
    ap.add_argument("--synthetic", action="store_true", help="force synthetic U(-1,1) batches even if the dataset exists")
    ap.add_argument("--max_iter", type=int, default=None, help="override config max_iter")
    ap.add_argument("--seed", type=int, default=None, help="seed of the run (noise, data order): default the config's `seed`, else 0; a resumed run takes the seed and "
                    "the epoch count its checkpoint directory recorded (loop_state.json), so that the data order continues instead of restarting")
    opts = ap.parse_args()
    if opts.trainer != "aclgan":
        sys.exit("Only support aclgan")   # train.py:40-41

    # ---- one process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); plain `python train.py` = 1 GPU ----
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ACLGAN_BENCH_SHARE_GPU") == "1":      # test hook (1-GPU box): every rank on GPU 0, gloo instead of RCCL
        local_rank = 0
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        sys.exit("rank %d: GPU %d not visible (%d devices); there is no CPU fallback" % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    torch.cuda.set_device(local_rank)
    # the lane scheduler's streams before the process group's (HIP binds streams to hardware queues in creation order: include/aclgan_hip.h)
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib as _L
    if os.environ.get("ACLGAN_WARM_STREAMS", "1") not in ("", "0"):
        _L.check(_L.lib.aclgan_warm_streams(0), "warm_streams")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        backend = os.environ.get("ACLGAN_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    is_main = rank == 0

    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib as L
    from aclgan_amd.trainer import aclgan_Trainer

    config = get_config(opts.config)
    max_iter = opts.max_iter if opts.max_iter is not None else config["max_iter"]
    trainer = aclgan_Trainer(config, device="cuda:%d" % local_rank)      # world > 1: broadcasts rank 0's replica, hooks the bucket reducer
    trainer.cuda()

    model_name = os.path.splitext(os.path.basename(opts.config))[0]
    output_directory = os.path.join(opts.output_path, "outputs", model_name)
    checkpoint_directory = os.path.join(output_directory, "checkpoints")
    if is_main:
        os.makedirs(checkpoint_directory, exist_ok=True)
        shutil.copy(opts.config, os.path.join(output_directory, "config.yaml"))   # train.py:61
    if world > 1:
        dist.barrier()      # the checkpoint directory exists before any rank resumes from it

    iterations = trainer.resume(checkpoint_directory, hyperparameters=config) if opts.resume else 0
    # The run's seed and the number of completed passes over the data live next to the checkpoints (loop_state.json, written with every
    # snapshot): the per-epoch permutations of the sharded loaders are a function of (seed, epoch index), so a restart continues the
    # sequence of the original run -- partial epochs of earlier restarts included -- instead of deriving an epoch from the iteration count
    import json
    state_path = os.path.join(checkpoint_directory, "loop_state.json")
    loop_state = {}
    if opts.resume and os.path.exists(state_path):
        with open(state_path) as f:
            loop_state = json.load(f)
    seed = opts.seed if opts.seed is not None else int(loop_state.get("seed", config.get("seed", 0)))
    torch.manual_seed(seed)      # (rank 0's torch seed is what aclgan_amd.data derives the shard seeds from; now it is the same after a restart)
    B, H, W = config["batch_size"], config["crop_image_height"], config["crop_image_width"]
    epoch0 = int(loop_state.get("epoch", 0)) if opts.resume else 0
    gen = torch.Generator().manual_seed(1234 + rank)      # each rank its own shard of the synthetic global batch
    steps_per_epoch = 1000

    def synthetic_epoch():
        for _ in range(steps_per_epoch):
            yield ((torch.rand(B, 3, H, W, generator=gen) * 2 - 1).cuda(), (torch.rand(B, 3, H, W, generator=gen) * 2 - 1).cuda())

    batches_per_pass = steps_per_epoch
    if opts.synthetic:
        epoch = synthetic_epoch
        if is_main:
            print("data: synthetic U(-1,1) batches (--synthetic), %d rank(s) x batch %d" % (world, B))
    else:
        # like the reference (utils.py:43-73 -> data.py ImageFolder raises on a missing / empty folder): a mistyped
        # data_root must not silently train on noise and write checkpoints under the real model name
        root = config.get("data_root")
        folder = os.path.join(root, "trainA") if root else config.get("data_folder_train_a")
        if not (folder and os.path.isdir(folder)):
            sys.exit("training images not found (%r): fix data_root / data_folder_train_a in %s, or pass --synthetic "
                     "to train on synthetic U(-1,1) batches" % (folder, opts.config))
        from aclgan_amd.data import get_all_data_loaders
        train_loader_a, train_loader_b, _, _ = get_all_data_loaders(config, device="cuda:%d" % local_rank, rank=rank, world_size=world)   # train.py:43
        if iterations and len(train_loader_a) and len(train_loader_b):           # resumed: do not replay the permutations of the epochs already seen
            # (recorded count; a checkpoint directory written before round 5 has none: every earlier epoch is then assumed complete)
            done = int(loop_state.get("epoch", iterations // min(len(train_loader_a), len(train_loader_b))))
            train_loader_a.set_epoch(done); train_loader_b.set_epoch(done)
            epoch0 = done
        epoch = lambda: zip(train_loader_a, train_loader_b)                      # train.py:66
        batches_per_pass = max(1, min(len(train_loader_a), len(train_loader_b)))
        if is_main:
            print("data: %d / %d training images, device input pipeline, %d rank(s) x batch %d" % (len(train_loader_a.source), len(train_loader_b.source), world, B))
    from aclgan_amd.train_loop import run_epochs, snapshot_due, log_due
    clock = {"t0": time.time()}

    def on_iteration(info):      # train.py:78-99: log / snapshot, between the updates and the learning-rate step
        iterations = info["iterations"]
        if is_main and log_due(iterations, config):
            vals = trainer._losses.cpu()          # one D2H copy, implies the sync of train.py:75
            # (wall time since the previous iteration ended: the updates AND the batch's decode / transform and the learning-rate step)
            print("Iteration: %08d/%08d  %.3fs/it  " % (iterations + 1, max_iter, time.time() - clock["t0"]) +
                  " ".join("%s=%.4g" % (n[5:], float(vals[i])) for i, n in enumerate(L.LOSS_NAMES)
                           if n in ("loss_gen_total", "loss_dis_total", "loss_idt_A", "loss_gen_adv_A")))
        if is_main and snapshot_due(iterations, config):      # replicas are identical: rank 0's copy is THE checkpoint
            trainer.save(checkpoint_directory, iterations)
            # "epoch": the pass a resumed run starts with -- the NEXT one when this snapshot falls on the last batch of a pass (a resume used to
            # replay the finished pass); "it": the per-pass index of this iteration (a resume inside a pass restarts that pass's permutation)
            with open(state_path, "w") as f:
                json.dump({"seed": seed, "epoch": info["epoch"] + (1 if info["it"] + 1 >= batches_per_pass else 0), "it": info["it"],
                           "iterations": iterations + 1}, f)
        clock["t0"] = time.time()

    last = {"epoch": epoch0}
    _on = on_iteration

    def on_iteration(info):      # noqa: F811  (remember the pass index for the final snapshot)
        last["epoch"] = info["epoch"]
        _on(info)

    iterations = run_epochs(trainer, epoch, config, iterations=iterations, max_iter=max_iter, on_iteration=on_iteration, epoch0=epoch0)
    if is_main:
        trainer.save(checkpoint_directory, iterations - 1)
        with open(state_path, "w") as f:
            json.dump({"seed": seed, "epoch": last["epoch"], "iterations": iterations}, f)
        print("Finish training")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
