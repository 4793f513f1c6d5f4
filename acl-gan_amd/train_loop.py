"""The body of the reference's training loop (reference train.py:65-101) as an importable generator, so that the cadence -- which update
runs on which iteration, when the learning rate steps, when a snapshot is written, when training stops -- is testable against
iterations of the reference itself (tests/golden/loop_reduced_64.json, tests/test_gpu_loop.py, tests/test_oracle_golden.py).

Works with any trainer that has the reference's surface (dis_update / gen_update(x_a, x_b, hp[, z]), update_learning_rate()):
the HIP trainer (acl-gan_amd/trainer.py) and the CPU oracle (tests only)."""


def run_epochs(trainer, epoch, config, iterations=0, max_iter=None, z_source=None, on_iteration=None, epoch0=0):
    """Iterate like train.py:64-101.

    epoch()        -> iterable of (images_a, images_b): ONE pass over the zipped loaders (train.py:66); called again when exhausted
                      (the `while True` of train.py:65).  `it`, the index the update cadence is taken on, restarts with every pass.
    iterations     -> global iteration count to start from (trainer.resume's return value, train.py:64)
    z_source(kind) -> optional: the style noise of the next update (kind "dis" / "gen"), forwarded as z=...; None: the trainer draws it
    on_iteration(info) -> called after the updates of an iteration and BEFORE update_learning_rate (where the reference logs and
                      snapshots, train.py:78-99) with {"iterations": global index, "it": per-epoch index, "epoch": index of the pass
                      (epoch0 + passes completed in this call), "ran_dis", "ran_gen"}
    epoch0         -> index of the first pass (a resumed run continues its count: the loaders' per-epoch permutations are keyed by it)
    Returns the global iteration count when max_iter is reached (train.py:103-104 exits there).
    A pass that yields no batch at all (a dataset smaller than world x batch with drop_last) raises instead of spinning forever."""
    max_iter = config["max_iter"] if max_iter is None else max_iter
    n_epoch = epoch0
    while True:
        it = -1
        for it, (images_a, images_b) in enumerate(epoch()):
            ran_dis = it % config["D_update"] == 0            # train.py:71-72: on the per-epoch index
            ran_gen = it % config["G_update"] == 0            # train.py:73-74
            if ran_dis:
                if z_source is None:
                    trainer.dis_update(images_a, images_b, config)
                else:
                    trainer.dis_update(images_a, images_b, config, z=z_source("dis"))
            if ran_gen:
                if z_source is None:
                    trainer.gen_update(images_a, images_b, config)
                else:
                    trainer.gen_update(images_a, images_b, config, z=z_source("gen"))
            if on_iteration is not None:
                on_iteration({"iterations": iterations, "it": it, "epoch": n_epoch, "ran_dis": ran_dis, "ran_gen": ran_gen})
            trainer.update_learning_rate()                    # train.py:101: every iteration, whichever updates ran
            iterations += 1
            if iterations >= max_iter:                        # train.py:103-104
                return iterations
        if it < 0:
            raise RuntimeError("run_epochs: a pass over the data yielded no batch (dataset smaller than world_size x batch_size?)")
        n_epoch += 1


def snapshot_due(iterations, config):
    """train.py:97-99: `trainer.save(checkpoint_directory, iterations)` after the updates of global iteration `iterations`"""
    return (iterations + 1) % config["snapshot_save_iter"] == 0


def log_due(iterations, config):
    """train.py:77-80"""
    return (iterations + 1) % config["log_iter"] == 0
