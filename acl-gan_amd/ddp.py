"""Gradient averaging across ranks for data-parallel training (one process per GPU).

Not present in the reference (single GPU); new functionality for the 2/4/8-GPU
rows of BASELINE.json.  All normalisation layers on the path are per-sample (IN / AdaIN / custom
LN), the LSGAN / L1 losses are batch means, so averaging per-rank gradients of a batch shard is
exactly the large-batch gradient -- except for the focus 'size' loss, which squares a sum over
the *local* batch (trainer.py:149-150); per-rank evaluation + averaging is standard DDP
semantics and is what this module implements (documented in DESIGN.md).

The flat gradient buffer of one optimizer (120 MB gen / 99 MB dis in fp32) is reduced in a few
large buckets: xGMI is point-to-point (7 links x ~153 GB/s per GPU), large messages keep RCCL on
its bandwidth-optimal algorithms, and the step is O(100 ms) so latency is irrelevant.
"""
import os

import torch
import torch.distributed as dist

BUCKET_ELEMS = 16 * 1024 * 1024   # 64 MB fp32 per collective (post-backward path)
# overlapped path: smaller buckets become ready earlier.  4 M floats = 16 MB: still far above the size where an
# xGMI all-reduce is latency-bound (each GPU pushes 1/8 of a bucket to each of its 7 peers over its own link).
OVERLAP_BUCKET_ELEMS = int(os.environ.get("ACLGAN_DDP_BUCKET_ELEMS", 4 * 1024 * 1024))


def allreduce_flat(flat: torch.Tensor, world_size: int, bucket_elems: int = BUCKET_ELEMS):
    """In-place average of ``flat`` over all ranks."""
    backend = dist.get_backend()
    use_avg = backend == "nccl"   # RCCL implements ReduceOp.AVG; gloo (CPU tests) does not
    works = []
    n = flat.numel()
    for s in range(0, n, bucket_elems):
        chunk = flat[s: min(n, s + bucket_elems)]
        works.append(dist.all_reduce(chunk, op=dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    if not use_avg:
        flat.mul_(1.0 / world_size)
    return flat


def shard_batch(x: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """rank r gets samples [r*b, (r+1)*b) of a global batch."""
    b = x.shape[0] // world_size
    assert b * world_size == x.shape[0], "global batch must divide by world size"
    return x[rank * b: (rank + 1) * b]


def broadcast_flat(flat: torch.Tensor, src: int = 0, bucket_elems: int = BUCKET_ELEMS):
    """Overwrite ``flat`` on every rank with rank ``src``'s copy (replica initialisation)."""
    n = flat.numel()
    works = [dist.broadcast(flat[s: min(n, s + bucket_elems)], src, async_op=True) for s in range(0, n, bucket_elems)]
    for w in works:
        w.wait()
    return flat


class BucketReducer:
    """Gradient all-reduce overlapped with the backward pass.

    The engine (csrc/engine.hip::run_tape) knows, for the static graph of an update, which backward closure is the last
    writer of every bucket of the trained group's flat gradient buffer, and calls back on the host right after that
    closure's kernels have been enqueued.  The callback starts ``dist.all_reduce(bucket, async_op=True)``:
    ProcessGroupNCCL orders the collective (on its own RCCL stream) after everything enqueued on the compute stream so
    far and nothing enqueued later waits for it, so the exchange runs concurrently with the remaining dgrad / wgrad
    kernels.  ``finish()`` makes the compute stream wait for all of them; Adam follows.  Completion order is a function
    of the graph only, hence identical on all ranks (collectives match up).

    ctx: the aclgan_ctx handle; grad_of(group) -> that group's flat gradient tensor."""

    def __init__(self, ctx, grad_of, world_size: int, bucket_elems: int = OVERLAP_BUCKET_ELEMS):
        from . import _lib as L
        self._L = L
        self.ctx, self.grad_of, self.world = ctx, grad_of, world_size
        self.bucket_elems = int(bucket_elems)
        self.use_avg = dist.get_backend() == "nccl"   # RCCL has ReduceOp.AVG; gloo (CPU tests) sums, then scales
        self.works, self.order, self.error, self.group = [], [], None, None
        self._cb = L.BUCKET_FN(self._on_bucket)       # must outlive the registration
        L.check(L.lib.aclgan_set_grad_buckets(ctx, self.bucket_elems, self._cb, None), "set_grad_buckets")

    def close(self):
        import ctypes
        self._L.lib.aclgan_set_grad_buckets(self.ctx, 0, ctypes.cast(None, self._L.BUCKET_FN), None)

    def begin(self, group: int):
        self.works, self.order, self.error, self.group = [], [], None, group

    def _on_bucket(self, user, group, bucket, offset, numel):
        try:   # an exception must not unwind through the C frames of the engine
            chunk = self.grad_of(group)[offset: offset + numel]
            op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
            self.works.append((dist.all_reduce(chunk, op=op, async_op=True), chunk))
            self.order.append(int(bucket))
        except BaseException as e:   # noqa: BLE001
            self.error = e

    def finish(self, group: int):
        if self.error is not None:
            raise self.error
        n = self.grad_of(group).numel()
        want = (n + self.bucket_elems - 1) // self.bucket_elems
        if len(self.works) != want or sorted(self.order) != list(range(want)):
            raise RuntimeError("bucket reducer: %d of %d buckets of group %d were reduced (%s)" % (len(self.works), want, group, self.order))
        for w, chunk in self.works:
            w.wait()
            if not self.use_avg:
                chunk.mul_(1.0 / self.world)
        self.works = []
