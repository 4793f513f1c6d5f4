"""Backward parity with the activation masks HELD FIXED (round 6).

The step-level gradient tests (tests/test_gpu_fullsize.py: every gradient tensor within 1e-2 relative L2 of the fp32 oracle, measured
5 - 6e-3) cannot tell a sloppy backward kernel from the "mask lottery": two correct fp32 implementations take different branches of a
ReLU / LeakyReLU wherever a pre-activation lies within forward rounding (~1e-5) of zero, and every flipped element moves all upstream
gradients.  Here the lottery is taken out: the HIP update records the masks it ran with (aclgan_debug_capture_masks: output > 0 of every
Conv2dBlock it back-propagates through, plus the step's two other sign decisions: |m - 0.5| of the focus digit losses and |x_recon - x| of
the identity losses), the oracle's autograd replays the same update with THOSE masks in place of its own (oracle.act_masks), and what is left is the error of the backward kernels themselves (summation order, Winograd transforms, atomics):

    fp32   every gradient tensor <= 1e-3 relative L2 -- measured 1.3e-5 (gen_update) / 2.5e-6 (dis_update) at 256x256 B=2, against 7.2e-3 /
           4.3e-4 un-frozen in the same run: 398 of 2.5e8 mask elements and ONE of 1.2e6 loss signs differed
    bf16 / fp16 against the EMULATED 16-bit contract (oracle.compute_dtype) with the masks frozen: per network bounds below.

Blocks are matched between the two implementations by CONTENT, not by order (the engine builds the passes of an update in its lane order
and runs the discriminators on joint batches): a recorded mask (split into the update's batch-sized chunks) belongs to the oracle
activation of the same shape it agrees with on >= 99 % of the elements -- unrelated passes agree on ~50 %."""
import ctypes as C
import os

import pytest
import torch

from oracle import aclgan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    import aclgan_amd  # noqa: F401
    from aclgan_amd import trainer
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    return trainer


def _make(T, cfg, nets, dt=None):
    tr = T.aclgan_Trainer(cfg, compute_dtype=dt) if dt else T.aclgan_Trainer(cfg)
    for name in O.OracleTrainer.NETS:
        getattr(tr, name).load_state_dict(nets[name], strict=False)
    return tr


def _inputs(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    x_a = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    x_b = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    z = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(6)]
    return x_a, x_b, z


def _hip_update_with_masks(tr, which, x_a, x_b, cfg, z, B, cap_bytes):
    """run one update with the mask recording on; returns the masks as NCHW bool CPU tensors in batch-sized chunks"""
    from aclgan_amd import _lib as L
    buf = torch.zeros(cap_bytes, dtype=torch.uint8, device="cuda")
    L.check(L.lib.aclgan_debug_capture_masks(tr._ctx, L.ptr(buf), cap_bytes), "debug_capture_masks")
    try:
        (tr.dis_update if which == "dis" else tr.gen_update)(x_a, x_b, cfg, z=z)
        torch.cuda.synchronize()
        chunks, signs = [], []
        dims = (C.c_int * 4)(); off = C.c_longlong(); act = C.c_int()
        for i in range(L.lib.aclgan_debug_mask_count(tr._ctx)):
            L.check(L.lib.aclgan_debug_mask_info(tr._ctx, i, dims, C.byref(off), C.byref(act)), "debug_mask_info")
            b, h, w, c = list(dims)
            m = buf[off.value: off.value + b * h * w * c].view(b, h, w, c).permute(0, 3, 1, 2).bool().cpu()
            if act.value == 100:        # sign of a focus mask's (m - 0.5): channel 3 of the decoder output (2 m - 1)
                signs.append(m[:, 3:4].contiguous())
            elif act.value == 101:      # sign of (x_recon - x) of an identity loss
                signs.append(m.contiguous())
            else:
                assert b % B == 0, (b, B)
                chunks.extend(m[j:j + B] for j in range(0, b, B))
    finally:
        L.check(L.lib.aclgan_debug_capture_masks(tr._ctx, None, 0), "debug_capture_masks(off)")
    return chunks, signs


def _match(recorded, chunks):
    """replay dictionary {oracle activation index: the HIP mask of the same block}; also the flip statistics"""
    by_shape = {}
    for ch in chunks:
        by_shape.setdefault(tuple(ch.shape), []).append(ch)
    replay, flips, total, unmatched = {}, 0, 0, []
    for i, own in enumerate(recorded):
        cands = by_shape.get(tuple(own.shape), [])
        best, best_agree = None, 0.0
        sub = own.flatten()[::13]
        for ch in cands:
            a = (ch.flatten()[::13] == sub).float().mean().item()
            if a > best_agree:
                best, best_agree = ch, a
        if best is not None and best_agree >= 0.99:
            replay[i] = best
            flips += int((best != own).sum()); total += own.numel()
        else:
            unmatched.append((i, tuple(own.shape), round(best_agree, 3)))
    return replay, flips, total, unmatched


def _grad_errors(tr, orc, nets_, scale=1.0, floor=1e-3):
    gmax = max(float(t.grad.norm()) for n in nets_ for t in orc.nets[n].values())
    out = []
    for n in nets_:
        for k, gr in getattr(tr, n).named_grads():
            ref = orc.nets[n][k].grad.double()
            rn = ref.norm().item()
            if rn >= floor * gmax:       # (tensors that are exactly zero in exact arithmetic -- biases in front of Instance / AdaIN norms -- are the fullsize test's business)
                out.append(((gr.cpu().double() / scale - ref).norm().item() / rn, n, k))
    out.sort(reverse=True)
    return out


def _frozen_and_free(T, dt, B, S, seed, forced=False):
    """forced: every eligible convolution through the one-launch Winograd kernel (tuning wino_fused = 2): at B = 2 the cost models keep the
    4x4 stride-2 layers and the small grids on the direct kernels / the pipeline, so the default run does not reach those kernels"""
    from aclgan_amd import _lib as L
    old = L.lib.aclgan_set_tuning(b"wino_fused", 2) if forced else None
    try:
        return _frozen_and_free_impl(T, dt, B, S, seed)
    finally:
        if forced:
            L.lib.aclgan_set_tuning(b"wino_fused", old)


def _frozen_and_free_impl(T, dt, B, S, seed):
    cfg = O.default_config()
    cfg["display_size"] = 1
    cfg["focus_epsilon"] = 0.5      # smooth fixture (tests/golden/make_golden.py: the default 0.01 has a sign discontinuity of 1e4 at m = 0.5)
    nets = O.test_nets(cfg, 0)
    x_a, x_b, z = _inputs(B, S, seed)
    scale = 65536.0 if dt == "fp16" else 1.0

    class ctx:      # the oracle under the build's arithmetic contract (plain fp32, or the emulated 16-bit contract)
        def __enter__(self):
            self.c = O.compute_dtype(dt, loss_scale=scale) if dt else None
            return self.c.__enter__() if self.c else None

        def __exit__(self, *a):
            return self.c.__exit__(*a) if self.c else None

    res = {}
    for which, zz, nets_ in (("dis", z[:3], ("dis_A", "dis_B", "dis_2")), ("gen", z[3:], ("gen_AB", "gen_BA"))):
        tr = _make(T, cfg, nets, dt)
        if dt:
            assert tr.grad_scale() == scale
        chunks, signs = _hip_update_with_masks(tr, which, x_a, x_b, cfg, zz, B, 2 << 30 if S >= 256 else 1 << 29)
        assert chunks, "nothing recorded"
        assert len(signs) == (5 if which == "gen" else 0), len(signs)      # focus B, A, A2 + identity A, B: the oracle's call order (gen_losses)
        with ctx(), O.act_masks() as rec:                    # the oracle with its OWN masks
            free = O.OracleTrainer(cfg, nets=nets)
            (free.dis_update if which == "dis" else free.gen_update)(x_a, x_b, zz, apply=False)
        replay, flips, total, unmatched = _match(rec.recorded, chunks)
        sflips = sum(int((a != b).sum()) for a, b in zip(signs, rec.signs))
        with ctx(), O.act_masks(replay, dict(enumerate(signs))) as rec2:             # ... and with the masks / signs of the HIP update
            frozen = O.OracleTrainer(cfg, nets=nets)
            (frozen.dis_update if which == "dis" else frozen.gen_update)(x_a, x_b, zz, apply=False)
        assert len(rec2.recorded) == len(rec.recorded)
        e_free, e_frozen = _grad_errors(tr, free, nets_, scale), _grad_errors(tr, frozen, nets_, scale)
        print("%s %s_update @%dx%d B=%d: %d of %d oracle activations matched to a recorded mask (%d recorded chunks), %d of %d mask elements differ (%.2e)"
              % (dt or "fp32", which, S, S, B, len(replay), len(rec.recorded), len(chunks), flips, total, flips / max(1, total)))
        print("   sign decisions of the focus digit / identity losses that differ: %d of %d" % (sflips, sum(a.numel() for a in signs)))
        print("   unmatched oracle activations (no gradient passes through them in the HIP update):", unmatched[:8], "..." if len(unmatched) > 8 else "")
        print("   worst gradient tensors, masks FROZEN:", [("%.2e" % e, n, k) for e, n, k in e_frozen[:4]])
        print("   worst gradient tensors, masks free  :", [("%.2e" % e, n, k) for e, n, k in e_free[:4]])
        res[which] = dict(frozen=e_frozen, free=e_free, matched=len(replay), acts=len(rec.recorded), unmatched=unmatched)
    return res


def _per_net(errs):
    out = {}
    for e, n, k in errs:
        key = n if n.startswith("dis") else n + (".enc" if k.startswith("enc_") else ".dec")
        out[key] = max(out.get(key, 0.0), e)
    return out


@pytest.mark.parametrize("forced", [False, True], ids=["default-paths", "fused-winograd-forced"])
def test_backward_parity_with_frozen_masks_fp32(T, forced):
    """256x256 B=2, full width: with the HIP update's own ReLU / LeakyReLU masks replayed by the oracle every gradient tensor agrees to 1e-3
    relative L2 (the bound of tests/test_gpu_fullsize.py on the un-frozen comparison stays 1e-2).  Second case: the one-launch Winograd kernel
    FORCED on every eligible layer -- the 3x3 ResBlock layers, the sub-pixel phases and (round 6) the four parity phases of the 4x4 stride-2
    layers, forward and input gradient -- the kernels the benchmarked batch runs but B = 2 would not reach."""
    res = _frozen_and_free(T, None, 2, 256, 31, forced=forced)
    for which in ("dis", "gen"):
        r = res[which]
        # every activation a gradient passes through was matched (dis_update: the generator pass is forward-only in the HIP update)
        assert r["matched"] >= (0.3 if which == "dis" else 0.9) * r["acts"], (which, r["matched"], r["acts"], r["unmatched"][:6])
        assert r["frozen"][0][0] <= 1e-3, (which, r["frozen"][:6])       # (measured 1.3e-5 / 2.5e-6: the bound of the review; 1e-4 would hold)
        assert r["free"][0][0] <= 1e-2, (which, r["free"][:6])


# 16-bit: bounds per network group <= 2x the measured value, masks and signs frozen, against the emulated contract.  Measured (round 6,
# profiles/r06_experiments.md): bf16 dis 7.8e-3, gen.dec 1.82e-2, gen.enc 1.63e-2 (un-frozen: 5.6e-2 / 1.96e-1); fp16 dis 9.2e-4, gen.dec
# 2.4e-3, gen.enc 2.2e-3 (un-frozen: 2.0e-2 / 7.0e-2) -- what remains is the rounding-flip noise of the 16-bit values themselves
# (tests/test_gpu_step16.py docstring), 10x below the un-frozen figures the round-5 bounds (3e-1 / 1.2e-1) had to admit.
ETOL_FROZEN = {
    "bf16": {"dis": 1.5e-2, "gen.dec": 3e-2, "gen.enc": 3e-2},
    "fp16": {"dis": 2e-3, "gen.dec": 5e-3, "gen.enc": 5e-3},
}


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_backward_parity_with_frozen_masks_16bit(T, dt):
    """the same construction for the 16-bit paths against the emulated contract (128x128 B=2 as tests/test_gpu_step16.py::test_step_gradients_16bit):
    with the masks frozen the remaining distance is rounding-flip noise of the 16-bit values themselves, held per network."""
    res = _frozen_and_free(T, dt, 2, 128, 32)
    worst = {}
    for which in ("dis", "gen"):
        for key, e in _per_net(res[which]["frozen"]).items():
            kk = "dis" if key.startswith("dis") else "gen" + key[key.index("."):]
            worst[kk] = max(worst.get(kk, 0.0), e)
    print("%s frozen-mask worst per network group:" % dt, {k: "%.2e" % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if v > ETOL_FROZEN[dt][k]}
    assert not bad, bad
