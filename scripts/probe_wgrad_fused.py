"""Fused Winograd weight gradient (csrc/conv_wino_wgrad_fused.hip) against the pipeline of conv_wino.hip and an fp64 reference: dw, db,
accumulation into non-zero gradients, ragged K slices; then HIP-event timings of both on the ResBlock shape at several batch sizes.
Run on the GPU box."""
import ctypes as C, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
from gpu_util import conv_desc, gpu_conv_wgrad, nhwc, ohwi

st = L.stream_ptr()


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


CASES = [(1, 16, 16, 64, 64), (2, 16, 16, 32, 64), (1, 8, 32, 64, 128), (3, 20, 48, 96, 64), (2, 64, 64, 256, 256), (5, 32, 32, 128, 128), (8, 64, 64, 256, 256),
         (1, 128, 128, 128, 128)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    CASES = CASES[:4]
bad = 0
for (B, H, W, Ci, Co) in CASES:
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Ci, H, W, generator=g).cuda(); dy = torch.randn(B, Co, H, W, generator=g).cuda()
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.double, device="cuda", requires_grad=True)
    bb = torch.zeros(Co, dtype=torch.double, device="cuda", requires_grad=True)
    y = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w, bb)
    y.backward(dy.double())
    d = conv_desc(L, B, H, W, Ci, Co, 3, 1, 1, 0, "none")
    res = {}
    for mode in (0, 2):
        L.lib.aclgan_set_tuning(b"wino_wgrad_fused", mode)
        dw, db = gpu_conv_wgrad(L, d, nhwc(x), nhwc(dy))
        torch.cuda.synchronize()
        res[mode] = (rel(dw, ohwi(w.grad)), rel(db, bb.grad))
    ok = max(res[2]) < 2e-4
    bad += 0 if ok else 1
    print("%-24s %s  rel L2 dw/db  pipeline %.2e %.2e | fused %.2e %.2e" % (str((B, H, W, Ci, Co)), "ok " if ok else "BAD", *res[0], *res[2]), flush=True)
L.lib.aclgan_set_tuning(b"wino_wgrad_fused", 1)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(1 if bad else 0)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (B, H, Ci, Co) in [(1, 64, 256, 256), (2, 64, 256, 256), (3, 64, 256, 256), (4, 64, 256, 256), (6, 64, 256, 256), (8, 64, 256, 256), (4, 128, 256, 256), (8, 64, 128, 128)]:
    x = torch.randn(B, H, H, Ci, device="cuda"); dy = torch.randn(B, H, H, Co, device="cuda")
    dw = torch.zeros(Co, 3, 3, Ci, device="cuda"); db = torch.zeros(Co, device="cuda")
    d = L.ConvDesc(B, H, H, Ci, Co, 3, 1, 1, 0, 0)
    scr = torch.empty(L.lib.aclgan_conv2d_wgrad_scratch_bytes(C.byref(d)) // 4 + 16, device="cuda")
    out = []
    for mode in (0, 2):
        L.lib.aclgan_set_tuning(b"wino_wgrad_fused", mode)
        out.append(timeit(lambda: L.check(L.lib.aclgan_conv2d_wgrad_ws(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(scr), st))))
    print("B=%d %dx%d %d->%d: pipeline %.1f us, fused %.1f us back to back" % (B, H, H, Ci, Co, out[0], out[1]), flush=True)
L.lib.aclgan_set_tuning(b"wino_wgrad_fused", 1)
print("BAD cases:", bad)
sys.exit(1 if bad else 0)
