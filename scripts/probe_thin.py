"""the image-side layers in isolation (for rocprofv3 PMC passes and HIP-event timing): CE0 / SE0 3->64 7x7, DO 64->4 7x7, and the first
discriminator layers 3->64 / 6->64 4x4 stride 2 -- forward, input gradient, weight gradient, at the shapes of the 256x256 B=8 step.
    python scripts/probe_thin.py [reps=5]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = [("CE0 3>64 7x7 @256 B8", 8, 256, 3, 64, 7, 1, 3), ("DO 64>4 7x7 @256 B8", 8, 256, 64, 4, 7, 1, 3),
          ("D0 3>64 4x4s2 @256 B16", 16, 256, 3, 64, 4, 2, 1), ("D0 6>64 4x4s2 @256 B16", 16, 256, 6, 64, 4, 2, 1), ("D0 3>64 4x4s2 @128 B16", 16, 128, 3, 64, 4, 2, 1)]
st = L.stream_ptr()
def timeit(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-26s %10s %10s %10s   us (GFLOP of the direct convolution: fwd = dgrad = wgrad)" % ("layer", "fwd", "dgrad", "wgrad"))
for name, B, Hi, Ci, Co, k, s, p in SHAPES:
    Ho = (Hi + 2 * p - k) // s + 1
    x = torch.randn(B, Hi, Hi, Ci, device="cuda"); w = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
    b = torch.zeros(Co, device="cuda"); y = torch.empty(B, Ho, Ho, Co, device="cuda"); dy = torch.randn_like(y)
    dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros(Co, device="cuda")
    d = L.ConvDesc(B, Hi, Hi, Ci, Co, k, s, p, 0, 0)
    scr = torch.empty(L.lib.aclgan_conv2d_dgrad_scratch_bytes(C.byref(d)) // 4 + 16, device="cuda")
    fscr = torch.empty(max(L.lib.aclgan_conv2d_fwd_scratch_bytes(C.byref(d)), L.lib.aclgan_conv2d_wgrad_scratch_bytes(C.byref(d))) // 4 + 16, device="cuda")
    f = lambda: L.check(L.lib.aclgan_conv2d_fwd_ws(C.byref(d), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(fscr), st))
    g = lambda: L.check(L.lib.aclgan_conv2d_dgrad(C.byref(d), L.ptr(dy), L.ptr(w), L.ptr(dx), L.ptr(scr), 0, st))
    h = lambda: L.check(L.lib.aclgan_conv2d_wgrad_ws(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(fscr), st))
    print("%-26s %10.1f %10.1f %10.1f   (%.1f)" % (name, timeit(f), timeit(g), timeit(h), 2.0 * B * Ho * Ho * Co * k * k * Ci / 1e9))
