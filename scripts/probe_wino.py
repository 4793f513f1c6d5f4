"""run the ResBlock conv through the scratch (Winograd) path a few times: fwd | dgrad | wgrad  (for rocprofv3)"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, Hi, Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 64, 256
x = torch.randn(B, Hi, Hi, Cc, device="cuda"); w = torch.randn(Cc, 3, 3, Cc, device="cuda") * 0.02
b = torch.zeros(Cc, device="cuda"); y = torch.empty(B, Hi, Hi, Cc, device="cuda"); dy = torch.randn_like(y)
dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros(Cc, device="cuda")
d = L.ConvDesc(B, Hi, Hi, Cc, Cc, 3, 1, 1, 0, 0)
nb = max(L.lib.aclgan_conv2d_fwd_scratch_bytes(C.byref(d)), L.lib.aclgan_conv2d_dgrad_scratch_bytes(C.byref(d)), L.lib.aclgan_conv2d_wgrad_scratch_bytes(C.byref(d)))
scr = torch.empty(nb // 4 + 64, device="cuda")
st = L.stream_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(8):
    if it == 3: e0.record()
    if which == "fwd": L.check(L.lib.aclgan_conv2d_fwd_ws(C.byref(d), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(scr), st))
    elif which == "dgrad": L.check(L.lib.aclgan_conv2d_dgrad(C.byref(d), L.ptr(dy), L.ptr(w), L.ptr(dx), L.ptr(scr), 0, st))
    else: L.check(L.lib.aclgan_conv2d_wgrad_ws(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(scr), st))
e1.record(); torch.cuda.synchronize()
flop = 2.0 * B * Hi * Hi * Cc * 9 * Cc
ms = e0.elapsed_time(e1) / 5
print("%s B=%d %dx%d: %.1f us  %.0f algorithmic TFLOP/s" % (which, B, Hi, Hi, ms * 1e3, flop / ms / 1e9))
