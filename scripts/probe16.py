"""run ONE 16-bit conv kernel shape a few times (for rocprofv3 --pmc / --kernel-trace).
usage: probe16.py fwd|dgrad|wgrad [bf16|fp16] [Hi Ci Co k s p up]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dt = L.DTYPE[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
Hi, Ci, Co, k, s, p, up = [int(v) for v in (sys.argv[3:10] if len(sys.argv) > 9 else "64 256 256 3 1 1 0".split())]
B = 8
Hu = Hi << up; Ho = (Hu + 2 * p - k) // s + 1
x = torch.randn(B, Hi, Hi, Ci, device="cuda"); w = torch.randn(Co, k, k, Ci, device="cuda") * 0.02
b = torch.zeros(Co, device="cuda"); y = torch.empty(B, Ho, Ho, Co, device="cuda"); dy = torch.randn_like(y)
dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros(Co, device="cuda")
w16 = torch.empty(w.numel(), dtype=torch.int16, device="cuda"); w16t = torch.empty_like(w16)
d = L.ConvDesc(B, Hi, Hi, Ci, Co, k, s, p, up, 0)
st = L.stream_ptr()
L.check(L.lib.aclgan_pack_weights16(L.ptr(w), L.ptr(w16), L.ptr(w16t), Co, k * k, Ci, dt, st))
nb = max(L.lib.aclgan_conv2d_fwd16_scratch_bytes(C.byref(d)), L.lib.aclgan_conv2d_dgrad16_scratch_bytes(C.byref(d)),
         L.lib.aclgan_conv2d_wgrad16_scratch_bytes(C.byref(d)))
scr = torch.empty(nb // 4 + 64, device="cuda")
x16 = torch.empty(x.numel(), dtype=torch.int16, device="cuda")
L.check(L.lib.aclgan_pack_weights16(L.ptr(x), L.ptr(x16), None, B * Hi * Hi, 1, Ci, dt, st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(8):
    if it == 3: e0.record()
    if which == "fwdx16": L.check(L.lib.aclgan_conv2d_fwd16_x16(C.byref(d), dt, L.ptr(x16), L.ptr(w), L.ptr(w16), L.ptr(b), L.ptr(y), L.ptr(scr), st))
    elif which == "fwd": L.check(L.lib.aclgan_conv2d_fwd16(C.byref(d), dt, L.ptr(x), L.ptr(w), L.ptr(w16), L.ptr(b), L.ptr(y), L.ptr(scr), st))
    elif which == "dgrad": L.check(L.lib.aclgan_conv2d_dgrad16(C.byref(d), dt, L.ptr(dy), L.ptr(w), L.ptr(w16t), L.ptr(dx), 0, L.ptr(scr), st))
    else: L.check(L.lib.aclgan_conv2d_wgrad16(C.byref(d), dt, L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(scr), st))
e1.record(); torch.cuda.synchronize()
flop = 2.0 * B * Ho * Ho * Co * k * k * Ci
ms = e0.elapsed_time(e1) / 5
print("%s %s: %.1f us  %.0f TFLOP/s" % (which, sys.argv[3:10], ms * 1e3, flop / ms / 1e9))
