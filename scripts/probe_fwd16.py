"""The ResBlock convolution of the 16-bit path (8 x 64 x 64 x 256 -> 256, 3 x 3, 16-bit activations in HBM) a few times, for rocprofv3:
    python scripts/probe_fwd16.py [bf16|fp16] [patch mode: 0 conv_fwd16s | 1 patch kernel, lockstep (default) | 2 patch kernel, counter-phase] [B]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
Hi, Cc = 64, 256
L.check(L.lib.aclgan_tuning(b"fwd16_patch", mode, None))
tdt = torch.bfloat16 if dt == "bf16" else torch.float16
code = L.DTYPE[dt]
x16 = torch.randn(B, Hi, Hi, Cc, device="cuda").to(tdt)
w = torch.randn(Cc, 3, 3, Cc, device="cuda") * 0.02
b = torch.zeros(Cc, device="cuda")
y16 = torch.empty(B, Hi, Hi, Cc, device="cuda", dtype=tdt)
w16 = torch.empty(w.numel(), dtype=torch.int16, device="cuda")
st = L.stream_ptr()
L.check(L.lib.aclgan_pack_weights16(L.ptr(w), L.ptr(w16), None, Cc, 9, Cc, code, st))
d = L.ConvDesc(B, Hi, Hi, Cc, Cc, 3, 1, 1, 0, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 12
for it in range(3 + N):
    if it == 3: e0.record()
    L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(b), L.ptr(y16), code, st))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
flop = 2.0 * B * Hi * Hi * Cc * 9 * Cc
print("fwd16 %s patch mode %d B=%d: %.1f us  %.0f TFLOP/s  (%.3f of 2500)" % (dt, mode, B, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 2500))
