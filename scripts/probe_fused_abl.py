"""time the fused Winograd forward on the ResBlock shape under each tuning mode given on the command line (ablation builds: mode | abl << 4)"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
B, Hi, Cc = int(os.environ.get("B", "8")), int(os.environ.get("HI", "64")), 256
x = torch.randn(B, Hi, Hi, Cc, device="cuda"); w = torch.randn(Cc, 3, 3, Cc, device="cuda") * 0.02
b = torch.zeros(Cc, device="cuda"); y = torch.empty(B, Hi, Hi, Cc, device="cuda")
d = L.ConvDesc(B, Hi, Hi, Cc, Cc, 3, 1, 1, 0, 0)
scr = torch.empty(L.lib.aclgan_conv2d_fwd_scratch_bytes(C.byref(d)) // 4 + 64, device="cuda")
st = L.stream_ptr()
def run(): L.check(L.lib.aclgan_conv2d_fwd_ws(C.byref(d), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(scr), st))
for mode in [int(a) for a in sys.argv[1:]] or [0, 1, 2]:
    L.lib.aclgan_set_tuning(b"wino_fused", mode)
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): run()
    e1.record(); torch.cuda.synchronize()
    print("mode %3d (kc %d abl %2d): %.1f us per forward (filter transform launch included)" % (mode, 8 * (mode & 15), mode >> 4, e0.elapsed_time(e1) / 60 * 1e3), flush=True)
