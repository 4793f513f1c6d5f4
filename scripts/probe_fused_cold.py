"""the fused ResBlock convolution with COLD caches (as inside the step: GBs stream through between two uses of a filter): a 1 GiB fill runs between
calls; the convolution launch alone is timed with events (the filter transform is done once, outside).  usage: probe_fused_cold.py [modes...]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
B, Hi, Cc = 8, 64, 256
x = torch.randn(B, Hi, Hi, Cc, device="cuda"); w = torch.randn(Cc, 3, 3, Cc, device="cuda") * 0.02
b = torch.zeros(Cc, device="cuda"); y = torch.empty(B, Hi, Hi, Cc, device="cuda")
Uf = torch.empty(36 * Cc * Cc, device="cuda")
big = torch.empty(1 << 28, device="cuda")      # 1 GiB of floats
st = L.stream_ptr()
if len(sys.argv) > 1: L.lib.aclgan_set_tuning(b"wino_fused", int(sys.argv[1]))
L.check(L.lib.aclgan_winograd_filter_frag(L.ptr(w), L.ptr(Uf), Cc, Cc, 0, st))
def conv(): L.check(L.lib.aclgan_conv3x3_winograd_fused(L.ptr(x), L.ptr(Uf), L.ptr(b), L.ptr(y), B, Hi, Hi, Cc, Cc, 0, 1, 0, None, st))
for cold in (0, 0):
    for _ in range(5): conv()
    ts = []
    for _ in range(20):
        if cold:
            big.fill_(1.0); x.mul_(1.0)      # x itself is warm in the step (its producer just wrote it); U and everything else is not
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("fused conv alone, %s caches: median %.1f us, min %.1f, max %.1f" % ("COLD" if cold else "warm", ts[len(ts) // 2], ts[0], ts[-1]), flush=True)
