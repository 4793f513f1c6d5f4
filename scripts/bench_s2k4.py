"""4x4 stride-2 layers: the direct implicit-GEMM kernels (tuning wino_fused = 0) against the four-parity-phase fused Winograd kernel
(mode 2 = forced), forward and input gradient, at the shapes of the 256x256 step (HIP events; the fused numbers INCLUDE the filter
transform launch, which the step pays once per update -- the kernel trace of the same run separates them)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L

SHAPES = [  # name, B, Hi, Ci, Co
    ("CE1 64>128@256 B8", 8, 256, 64, 128),
    ("CE2 128>256@128 B8", 8, 128, 128, 256),
    ("SE2 128>256@128 B8", 8, 128, 128, 256),
    ("SE3 256>256@64 B8", 8, 64, 256, 256),
    ("SE4 256>256@32 B8", 8, 32, 256, 256),
    ("D1 64>128@128 B8", 8, 128, 64, 128),
    ("D1 64>128@128 B16", 16, 128, 64, 128),
    ("D2 128>256@64 B8", 8, 64, 128, 256),
    ("D2 128>256@64 B16", 16, 64, 128, 256),
    ("D3 256>512@32 B8", 8, 32, 256, 512),
    ("D3 256>512@32 B16", 16, 32, 256, 512),
    ("D1s2 64>128@64 B16", 16, 64, 64, 128),
    ("D2s2 128>256@32 B16", 16, 32, 128, 256),
    ("CE1 64>128@512 B4", 4, 512, 64, 128),
    ("CE2 128>256@256 B4", 4, 256, 128, 256),
    ("CE2 128>256@128 B3", 3, 128, 128, 256),
]
st = L.stream_ptr()
def timeit(fn, reps=int(os.environ.get("REPS", "10"))):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("%-24s %22s %22s   us: direct | fused (forced) | model says" % ("shape", "forward", "input gradient"))
old = L.lib.aclgan_set_tuning(b"wino_fused", 1)
for name, B, Hi, Ci, Co in SHAPES:
    Ho = Hi // 2
    x = torch.randn(B, Hi, Hi, Ci, device="cuda"); w = torch.randn(Co, 4, 4, Ci, device="cuda") * 0.02
    b = torch.zeros(Co, device="cuda"); y = torch.empty(B, Ho, Ho, Co, device="cuda"); dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    d = L.ConvDesc(B, Hi, Hi, Ci, Co, 4, 2, 1, 0, 0)
    res = {}
    for mode in (0, 2, 1):
        L.lib.aclgan_set_tuning(b"wino_fused", mode)
        scr = torch.empty(L.lib.aclgan_conv2d_dgrad_scratch_bytes(C.byref(d)) // 4 + 16, device="cuda")
        fscr = torch.empty(L.lib.aclgan_conv2d_fwd_scratch_bytes(C.byref(d)) // 4 + 16, device="cuda")
        f = lambda: L.check(L.lib.aclgan_conv2d_fwd_ws(C.byref(d), L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(fscr), st))
        g = lambda: L.check(L.lib.aclgan_conv2d_dgrad(C.byref(d), L.ptr(dy), L.ptr(w), L.ptr(dx), L.ptr(scr), 0, st))
        res[mode] = (timeit(f) * 1e3, timeit(g) * 1e3)
    print("%-24s %6.0f %6.0f %6.0f    %6.0f %6.0f %6.0f" % (name, res[0][0], res[2][0], res[1][0], res[0][1], res[2][1], res[1][1]))
L.lib.aclgan_set_tuning(b"wino_fused", old)
