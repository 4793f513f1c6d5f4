"""time the 16-bit-storage conv kernels on a few step shapes (HIP events):  python scripts/probe16s.py [bf16|fp16]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
code = L.DTYPE[dt]; T = torch.bfloat16 if dt == "bf16" else torch.float16
st = L.stream_ptr()


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (B, H, Ci, Co, k, s, p, tag) in [(8, 64, 256, 256, 3, 1, 1, "ResBlock"), (8, 128, 128, 256, 4, 2, 1, "CE2"), (24, 128, 64, 128, 4, 2, 1, "D2 (3B joint)"),
                                     (32, 64, 256, 256, 3, 1, 1, "ResBlock B=32"), (24, 64, 128, 256, 4, 2, 1, "D3 scale0"), (24, 32, 256, 512, 4, 2, 1, "D4 scale0"),
                                     (24, 32, 128, 256, 4, 2, 1, "D3 scale1"), (24, 16, 256, 512, 4, 2, 1, "D4 scale1"), (8, 32, 256, 256, 3, 1, 1, "ResBlock 128px input")]:
    if os.environ.get("PROBE_ONLY") and os.environ["PROBE_ONLY"] != tag: continue
    Ho = (H + 2 * p - k) // s + 1
    d = L.ConvDesc(B, H, H, Ci, Co, k, s, p, 0, 0)
    x = torch.randn(B, H, H, Ci, device="cuda").to(T); w = (torch.randn(Co, k, k, Ci, device="cuda") * 0.02)
    b = torch.zeros(Co, device="cuda"); y = torch.empty(B, Ho, Ho, Co, device="cuda", dtype=T); dy = torch.randn(B, Ho, Ho, Co, device="cuda").to(T)
    dx = torch.empty(B, H, H, Ci, device="cuda", dtype=T); dw = torch.zeros(Co, k, k, Ci, device="cuda"); db = torch.zeros(Co, device="cuda")
    w16 = torch.empty(w.numel(), dtype=torch.int16, device="cuda"); w16t = torch.empty_like(w16)
    L.check(L.lib.aclgan_pack_weights16(L.ptr(w), L.ptr(w16), L.ptr(w16t), Co, k * k, Ci, code, st))
    flop = 2.0 * B * Ho * Ho * Co * k * k * Ci
    out = [tag, "%.1f GFLOP" % (flop / 1e9)]
    if L.lib.aclgan_conv16s_ok(C.byref(d), 0):
        us = timed(lambda: L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x), L.ptr(w16), L.ptr(b), L.ptr(y), code, st)))
        out.append("fwd16s %.1f us %.0f TF" % (us, flop / us / 1e6))
    scr = torch.empty(L.lib.aclgan_conv2d_dgrad16s_scratch_bytes(C.byref(d)) // 4 + 64, device="cuda")
    us = timed(lambda: L.check(L.lib.aclgan_conv2d_dgrad16s(C.byref(d), code, L.ptr(dy), L.ptr(w16t), L.ptr(dx), code, 0, L.ptr(scr), st)))
    out.append("dgrad16s+fold %.1f us %.0f TF" % (us, flop / us / 1e6))
    scr2 = torch.empty(L.lib.aclgan_conv2d_wgrad16_scratch_bytes(C.byref(d)) // 4 + 64, device="cuda")
    us = timed(lambda: L.check(L.lib.aclgan_conv2d_wgrad16_st(C.byref(d), code, L.ptr(x), code, L.ptr(dy), code, L.ptr(dw), L.ptr(db), L.ptr(scr2), st)))
    out.append("wgrad16(x16,dy16) %.1f us %.0f TF" % (us, flop / us / 1e6))
    print("  ".join(out), flush=True)
