"""time the fp32 GEMM slices against the split-bf16 (x3) launch on the Winograd shapes of the step (HIP events, launch alone)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
st = L.stream_ptr()


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (T, K, N, ns, tag) in [(2048, 256, 256, 36, "ResBlock 8x64x64 256->256"), (8192, 128, 128, 36, "64x64 tiles 128->128"), (7688, 256, 128, 144, "up5 phases 256->128 (62x62 valid)"),
                           (2048, 256, 256, 144, "144 slices 256->256"), (512, 256, 256, 36, "B=2 ResBlock")]:
    A = torch.randn(ns, T, K, device="cuda"); B = torch.randn(ns, N, K, device="cuda"); C = torch.empty(ns, T, N, device="cuda")
    scr = torch.empty(L.lib.aclgan_gemm_slices_x3_scratch_bytes(T, K, N, ns) // 4 + 64, device="cuda")
    L.check(L.lib.aclgan_gemm_slices_x3(L.ptr(A), L.ptr(B), L.ptr(C), T, K, N, ns, L.ptr(scr), st))
    ref = torch.bmm(A[:2].double(), B[:2].double().transpose(1, 2))
    err = ((C[:2].double() - ref).abs().max() / ref.abs().max()).item()
    flop = 2.0 * ns * T * K * N
    u1 = timed(lambda: L.check(L.lib.aclgan_gemm_slices_f32(L.ptr(A), L.ptr(B), L.ptr(C), T, K, N, ns, st)))
    u3 = timed(lambda: L.check(L.lib.aclgan_gemm_slices_x3(None, None, L.ptr(C), T, K, N, ns, L.ptr(scr), st)))
    us = timed(lambda: L.check(L.lib.aclgan_gemm_slices_x3(L.ptr(A), L.ptr(B), L.ptr(C), T, K, N, ns, L.ptr(scr), st)))
    print("%-36s %.1f GFLOP  f32 %.1f us %.0f TF | x3 %.1f us %.0f TF (with the split passes %.1f us)  max rel err %.2g" % (tag, flop / 1e9, u1, flop / u1 / 1e6, u3, flop / u3 / 1e6, us, err), flush=True)
