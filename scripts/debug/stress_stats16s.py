"""Stress the run-to-run reproducibility of the conv_fwd16s statistics epilogue (tests/test_gpu_ops16s.py::test_conv_fwd16s_epilogue_statistics
failed ONCE in ~10 full-suite runs on the 256 x 256 eight-wave tile): repeat the launch, report where and by how much the (mean, M2) pairs or y
differ.  Run on the GPU box:  python scripts/debug/stress_stats16s.py [iterations] [tile]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aclgan_amd  # noqa
from aclgan_amd import _lib as L
from gpu_util import conv_desc, out_hw, nhwc, ohwi
import test_gpu_ops16s as T

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L.lib.aclgan_set_tuning(b"glds_tile", tile)
bad_total = 0
for case, dt in [(T.CASES[7], "fp16"), (T.CASES[6], "bf16"), (T.CASES[0], "bf16")]:
    B, Hi, Wi, Ci, Co, k, s, p, act = case
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, 0, act)
    R = L.lib.aclgan_conv2d_fwd16s_stats_chunk(C.byref(d))
    x, w, b = T._t(case, 2)
    wg, bg = ohwi(w).cuda(), b.cuda()
    x16 = nhwc(x).cuda().to(T.TDT[dt]); w16, _ = T._packs(L, wg, dt)
    Ho, Wo = out_hw(Hi, Wi, k, s, p, 0); code = L.DTYPE[dt]; M = B * Ho * Wo
    y0 = torch.empty(B, Ho, Wo, Co, device="cuda"); s0 = torch.empty(M // R, Co, 2, device="cuda")
    L.check(L.lib.aclgan_conv2d_fwd16s_stats(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y0), 0, L.ptr(s0), L.stream_ptr()))
    bad = 0
    for it in range(iters):
        y1 = torch.empty_like(y0); s1 = torch.empty_like(s0)
        L.check(L.lib.aclgan_conv2d_fwd16s_stats(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(bg), L.ptr(y1), 0, L.ptr(s1), L.stream_ptr()))
        es, ey = torch.equal(s0, s1), torch.equal(y0, y1)
        if not (es and ey):
            bad += 1
            ds = (s0 != s1).nonzero(); dy = (y0 != y1).nonzero()
            print("iter %d case %s: stats differ at %d entries, y at %d; first stats idx %s: %s vs %s; first y idx %s" %
                  (it, case, ds.shape[0], dy.shape[0], ds[0].tolist() if len(ds) else None,
                   s0[tuple(ds[0])].item() if len(ds) else None, s1[tuple(ds[0])].item() if len(ds) else None, dy[0].tolist() if len(dy) else None), flush=True)
            if bad > 5: break
    print("case %s %s R=%d: %d / %d launches differ from the first" % (case, dt, R, bad, iters), flush=True)
    bad_total += bad
sys.exit(1 if bad_total else 0)
