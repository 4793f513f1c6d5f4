"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol that
include/aclgan_hip.h declares; host-side logic that needs no GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _lib():
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol():
    L = _lib()
    hdr = open(os.path.join(ROOT, "include", "aclgan_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(aclgan_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert declared == set(L.SIGNATURES.keys()), declared ^ set(L.SIGNATURES.keys())
    for name in declared:
        assert getattr(L.lib, name) is not None
    assert L.lib.aclgan_version() >= 100


def test_context_and_parameter_layout_without_gpu():
    """ctx creation and the flat layout are host logic: reference parameters() order, OIHW shapes,
    16-byte aligned offsets, totals equal to the reference's parameter counts."""
    import json
    L = _lib()
    a = L.Arch(3, 6, 64, 256, 8, 4, 2, 4, 64, 4, 3)
    ctx = C.c_void_p()
    L.check(L.lib.aclgan_ctx_create(C.byref(a), C.byref(ctx)))
    order = json.load(open(os.path.join(ROOT, "tests", "golden", "param_order.json")))
    want_shapes = {}
    for line in open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.txt")):
        net, key, shp = line.split()
        want_shapes[net + "/" + key] = tuple(int(s) for s in shp.split("x"))
    for grp, key in ((0, "gen"), (1, "dis")):
        n = L.lib.aclgan_tensor_count(ctx, grp)
        assert n == len(order[key])
        name = C.create_string_buffer(256); off = C.c_int64(); shp = (C.c_int * 4)(); nd = C.c_int()
        total = 0
        for i in range(n):
            L.check(L.lib.aclgan_tensor_info(ctx, grp, i, name, 256, C.byref(off), shp, C.byref(nd)))
            full = name.value.decode()
            assert full == order[key][i]
            shape = tuple(shp[j] for j in range(nd.value))
            assert shape == want_shapes[full], full
            assert off.value % 4 == 0
            numel = 1
            for s in shape:
                numel *= s
            total += numel
        assert total == (30058648 if grp == 0 else 24822729)
        assert L.lib.aclgan_group_numel(ctx, grp) >= total
    # errors are codes + messages, never aborts
    assert L.lib.aclgan_tensor_info(ctx, 0, 10 ** 6, None, 0, None, None, None) == -1
    assert "out of range" in L.last_error()
    bad = L.Arch(3, 3, 64, 256, 8, 4, 2, 4, 64, 4, 3)
    ctx2 = C.c_void_p()
    assert L.lib.aclgan_ctx_create(C.byref(bad), C.byref(ctx2)) == -1
    L.lib.aclgan_ctx_destroy(ctx)


def test_conv_descriptor_validation_without_gpu():
    L = _lib()
    d = L.ConvDesc(1, 2, 2, 4, 4, 7, 1, 3, 0, 0)    # reflect pad 3 on a 2x2 map (torch raises as well)
    assert L.lib.aclgan_conv2d_dgrad_scratch_bytes(C.byref(d)) == 0
    d = L.ConvDesc(2, 8, 8, 16, 8, 5, 1, 2, 1, 0)   # 5x5 on the 2x upsampled 16x16 map, pad 2 -> 20x20 grid
    assert L.lib.aclgan_conv2d_dgrad_scratch_bytes(C.byref(d)) == 2 * 20 * 20 * 16 * 4
    assert L.lib.aclgan_norm_scratch_bytes(2, 64, 16) > 0


def test_weight_gradient_scratch_covers_both_winograd_paths_without_gpu():
    """aclgan_conv2d_wgrad_scratch_bytes is sized BEFORE the tuning switch may change: it must cover the one-kernel Winograd weight gradient
    (csrc/conv_wino_wgrad_fused.hip: ordered K-slice partials [ks][Cout][9][Cin] + bias partials) as well as the seven-launch pipeline, for
    every batch size; the switch round-trips its previous value."""
    L = _lib()
    old = L.lib.aclgan_set_tuning(b"wino_wgrad_fused", 0)
    assert old in (0, 1, 2)
    try:
        for B in (1, 3, 8):
            d = L.ConvDesc(B, 64, 64, 256, 256, 3, 1, 1, 0, 0)
            L.lib.aclgan_set_tuning(b"wino_wgrad_fused", 0)
            pipe = L.lib.aclgan_conv2d_wgrad_scratch_bytes(C.byref(d))
            assert L.lib.aclgan_set_tuning(b"wino_wgrad_fused", 2) == 0
            both = L.lib.aclgan_conv2d_wgrad_scratch_bytes(C.byref(d))
            assert both == pipe                              # (sized for either path whatever the switch says)
            groups = B * 16 * 4                              # strips of four 4x4 tiles
            ks = max(1, min(groups, 64, 256 // 32))          # 32 channel blocks of 64 x 32
            assert both >= ks * 256 * 9 * 256 * 4 + ks * 256 * 4
    finally:
        L.lib.aclgan_set_tuning(b"wino_wgrad_fused", old)


def test_product_path_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import aclgan_amd  # noqa: F401
    from aclgan_amd import trainer
    from oracle import aclgan_oracle as O
    with pytest.raises(trainer.L.AclganError):
        trainer.aclgan_Trainer(O.default_config())
    cfg = O.default_config(); cfg["dis"]["gan_type"] = "nsgan"
    with pytest.raises(trainer.L.AclganError):
        trainer.arch_from_config(cfg)


def test_step_executed_flops_dry_run_without_gpu():
    """aclgan_step_executed_flops (round 6; bench.py roofline.flop_per_launch): the matrix-pipe FLOPs an update executes, from a dry run of the
    scheduler with the launchers' own path decisions.  fp32: Winograd / sub-pixel / parity phases execute far fewer FLOPs than the SURVEY 8d
    contract (2.623 TFLOP per image at 256x256); the 16-bit paths have no Winograd; the count scales with the image area and (beyond the grid
    quantisation of the fused kernels) with the batch; forcing the direct kernels raises it."""
    L = _lib()
    a = L.Arch(3, 6, 64, 256, 8, 4, 2, 4, 64, 4, 3)
    fake = C.c_void_p(0x10000)

    def total(B, S, dtype=0):
        ctx = C.c_void_p()
        L.check(L.lib.aclgan_ctx_create(C.byref(a), C.byref(ctx)))
        L.check(L.lib.aclgan_set_compute_dtype(ctx, dtype))
        for grp in (0, 1):
            L.check(L.lib.aclgan_bind_params(ctx, grp, fake, fake, fake, fake))
            if dtype:
                L.check(L.lib.aclgan_bind_params16(ctx, grp, fake, fake))
        t = 0.0
        for which in (0, 1):
            v = C.c_double()
            L.check(L.lib.aclgan_step_executed_flops(ctx, which, B, S, S, C.byref(v)))
            t += v.value
        L.lib.aclgan_ctx_destroy(ctx)
        return t
    contract = 2.623e12 * 8
    f32 = total(8, 256)
    assert 0.25 * contract < f32 < 0.35 * contract, f32 / contract      # measured by SQ_INSTS_MFMA on the round-6 build: 6.34e12 (profiles/r06_step_traffic.json)
    assert 5.8e12 < f32 < 6.6e12, f32
    bf = total(8, 256, 1)
    assert 0.55 * contract < bf < 0.85 * contract, bf / contract          # sub-pixel phases only: no Winograd on the 16-bit pipes
    assert abs(total(4, 512) / (2 * f32) - 1.0) < 0.1                     # 4 x the area, half the batch (different tile-block quantisation)
    old = L.lib.aclgan_set_tuning(b"wino_fused", 0)
    try:
        assert total(8, 256) > f32 * 1.03                                  # the stride-2 layers back on the direct kernels: 16 / 9 of their FLOPs
    finally:
        L.lib.aclgan_set_tuning(b"wino_fused", old)
    assert L.lib.aclgan_launch_count() == 0


def test_step_algorithmic_bytes_dry_run_without_gpu():
    """aclgan_step_algorithmic_bytes (bench.py roofline.algorithmic_bytes) is a launch-free dry run of the step scheduler: host logic.
    Conv / norm / loss traffic scales with B*H*W exactly; the parameter traffic (weights read per pass, zero_grad, Adam) does not."""
    import torch
    L = _lib()
    a = L.Arch(3, 6, 64, 256, 8, 4, 2, 4, 64, 4, 3)
    ctx = C.c_void_p()
    L.check(L.lib.aclgan_ctx_create(C.byref(a), C.byref(ctx)))
    t = torch.zeros(8)
    for grp in (0, 1):
        L.check(L.lib.aclgan_bind_params(ctx, grp, L.ptr(t), L.ptr(t), None, None))

    def q(which, B, S):
        v = C.c_double()
        L.check(L.lib.aclgan_step_algorithmic_bytes(ctx, which, B, S, S, C.byref(v)))
        return v.value
    g8, d8, g16, g8_512 = q(0, 8, 256), q(1, 8, 256), q(0, 16, 256), q(0, 8, 512)
    assert 40e9 < g8 < 90e9 and 10e9 < d8 < 30e9            # ~60 GB + ~19 GB per step at 256x256 B=8 (fp32 storage): ~10 ms at 8 TB/s
    fixed = 2 * g8 - g16                                     # what does not scale with the batch: parameter traffic
    assert 0 < fixed < 0.1 * g8
    assert abs((g8_512 - fixed) - 4 * (g8 - fixed)) <= 1e-4 * g8_512      # (the MLP and the style vectors scale with B only)
    assert L.lib.aclgan_step_algorithmic_bytes(ctx, 2, 8, 256, 256, C.byref(C.c_double())) != 0
    assert L.lib.aclgan_launch_count() == 0                  # a dry run launches nothing
    L.lib.aclgan_ctx_destroy(ctx)


def test_host_side_under_address_sanitizer(tmp_path):
    """SURVEY.md section 5 (sanitizers): the HOST half of the library -- C-ABI argument checks, flat parameter layout, the step scheduler's
    launch-free dry runs (workspace sizing, algorithmic bytes, the bucket schedule of both updates, focus and non-focus architectures),
    error paths -- instrumented by AddressSanitizer (`make -C acl-gan_amd/csrc asan`: host-only objects) and driven in a subprocess under
    the ASan runtime.  Any heap / stack / use-after-free error aborts the child; leaks are not checked (ctypes / Python own the process)."""
    import glob
    import shutil
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "acl-gan_amd", "csrc")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not (os.path.exists(hipcc) or shutil.which(hipcc)) or not rt:
        pytest.skip("no hipcc / ASan runtime on this machine: the sanitizer build is a toolchain matter, not a test failure")
    # built into the test's temporary directory: no artefacts in the source tree
    asan_dir = str(tmp_path / "asan")
    r = subprocess.run(["make", "-C", csrc, "-j8", "asan", "ASAN_DIR=" + asan_dir], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(asan_dir, "libaclgan_hip_asan.so")
    prog = r'''
import ctypes as C, sys
L = C.CDLL(%r)
class Arch(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("input_dim_a", "input_dim_b", "gen_dim", "gen_mlp_dim", "gen_style_dim", "gen_output_dim", "gen_n_downsample", "gen_n_res",
                                        "dis_dim", "dis_n_layer", "dis_num_scales")]
L.aclgan_last_error.restype = C.c_char_p
L.aclgan_group_numel.restype = C.c_int64
assert L.aclgan_version() >= 100
for out_dim, dims in ((4, (64, 256, 64)), (3, (8, 16, 8)), (4, (16, 32, 16))):
    a = Arch(3, 6, dims[0], dims[1], 8, out_dim, 2, 4, dims[2], 4, 3)
    ctx = C.c_void_p()
    assert L.aclgan_ctx_create(C.byref(a), C.byref(ctx)) == 0, L.aclgan_last_error()
    name = C.create_string_buffer(256); off = C.c_int64(); shp = (C.c_int * 4)(); nd = C.c_int()
    for grp in (0, 1):
        for i in range(L.aclgan_tensor_count(ctx, grp)):
            assert L.aclgan_tensor_info(ctx, grp, i, name, 256, C.byref(off), shp, C.byref(nd)) == 0
        assert L.aclgan_tensor_info(ctx, grp, 10 ** 6, None, 0, None, None, None) != 0
        assert L.aclgan_tensor_info(ctx, grp, 0, name, 4, C.byref(off), shp, C.byref(nd)) in (0, -1)      # short name buffer
        fake = C.c_void_p(0x10000)      # device pointers are never dereferenced on the host
        assert L.aclgan_bind_params(ctx, grp, fake, fake, fake, fake) == 0
    for (B, H, W) in ((1, 64, 64), (2, 128, 64), (3, 72, 100)):
        ws = C.c_size_t()
        rc = L.aclgan_workspace_bytes(ctx, B, H, W, C.byref(ws))
        assert (rc == 0 and ws.value > 0) or (H %% 4 or W %% 4), (rc, L.aclgan_last_error())
        fw = C.c_size_t()
        L.aclgan_forward_workspace_bytes(ctx, B, H, W, C.byref(fw))
        v = C.c_double()
        for which in (0, 1, 2):
            L.aclgan_step_algorithmic_bytes(ctx, which, B, H, W, C.byref(v))
        order = (C.c_int * 8192)(); cnt = C.c_int()
        for grp in (0, 1):
            for bucket in (4096, 1 << 20):
                if L.aclgan_set_grad_buckets(ctx, C.c_int64(bucket), None, None) == 0:
                    L.aclgan_bucket_schedule(ctx, grp, B, H, W, 0, order, 8192, C.byref(cnt))
                    L.aclgan_bucket_schedule(ctx, grp, B, H, W, 0, order, 1, C.byref(cnt))      # capacity too small: an error code, no overrun
    assert L.aclgan_workspace_bytes(ctx, 0, 64, 64, C.byref(C.c_size_t())) != 0
    assert L.aclgan_workspace_bytes(None, 1, 64, 64, C.byref(C.c_size_t())) != 0
    # round 5: the scheduler with 1 .. 4 lanes (branches of the update on separate streams; dry runs follow the same allocation rules),
    # and the workspace check: exactly aclgan_workspace_bytes passes, one byte less is ACLGAN_ENOMEM (-4) -- before anything is enqueued
    prev = C.c_int()
    for lanes in (1, 2, 3, 4):
        assert L.aclgan_tuning(b"lanes", lanes, C.byref(prev)) == 0
        ws = C.c_size_t()
        assert L.aclgan_workspace_bytes(ctx, 2, 64, 64, C.byref(ws)) == 0 and ws.value > 0
        fake = C.c_void_p(0x100000)
        assert L.aclgan_bind_workspace(ctx, fake, C.c_size_t(ws.value)) == 0
        assert L.aclgan_check_workspace(ctx, 2, 64, 64) == 0, L.aclgan_last_error()
        assert L.aclgan_bind_workspace(ctx, fake, C.c_size_t(ws.value - 1)) == 0
        assert L.aclgan_check_workspace(ctx, 2, 64, 64) == -4 and b"too small" in L.aclgan_last_error()
        for grp in (0, 1):
            if L.aclgan_set_grad_buckets(ctx, C.c_int64(1 << 16), None, None) == 0:
                assert L.aclgan_bucket_schedule(ctx, grp, 2, 64, 64, 0, order, 8192, C.byref(cnt)) == 0
    assert L.aclgan_tuning(b"lanes", 2, None) == 0
    L.aclgan_ctx_destroy(ctx)
bad = Arch(3, 3, 64, 256, 8, 4, 2, 4, 64, 4, 3); ctx = C.c_void_p()
assert L.aclgan_ctx_create(C.byref(bad), C.byref(ctx)) != 0 and L.aclgan_last_error()
assert L.aclgan_set_tuning(b"no such knob", 1) == -1
prev = C.c_int(12345)
assert L.aclgan_tuning(b"no such knob", 1, C.byref(prev)) == -1 and prev.value == 12345 and b"unknown key" in L.aclgan_last_error()
assert L.aclgan_tuning(b"u_batch", 0, C.byref(prev)) == 0 and prev.value in (0, 1)
assert L.aclgan_tuning(b"u_batch", prev.value, None) == 0
assert L.aclgan_tuning(None, 1, None) == -1
assert L.aclgan_launch_count() == 0
print("ASAN_CHILD_OK")
''' % lib
    env = dict(os.environ)
    env.update(LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0")
    p = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ASAN_CHILD_OK" in p.stdout and "AddressSanitizer" not in p.stderr, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
