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
