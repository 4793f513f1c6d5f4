"""Randomised shape sweep of the conv kernels (hypothesis) and the train.py loop.  -m gpu"""
import os
import subprocess
import sys

import pytest
import torch
import yaml
from hypothesis import given, settings, strategies as st, HealthCheck

from oracle import aclgan_oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _L():
    import aclgan_amd  # noqa: F401
    from aclgan_amd import _lib
    return _lib


conv_shapes = st.tuples(
    st.integers(1, 3),                              # B
    st.integers(4, 40), st.integers(4, 80),         # Hi, Wi (wide enough to cross the 70-position segments of the thin kernels)
    st.sampled_from([3, 4, 6, 8, 16, 32, 48, 64]),  # Ci
    st.sampled_from([1, 4, 8, 16, 32, 40, 64, 128]),  # Co
    st.sampled_from([(1, 1, 0), (3, 1, 1), (4, 2, 1), (5, 1, 2), (7, 1, 3)]),  # (k, s, p)
    st.booleans(),                                  # upsample
    st.integers(0, 10 ** 6))


@settings(max_examples=int(os.environ.get("ACLGAN_SWEEP_EXAMPLES", "40")), deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(conv_shapes)
def test_conv_fwd_dgrad_wgrad_random_shapes(shape):
    from gpu_util import conv_desc, gpu_conv_fwd, gpu_conv_dgrad, gpu_conv_wgrad, nhwc, nchw, ohwi, rel_err
    L = _L()
    B, Hi, Wi, Ci, Co, (k, s, p), up, seed = shape
    up = int(up)
    if p >= (Hi << up) or p >= (Wi << up) or (Hi << up) + 2 * p < k or (Wi << up) + 2 * p < k:
        return
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, Hi, Wi, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5).requires_grad_(True)
    b = (torch.randn(Co, generator=g) * 0.1).requires_grad_(True)
    y = O.conv_block(x, w, b, s, p, "none", upsample=bool(up))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    d = conv_desc(L, B, Hi, Wi, Ci, Co, k, s, p, up, "none")
    xg, wg, dyg = nhwc(x.detach()).cuda(), ohwi(w.detach()).cuda(), nhwc(dy).cuda()
    assert rel_err(nchw(gpu_conv_fwd(L, d, xg, wg, b.detach().cuda())), y) < 2e-4
    assert rel_err(nchw(gpu_conv_dgrad(L, d, dyg, wg)), x.grad) < 2e-4
    dw, db = gpu_conv_wgrad(L, d, xg, dyg)
    assert rel_err(dw, ohwi(w.grad)) < 2e-4
    assert rel_err(db, b.grad) < 2e-4


def test_train_loop_runs_and_resumes(tmp_path):
    """train.py counterpart (reference train.py:22-104): 3 iterations at a tiny config, snapshot, --resume."""
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female.yaml")))
    cfg["gen"].update(dim=8, mlp_dim=16, n_res=1)
    cfg["dis"].update(dim=8)
    cfg.update(batch_size=2, crop_image_height=64, crop_image_width=64, display_size=2, snapshot_save_iter=2, max_iter=3)
    path = os.path.join(tmp_path, "tiny.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    # a configured dataset that does not exist is an error, like the reference -- never a silent fallback to noise
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--config", path, "--output_path", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--synthetic" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--config", path, "--output_path", str(tmp_path), "--synthetic"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Iteration: 00000003/00000003" in r.stdout and "Finish training" in r.stdout
    ck = os.path.join(tmp_path, "outputs", "tiny", "checkpoints")
    assert {"gen_00000002.pt", "dis_00000002.pt", "gen_00000003.pt", "dis_00000003.pt", "optimizer.pt"} <= set(os.listdir(ck))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--config", path, "--output_path", str(tmp_path), "--resume",
                        "--max_iter", "4", "--synthetic"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Resume from iteration 3" in r.stdout and "Iteration: 00000004/00000004" in r.stdout
