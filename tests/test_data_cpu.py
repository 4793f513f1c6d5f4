"""Input pipeline -- CPU side: the oracle against Pillow and the golden vectors, the library's
HOST coefficient tables against Pillow (integer arithmetic: bit-exact), dataset discovery and the order of the
random draws.  No GPU needed: aclgan_image_resample_* are host functions of libaclgan_hip.so."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st
from PIL import Image

from conftest import GOLDEN, ROOT  # noqa: F401
from oracle import data_oracle as D

import aclgan_amd  # noqa: F401
from aclgan_amd import _lib as L
from aclgan_amd import data as PD


def lib_table(n_in, n_out):
    ks = L.lib.aclgan_image_resample_ksize(n_in, n_out)
    b = np.zeros((n_out, 2), np.int32); k = np.zeros((n_out, ks), np.int32)
    ip = C.POINTER(C.c_int)
    L.check(L.lib.aclgan_image_resample_coeffs(n_in, n_out, b.ctypes.data_as(ip), k.ctypes.data_as(ip)))
    return b, k


def two_pass(img, ow, oh):
    """the kernel's arithmetic (csrc/image.hip) in numpy, driven by the LIBRARY's tables"""
    h, w = img.shape[:2]
    t = D._pass(img, *[a.astype(np.int64) for a in lib_table(w, ow)])
    return np.ascontiguousarray(D._pass(t.transpose(1, 0, 2), *[a.astype(np.int64) for a in lib_table(h, oh)]).transpose(1, 0, 2))


@settings(max_examples=40, deadline=None)
@given(h=st.integers(1, 200), w=st.integers(1, 200), oh=st.integers(1, 200), ow=st.integers(1, 200), seed=st.integers(0, 2**31 - 1))
def test_library_tables_and_restatement_equal_pillow(h, w, oh, ow, seed):
    img = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(two_pass(img, ow, oh), want)
    assert np.array_equal(D.resize_restated(img, ow, oh), want)


def test_tables_match_restatement_including_identity():
    for n_in, n_out in [(256, 256), (1024, 256), (178, 256), (3, 7), (7, 3), (1, 5), (5, 1)]:
        b, k = lib_table(n_in, n_out)
        if n_in == n_out:
            assert np.array_equal(b[:, 0], np.arange(n_out)) and np.all(b[:, 1] == 1) and np.all(k[:, 0] == 1 << 22)
        else:
            rb, rk = D._coeffs(n_in, n_out)
            assert np.array_equal(b, rb) and np.array_equal(k, rk)
        assert k.shape[1] == L.lib.aclgan_image_resample_ksize(n_in, n_out)
    assert L.lib.aclgan_image_resample_ksize(0, 5) == 0
    assert L.lib.aclgan_image_resample_coeffs(0, 5, None, None) != 0 and "positive" in L.last_error()


def test_oracle_against_golden_vectors():
    g = np.load(os.path.join(GOLDEN, "data_vectors.npz"))
    n = 0
    while "img%d" % n in g.files:
        img = g["img%d" % n]
        ns, ch, cw, flip, i, j = [int(v) for v in g["par%d" % n]]
        got = D.transform(img, ns, ch, cw, bool(flip), i, j).numpy()
        assert np.array_equal(got, g["out%d" % n]), n
        h, w = img.shape[:2]
        ow, oh = D.resized_size(w, h, ns)
        assert np.array_equal(D.resize_restated(img, ow, oh), g["res%d" % n]), n
        assert np.array_equal(two_pass(img, ow, oh), g["res%d" % n]), n
        n += 1
    assert n == 4


def test_oracle_chain_on_a_hand_case():
    """no Resize (smaller edge already new_size): the chain is flip -> crop -> (v/255 - 0.5)/0.5, checkable by slicing"""
    img = np.random.default_rng(3).integers(0, 256, (16, 24, 3), dtype=np.uint8)
    t = D.transform(img, 16, 8, 10, True, 3, 5)
    want = torch.from_numpy(img[:, ::-1][3:11, 5:15].copy()).permute(2, 0, 1).float().div(255).sub(0.5).div(0.5)
    assert torch.equal(t, want) and t.shape == (3, 8, 10)
    assert D.resized_size(300, 200, 100) == (150, 100) and D.resized_size(200, 300, 100) == (100, 150)
    assert D.resized_size(100, 300, 100) == (100, 300) and D.resized_size(7, 5, None) == (7, 5)
    with pytest.raises(ValueError):
        D.transform(img, 16, 32, 32, False, 0, 0)


def test_dataset_discovery_and_draw_order(tmp_path):
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / "trainA" / "sub")
    for name in ["b.png", "a.jpg", "sub/c.PNG", "notes.txt"]:
        p = tmp_path / "trainA" / name
        if name.endswith(".txt"):
            p.write_text("x")
        else:
            Image.fromarray(rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)).save(p)
    ds = PD.ImageFolder(str(tmp_path / "trainA"))
    assert [os.path.relpath(p, tmp_path / "trainA") for p in ds.imgs] == ["a.jpg", "b.png", "sub/c.PNG"]   # sorted, recursive, images only
    assert ds[0].mode == "RGB" and ds[0].size == (30, 20)
    os.makedirs(tmp_path / "empty")
    with pytest.raises(RuntimeError, match="Found 0 images"):
        PD.ImageFolder(str(tmp_path / "empty"))
    (tmp_path / "list.txt").write_text("b.png\nsub/c.PNG\n")
    fl = PD.ImageFilelist(str(tmp_path / "trainA"), str(tmp_path / "list.txt"))
    assert len(fl) == 2 and fl[1].size == (30, 20)

    # torchvision's draw order: flip (random.random), then RandomCrop.get_params (randint i, randint j)
    tf = PD.GpuBatchTransform(40, 32, 32, train=True, device="cpu")
    random.seed(7)
    flip, ow, oh, i, j, th, tw = tf.draw(30, 20)
    random.seed(7)
    e_flip = random.random() < 0.5; e_i = random.randint(0, 40 - 32); e_j = random.randint(0, 60 - 32)
    assert (ow, oh) == (60, 40) and (flip, i, j, th, tw) == (e_flip, e_i, e_j, 32, 32)
    # sizes match -> (0, 0) and NO draw; eval loaders never flip
    te = PD.GpuBatchTransform(20, 20, 30, train=False, device="cpu")
    random.seed(1); s0 = random.getstate()
    assert te.draw(30, 20) == (False, 30, 20, 0, 0, 20, 30) and random.getstate() == s0
    with pytest.raises(ValueError, match="smaller than the crop"):
        PD.GpuBatchTransform(20, 64, 64, train=False, device="cpu").draw(30, 20)


def test_transform_abi_validates_before_launch():
    d = (L.ImageDesc * 1)()
    assert L.lib.aclgan_image_batch_transform(None, d, None, 1, None, None, 8, 8, None) != 0
    assert "null buffer" in L.last_error()


def test_inference_script_host_helpers(tmp_path):
    """test.py's host side (reference test.py:89-93,110-124): Resize(new_size) + ToTensor + Normalize on load, and
    torchvision.utils.save_image(..., normalize=True) on store (min-max scaling, *255 + 0.5, clamp, uint8)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("aclgan_test_script_cpu", os.path.join(ROOT, "test.py"))
    script = importlib.util.module_from_spec(spec); spec.loader.exec_module(script)
    img = np.random.default_rng(4).integers(0, 256, (30, 45, 3), dtype=np.uint8)
    path = str(tmp_path / "in.png")
    Image.fromarray(img).save(path)
    t = script.load_image(path, 20)                                   # smaller edge 30 -> 20, width 45 -> 30
    assert t.shape == (1, 3, 20, 30)
    assert torch.equal(t[0], D.transform(img, 20, 0, 0, False, 0, 0, crop=False))
    assert script.resize_smaller_edge(Image.fromarray(img), 30).size == (45, 30)      # already the right size: untouched
    x = torch.linspace(-0.3, 0.9, 3 * 4 * 5).reshape(1, 3, 4, 5)
    out = str(tmp_path / "o" / "x.png")
    script.save_image(x, out)
    got = np.asarray(Image.open(out))
    lo, hi = float(x.min()), float(x.max())
    want = ((x[0] - lo) / (hi - lo + 1e-5)).mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    assert np.array_equal(got, want)
    # the blend of test.py:73-76 equals the training-time formula (trainer.py:85-88) up to rounding
    fg, bg, m = torch.rand(2, 3, 4, 4) * 2 - 1, torch.rand(2, 3, 4, 4) * 2 - 1, torch.rand(2, 1, 4, 4) * 2 - 1
    mm = ((m + 1) / 2).repeat(1, 3, 1, 1)
    assert torch.allclose(script.focus_translation(fg, bg, m), fg * mm + bg * (1 - mm), atol=1e-6)


def test_sharded_loaders_of_the_two_domains_shuffle_independently(tmp_path):
    """data-parallel epochs: ranks agree on one permutation per loader and epoch, the A and the B loader draw DIFFERENT ones (with one
    shared seed, equally long folders would pair A[i] with B[i] at every step), the permutation follows the user's torch seed, and a
    resumed run continues with the next epoch's permutation instead of replaying epoch 0."""
    from PIL import Image
    import aclgan_amd  # noqa: F401
    from aclgan_amd import data as D
    for sub in ("trainA", "trainB", "testA", "testB"):
        os.makedirs(tmp_path / sub)
        for i in range(12):
            Image.new("RGB", (8, 8), (i, i, i)).save(tmp_path / sub / ("%02d.png" % i))
    conf = dict(batch_size=2, num_workers=1, new_size=8, crop_image_height=8, crop_image_width=8, data_root=str(tmp_path))

    def order(loader):
        return [i for b in loader.batch_indices() for i in b]

    torch.manual_seed(1234)
    a0, b0, _, _ = D.get_all_data_loaders(conf, device="cpu", rank=0, world_size=2)
    a1, b1, _, _ = D.get_all_data_loaders(conf, device="cpu", rank=1, world_size=2)
    ea0, ea1, eb0 = order(a0), order(a1), order(b0)
    assert sorted(ea0 + ea1) == list(range(12)) and not set(ea0) & set(ea1)      # disjoint shards of one permutation
    full_a = [i for pair in zip(*[iter(ea0)] * 2, *[iter(ea1)] * 2) for i in pair]
    eb1 = order(b1)
    full_b = [i for pair in zip(*[iter(eb0)] * 2, *[iter(eb1)] * 2) for i in pair]
    assert sorted(full_b) == list(range(12)) and full_a != full_b                 # A and B are not paired index by index
    second = order(a0)
    assert second != ea0                                                          # next epoch, next permutation
    torch.manual_seed(1234)
    a0r, _, _, _ = D.get_all_data_loaders(conf, device="cpu", rank=0, world_size=2)
    a0r.set_epoch(1)
    assert order(a0r) == second                                                   # resume at epoch 1 = what the uninterrupted run drew
    torch.manual_seed(99)
    a0s, _, _, _ = D.get_all_data_loaders(conf, device="cpu", rank=0, world_size=2)
    assert order(a0s) != ea0                                                      # the user's seed matters
