"""Training input pipeline on the MI355X.

Mirror of the reference's loader surface:
  get_all_data_loaders(conf)            utils.py:43-76
  get_data_loader_folder / _list(...)   utils.py:78-100
  ImageFolder / ImageFilelist           data.py:30-47, 95-125 (same file discovery, same ordering)
with one difference in WHERE the work runs: the reference decodes AND transforms every sample on the host
(torchvision transforms on PIL images inside DataLoader workers); here the host only decodes (a thread pool,
PIL releases the GIL) and draws the random numbers, and ONE HIP kernel per batch does
flip -> Resize(new_size) -> crop -> ToTensor -> Normalize on the device, bit-identical to the PIL/torchvision
result (csrc/image.hip, C ABI aclgan_image_batch_transform).  A loader yields float32 [B,3,H,W] CUDA tensors in
[-1, 1] -- exactly what aclgan_Trainer.dis_update / gen_update take.

Random draws follow torchvision 0.4.0 (the reference's pin): RandomHorizontalFlip: random.random() < 0.5;
RandomCrop.get_params: (0, 0) when the sizes match, else i = random.randint(0, h - th), j = random.randint(0, w - tw).
Shuffling (DataLoader(shuffle=train, drop_last=True)) uses torch.randperm.
"""
import ctypes as C
import os
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib as L

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP']   # data.py:72-75


def default_loader(path):                      # data.py:12-13
    from PIL import Image
    return Image.open(path).convert('RGB')


def default_flist_reader(flist):               # data.py:16-27: one path per line
    with open(flist, 'r') as rf:
        return [line.strip() for line in rf.readlines()]


def is_image_file(filename):                   # data.py:78-79
    return any(filename.endswith(e) for e in IMG_EXTENSIONS)


def make_dataset(dir):                         # data.py:82-92
    assert os.path.isdir(dir), '%s is not a valid directory' % dir
    images = []
    for root, _, fnames in sorted(os.walk(dir)):
        for fname in fnames:
            if is_image_file(fname):
                images.append(os.path.join(root, fname))
    return images


class ImageFolder:
    """data.py:95-125 without the transform hook: __getitem__ returns the decoded PIL image."""

    def __init__(self, root, return_paths=False, loader=default_loader):
        imgs = sorted(make_dataset(root))
        if len(imgs) == 0:
            raise RuntimeError("Found 0 images in: " + root + "\nSupported image extensions are: " + ",".join(IMG_EXTENSIONS))
        self.root, self.imgs, self.return_paths, self.loader = root, imgs, return_paths, loader

    def __getitem__(self, index):
        img = self.loader(self.imgs[index])
        return (img, self.imgs[index]) if self.return_paths else img

    def __len__(self):
        return len(self.imgs)


class ImageFilelist:
    """data.py:30-47"""

    def __init__(self, root, flist, flist_reader=default_flist_reader, loader=default_loader):
        self.root, self.imlist, self.loader = root, flist_reader(flist), loader

    def __getitem__(self, index):
        return self.loader(os.path.join(self.root, self.imlist[index]))

    def __len__(self):
        return len(self.imlist)


def resized_size(w, h, size):
    """torchvision.transforms.functional.resize with an int size: smaller edge -> size (returns (ow, oh))"""
    if size is None or (w <= h and w == size) or (h <= w and h == size):
        return w, h
    if w < h:
        return size, int(size * h / w)
    return int(size * w / h), size


class GpuBatchTransform:
    """RandomHorizontalFlip (train) -> Resize(new_size) -> RandomCrop((height, width)) -> ToTensor -> Normalize on the
    device, for a list of decoded uint8 HWC arrays (utils.py:78-86 / 91-99)."""

    def __init__(self, new_size, height, width, train, crop=True, device="cuda"):
        self.new_size, self.height, self.width, self.train, self.crop = new_size, height, width, train, crop
        self.device = torch.device(device)
        self._tables = {}          # (in, out) -> (ksize, int32 array [out*2 + out*ksize])

    def _table(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            ks = L.lib.aclgan_image_resample_ksize(n_in, n_out)
            buf = np.zeros(n_out * 2 + n_out * ks, np.int32)
            ip = C.POINTER(C.c_int)
            L.check(L.lib.aclgan_image_resample_coeffs(n_in, n_out, buf[:2 * n_out].ctypes.data_as(ip), buf[2 * n_out:].ctypes.data_as(ip)),
                    "image_resample_coeffs")
            self._tables[key] = (ks, buf)
        return self._tables[key]

    def draw(self, w, h):
        """the random decisions for one sample, in torchvision's order; returns (flip, ow, oh, i, j, th, tw)"""
        flip = bool(self.train and random.random() < 0.5)              # RandomHorizontalFlip first (utils.py:84)
        ow, oh = resized_size(w, h, self.new_size)
        th, tw = (self.height, self.width) if self.crop else (oh, ow)
        if ow == tw and oh == th:
            i = j = 0
        else:
            if oh < th or ow < tw:
                raise ValueError("RandomCrop: image %dx%d (after Resize) is smaller than the crop %dx%d" % (oh, ow, th, tw))
            i = random.randint(0, oh - th)
            j = random.randint(0, ow - tw)
        return flip, ow, oh, i, j, th, tw

    def stage(self, images, params=None):
        """host half: random draws, descriptors, tables, ONE pinned staging buffer [pixels | descriptors | tables]
        and one H2D copy.  images: list of uint8 [H][W][3] arrays (or PIL RGB images); params: optional explicit
        (flip, i, j) per image (parity tests).  Returns the launch state for launch()."""
        arrs = [np.ascontiguousarray(np.asarray(im, dtype=np.uint8)) for im in images]
        n = len(arrs)
        descs = (L.ImageDesc * n)()
        tabs, tab_off, tab_idx = [], 0, {}
        off = 0
        out_hw = None
        for k, a in enumerate(arrs):
            assert a.ndim == 3 and a.shape[2] == 3, "RGB HWC uint8 images expected"
            h, w = a.shape[:2]
            if params is None:
                flip, ow, oh, i, j, th, tw = self.draw(w, h)
            else:
                flip, i, j = params[k]
                ow, oh = resized_size(w, h, self.new_size)
                th, tw = (self.height, self.width) if self.crop else (oh, ow)
            if out_hw is None:
                out_hw = (th, tw)
            elif out_hw != (th, tw):
                raise ValueError("batch elements have different output sizes (%s vs %s): crop=False needs equal image sizes" % (out_hw, (th, tw)))
            d = descs[k]
            d.src_offset, d.src_h, d.src_w, d.res_h, d.res_w = off, h, w, oh, ow
            d.crop_y, d.crop_x, d.flip = i, j, int(flip)
            for axis, (n_in, n_out) in (("x", (w, ow)), ("y", (h, oh))):
                ks, buf = self._table(n_in, n_out)
                if (n_in, n_out) not in tab_idx:
                    tab_idx[(n_in, n_out)] = tab_off
                    tabs.append(buf)
                    tab_off += buf.size
                setattr(d, "tab_" + axis, tab_idx[(n_in, n_out)])
                setattr(d, "ksize_" + axis, ks)
            off += a.size
        nb_pix = (off + 15) // 16 * 16
        nb_desc = C.sizeof(descs)
        tab = np.concatenate(tabs)
        total = nb_pix + (nb_desc + 15) // 16 * 16 + tab.nbytes
        host = torch.empty(total, dtype=torch.uint8)
        if self.device.type == "cuda":
            host = host.pin_memory()
        hv = host.numpy()
        p = 0
        for a in arrs:
            hv[p:p + a.size] = a.reshape(-1); p += a.size
        o_desc = nb_pix
        hv[o_desc:o_desc + nb_desc] = np.frombuffer(bytes(descs), dtype=np.uint8)
        o_tab = o_desc + (nb_desc + 15) // 16 * 16
        hv[o_tab:o_tab + tab.nbytes] = tab.view(np.uint8)
        dev = host.to(self.device, non_blocking=True)
        return {"descs": descs, "dev": dev, "o_desc": o_desc, "o_tab": o_tab, "n": n, "out_hw": out_hw}

    def launch(self, st, stream=None):
        """device half: one kernel for the whole batch -> float32 [n][3][th][tw]"""
        th, tw = st["out_hw"]
        out = torch.empty(st["n"], 3, th, tw, device=self.device, dtype=torch.float32)
        base = st["dev"].data_ptr()
        L.check(L.lib.aclgan_image_batch_transform(C.c_void_p(base), st["descs"], C.c_void_p(base + st["o_desc"]), st["n"],
                                                   C.c_void_p(base + st["o_tab"]), L.ptr(out), th, tw,
                                                   stream if stream is not None else L.stream_ptr()), "image_batch_transform")
        # (the staging buffers are released stream-ordered by torch's caching allocators: safe without a sync)
        return out

    def __call__(self, images, params=None, stream=None):
        return self.launch(self.stage(images, params), stream)


class GpuImageLoader:
    """DataLoader(dataset, batch_size, shuffle=train, drop_last=True, num_workers) counterpart: iterating yields
    device batches.  `.dataset[i]` returns the transformed sample i as a [3,H,W] device tensor (train.py:44-47)."""

    def __init__(self, dataset, batch_size, train, new_size, height, width, num_workers=4, crop=True, device="cuda", rank=0, world_size=1,
                 shard_seed=0):
        """rank / world_size (data parallel, not in the reference): batch_size is PER RANK; every epoch all ranks draw the SAME
        permutation (generator seeded from shard_seed and the epoch) and rank r takes slice r of each global batch of world_size *
        batch_size samples -- disjoint shards, every sample at most once per epoch, equal batch counts on all ranks.
        shard_seed must be equal on all ranks and DIFFERENT for the A and the B loader (get_all_data_loaders derives both from the
        user's torch seed): with one seed for both, equally long folders (selfie2anime: 3400 / 3400) would pair A[i] with B[i] at
        every step of every epoch, where the reference's two DataLoaders shuffle independently (utils.py:91-100)."""
        self.source, self.batch_size, self.train = dataset, batch_size, train
        self.rank, self.world, self.shard_seed, self._epoch = int(rank), max(1, int(world_size)), int(shard_seed), 0
        assert 0 <= self.rank < self.world
        self.transform = GpuBatchTransform(new_size, height, width, train, crop, device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, num_workers))
        self.dataset = _TransformedView(self)

    def __len__(self):
        return len(self.source) // (self.batch_size * self.world)          # drop_last=True

    def batch_indices(self):
        """the sample indices of this rank's batches for the next epoch (advances the epoch counter)"""
        n = len(self.source)
        if not self.train:
            order = list(range(n))
        elif self.world == 1:
            order = torch.randperm(n).tolist()          # the default generator, like DataLoader(shuffle=True)
        else:
            order = torch.randperm(n, generator=torch.Generator().manual_seed((self.shard_seed + 1000003 * self._epoch) % (2 ** 63 - 1))).tolist()
        self._epoch += 1
        gb = self.batch_size * self.world
        return [order[b * gb + self.rank * self.batch_size: b * gb + (self.rank + 1) * self.batch_size] for b in range(len(self))]

    def set_epoch(self, epoch):
        """resume support: the next pass draws the permutation of epoch `epoch` (train.py derives it from the restored iteration count,
        so a resumed data-parallel run does not replay the permutations of epochs 0, 1, ...)"""
        self._epoch = int(epoch)

    def __iter__(self):
        batches = self.batch_indices()
        nxt = self._decode(batches[0]) if batches else None
        for b in range(len(batches)):
            cur = nxt
            nxt = self._decode(batches[b + 1]) if b + 1 < len(batches) else None    # decode of batch b+1 overlaps the step on batch b
            yield self.transform([f.result() for f in cur])

    def _decode(self, idxs):
        return [self.pool.submit(self.source.__getitem__, i) for i in idxs]


class _TransformedView:
    def __init__(self, loader):
        self._l = loader

    def __len__(self):
        return len(self._l.source)

    def __getitem__(self, i):
        return self._l.transform([self._l.source[i]])[0]


def get_data_loader_folder(input_folder, batch_size, train, new_size=None, height=256, width=256, num_workers=4, crop=True,
                           datakind='', device="cuda", rank=0, world_size=1, shard_seed=0):
    """utils.py:91-100"""
    return GpuImageLoader(ImageFolder(input_folder), batch_size, train, new_size, height, width, num_workers, crop, device, rank, world_size, shard_seed)


def get_data_loader_list(root, file_list, batch_size, train, new_size=None, height=256, width=256, num_workers=4, crop=True,
                         datakind='', device="cuda", rank=0, world_size=1, shard_seed=0):
    """utils.py:78-89"""
    return GpuImageLoader(ImageFilelist(root, file_list), batch_size, train, new_size, height, width, num_workers, crop, device, rank, world_size, shard_seed)


def shard_base_seed(world_size=1):
    """the seed the sharded loaders derive their per-epoch permutations from: the user's torch seed (torch.manual_seed / the process
    default), agreed across ranks -- rank 0's value is broadcast when a process group exists (ranks that never called
    torch.manual_seed hold different random default seeds)"""
    seed = int(torch.initial_seed()) % (2 ** 62)
    if world_size > 1:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            box = [seed]
            dist.broadcast_object_list(box, src=0)
            seed = int(box[0])
    return seed


def get_all_data_loaders(conf, device="cuda", rank=0, world_size=1, base_seed=None):
    """utils.py:43-76: (train_a, train_b, test_a, test_b); rank / world_size shard the TRAIN loaders (see GpuImageLoader); the A and B
    train loaders get different shard seeds derived from base_seed (default: shard_base_seed(world_size))"""
    if base_seed is None:
        base_seed = shard_base_seed(world_size)
    seed_of = {"a": (2 * base_seed + 1) % (2 ** 62), "b": (2 * base_seed + 0x5bd1e995) % (2 ** 62)}
    batch_size, num_workers = conf['batch_size'], conf['num_workers']
    if 'new_size' in conf:
        new_size_a = new_size_b = conf['new_size']
    else:
        new_size_a, new_size_b = conf['new_size_a'], conf['new_size_b']
    height, width = conf['crop_image_height'], conf['crop_image_width']
    datakind = conf.get('data_kind', '')
    if 'data_root' in conf:
        r = conf['data_root']
        mk = lambda sub, train, ns, h, w: get_data_loader_folder(os.path.join(r, sub), batch_size, train, ns, h, w, num_workers, True, datakind, device,
                                                                 rank if train else 0, world_size if train else 1, seed_of[sub[-1].lower()])
        return (mk('trainA', True, new_size_a, height, width), mk('trainB', True, new_size_b, height, width),
                mk('testA', False, new_size_a, new_size_a, new_size_a), mk('testB', False, new_size_b, new_size_b, new_size_b))
    mk = lambda f, l, train, ns, h, w: get_data_loader_list(conf[f], conf[l], batch_size, train, ns, h, w, num_workers, True, datakind, device,
                                                            rank if train else 0, world_size if train else 1, seed_of[f[-1].lower()])
    return (mk('data_folder_train_a', 'data_list_train_a', True, new_size_a, height, width),
            mk('data_folder_train_b', 'data_list_train_b', True, new_size_b, height, width),
            mk('data_folder_test_a', 'data_list_test_a', False, new_size_a, new_size_a, new_size_a),
            mk('data_folder_test_b', 'data_list_test_b', False, new_size_b, new_size_b, new_size_b))
