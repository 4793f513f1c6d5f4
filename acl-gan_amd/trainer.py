"""aclgan_Trainer -- drop-in counterpart of the reference's trainer.aclgan_Trainer
(reference trainer.py:14-331) running on libaclgan_hip.so.

Same constructor argument (the YAML dict), same methods (gen_update / dis_update / sample /
update_learning_rate / save / resume), same ``loss_*`` attributes (0-d tensors), same attribute
names for the five networks, same checkpoint file names / dict keys / state_dict key names
(OIHW fp32 weights on disk; the OHWI <-> OIHW repack happens here, at the boundary).

PyTorch's role: device memory (flat parameter / gradient / Adam-state buffers, workspace),
the current stream, torch.save/torch.load, torch.distributed.  All arithmetic of the step is in
the HIP library; if it is missing the import of this package fails -- there is no fallback.
"""
import ctypes as C
import math
import os
from collections import OrderedDict

import torch

from . import _lib as L

__all__ = ["aclgan_Trainer", "AdaINGen", "MsImageDis", "arch_from_config", "hparams_from_config"]


def arch_from_config(hp):
    g, d = hp["gen"], hp["dis"]
    for key, want in (("activ", "relu"), ("pad_type", "reflect")):
        if g.get(key, want) != want:
            raise L.AclganError("gen.%s=%r unsupported (only %r is reached by the shipped config)" % (key, g.get(key), want))
    for key, want in (("activ", "lrelu"), ("pad_type", "reflect"), ("norm", "none"), ("gan_type", "lsgan")):
        if d.get(key, want) != want:
            raise L.AclganError("dis.%s=%r unsupported (only %r is reached by the shipped config)" % (key, d.get(key), want))
    return L.Arch(int(hp["input_dim_a"]), int(hp["input_dim_b"]), int(g["dim"]), int(g["mlp_dim"]), int(g["style_dim"]),
                  int(g["output_dim"]), int(g["n_downsample"]), int(g["n_res"]), int(d["dim"]), int(d["n_layer"]),
                  int(d["num_scales"]))


def hparams_from_config(hp):
    """The per-call hyper-parameters gen_update / dis_update read (trainer.py:93-97,142-144,164,288-290)."""
    return L.HParams(float(hp["gan_w"]), float(hp["gan_cw"]), float(hp["recon_x_w"]), float(hp["focus_loss"]),
                     float(hp["focus_delta"]), float(hp["focus_upper"]), float(hp["focus_lower"]),
                     float(hp["focus_epsilon"]), float(hp.get("alpha", 1)))


class _Net:
    """A view of one network's tensors inside the flat group buffers, with the reference's
    state_dict surface."""

    def __init__(self, trainer, name):
        self._t = trainer
        self.name = name
        self.net_id = L.NETS[name]
        self.group = L.GROUP_GEN if name.startswith("gen") else L.GROUP_DIS
        self._entries = [e for e in trainer._tensors[self.group] if e["net"] == name]

    # ---- tensor access ----
    def _view(self, e, buf):
        flat = buf[e["offset"]: e["offset"] + e["numel"]]
        shp = e["shape"]
        if len(shp) == 4:   # OHWI in memory -> OIHW view (what the reference stores)
            co, ci, kh, kw = shp
            return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat.view(*shp)

    def named_parameters(self):
        for e in self._entries:
            yield e["key"], self._view(e, self._t._param[self.group])

    def named_grads(self):
        for e in self._entries:
            yield e["key"], self._view(e, self._t._grad[self.group])

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def state_dict(self):
        sd = OrderedDict()
        params = OrderedDict((k, v.contiguous().clone()) for k, v in self.named_parameters())
        bufs = self._buffers()
        # reference ordering: per module, parameters then buffers
        for k in self._t._ref_key_order(self.name, list(params.keys()), list(bufs.keys())):
            sd[k] = params[k] if k in params else bufs[k]
        return sd

    def _buffers(self):
        return OrderedDict()

    def load_state_dict(self, sd, strict=True):
        mine = OrderedDict(self.named_parameters())
        missing = [k for k in mine if k not in sd]
        unexpected = [k for k in sd if k not in mine and k not in self._buffers()]
        if strict and (missing or unexpected):
            raise L.AclganError("load_state_dict(%s): missing %s unexpected %s" % (self.name, missing, unexpected))
        with torch.no_grad():
            for k, v in mine.items():
                if k in sd:
                    src = sd[k].to(device=v.device, dtype=torch.float32)
                    if tuple(src.shape) != tuple(v.shape):
                        raise L.AclganError("load_state_dict(%s): shape mismatch for %s: %s vs %s" % (self.name, k, tuple(src.shape), tuple(v.shape)))
                    v.copy_(src)
        for k in self._buffers():
            if k in sd:
                self._t._dummy_buffers[self.name][k] = sd[k].detach().clone().cpu()

    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self


class AdaINGen(_Net):
    """reference networks.py:112-171 -- encode / decode on the HIP forward path."""

    def _buffers(self):
        return self._t._dummy_buffers[self.name]

    def encode(self, images):
        t = self._t
        images = images.to(t.device, torch.float32).contiguous()
        B, Cin, H, W = images.shape
        a = t.arch
        q = 1 << a.gen_n_downsample
        content = torch.empty(B, a.gen_dim * q, H // q, W // q, device=t.device)
        style = torch.empty(B, a.gen_style_dim, 1, 1, device=t.device)
        with torch.cuda.device(t.device):
            t._ensure_workspace(B, H, W, forward_only=True)
            L.check(L.lib.aclgan_gen_encode(t._ctx, self.net_id, L.ptr(images), B, H, W, L.ptr(content), L.ptr(style), t._st()), "gen_encode")
        return content, style

    def decode(self, content, style):
        t = self._t
        content = content.to(t.device, torch.float32).contiguous()
        style = style.to(t.device, torch.float32).contiguous()
        B, Cc, h, w = content.shape
        a = t.arch
        q = 1 << a.gen_n_downsample
        out = torch.empty(B, a.gen_output_dim, h * q, w * q, device=t.device)
        with torch.cuda.device(t.device):
            t._ensure_workspace(B, h * q, w * q, forward_only=True)
            L.check(L.lib.aclgan_gen_decode(t._ctx, self.net_id, L.ptr(content), L.ptr(style), B, h, w, L.ptr(out), t._st()), "gen_decode")
        return out

    def forward(self, images):
        content, style = self.encode(images)
        return self.decode(content, style)

    __call__ = forward


class MsImageDis(_Net):
    """reference networks.py:21-106 -- forward (list of per-scale maps) on the HIP path."""

    def forward(self, x):
        t = self._t
        x = x.to(t.device, torch.float32).contiguous()
        B, Cin, H, W = x.shape
        a = t.arch
        outs = []
        h, w = H, W
        for s in range(a.dis_num_scales):
            hs, ws = h, w
            for _ in range(a.dis_n_layer):
                hs, ws = (hs + 2 - 4) // 2 + 1, (ws + 2 - 4) // 2 + 1
            outs.append(torch.empty(B, 1, hs, ws, device=t.device))
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        arr = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        with torch.cuda.device(t.device):
            t._ensure_workspace(B, H, W, forward_only=True)
            L.check(L.lib.aclgan_dis_forward(t._ctx, self.net_id, L.ptr(x), B, H, W, arr, t._st()), "dis_forward")
        return outs

    __call__ = forward


class aclgan_Trainer:
    """See module docstring.  Extra (optional) arguments over the reference: ``device``; and
    ``z=(z_1, z_2, z_3)`` on the update calls for seed-independent parity tests (by default z is
    drawn from the CPU generator exactly like trainer.py:99-101)."""

    def __init__(self, hyperparameters, device=None, compute_dtype=None, deterministic=None, hip_graph=None):
        if not torch.cuda.is_available():
            raise L.AclganError("aclgan_Trainer needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        hp = hyperparameters
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.arch = arch_from_config(hp)
        # compute dtype of the heavy convolutions (not in the reference, which is fp32 only): "fp32" | "bf16" | "fp16"
        self.compute_dtype = str(compute_dtype or hp.get("compute_dtype", "fp32"))
        if self.compute_dtype not in L.DTYPE:
            raise L.AclganError("compute_dtype=%r: expected one of %s" % (self.compute_dtype, sorted(L.DTYPE)))
        # deterministic=True (or config key "deterministic", or ACLGAN_DETERMINISTIC=1): every reduction takes an ordered path and a step
        # is reproducible bit for bit run to run (include/aclgan_hip.h: aclgan_set_deterministic).  Process-wide, like
        # torch.use_deterministic_algorithms; must be chosen before the workspaces are sized, i.e. here.
        det = deterministic if deterministic is not None else hp.get("deterministic", None)
        if det is not None:
            L.check(L.lib.aclgan_set_deterministic(1 if det else 0), "set_deterministic")
        self.deterministic = bool(L.lib.aclgan_get_deterministic())
        # hip_graph=True (or config key "hip_graph", or ACLGAN_GRAPH=1): the launch sequence of an update (zero_grad + forward +
        # losses + backward, ~1 000-2 000 kernels) is captured once per (shape, hyper-parameters, workspace) into a HIP graph and
        # replayed; Adam stays a separate launch (its step count is a kernel argument).  Same kernels in the same order: same
        # results.  Pays when the step is launch-bound (small batches / images); off by default.
        hg = hip_graph if hip_graph is not None else hp.get("hip_graph", os.environ.get("ACLGAN_GRAPH", "0") not in ("", "0"))
        self.hip_graph = bool(hg)
        self._graphs = {}
        self._ctx = C.c_void_p()
        L.check(L.lib.aclgan_ctx_create(C.byref(self.arch), C.byref(self._ctx)), "ctx_create")
        L.check(L.lib.aclgan_set_compute_dtype(self._ctx, L.DTYPE[self.compute_dtype]), "set_compute_dtype")
        # the lane scheduler's streams first, before anything else of this process creates one (the rank-0 broadcast below initialises the process
        # group's communicator and its stream): HIP binds streams to hardware queues in creation order (include/aclgan_hip.h, aclgan_warm_streams)
        if os.environ.get("ACLGAN_WARM_STREAMS", "1") not in ("", "0"):
            with torch.cuda.device(self.device):
                L.check(L.lib.aclgan_warm_streams(0), "warm_streams")
        if self.hip_graph:      # (a captured update runs on one lane with a parameter-gradient stream of its own: created here, outside any capture)
            with torch.cuda.device(self.device):
                L.check(L.lib.aclgan_ctx_enable_capture(self._ctx), "ctx_enable_capture")
        self.style_dim = hp["gen"]["style_dim"]
        self.alpha = hp["alpha"]
        self.focus_lam = hp["focus_loss"]
        self._hp = hp
        # ---- flat buffers (one contiguous param / grad / exp_avg / exp_avg_sq per optimizer) ----
        self._tensors, self._param, self._grad, self._m, self._v = {}, {}, {}, {}, {}
        self._w16, self._w16t = {}, {}
        for grp in (L.GROUP_GEN, L.GROUP_DIS):
            n = L.lib.aclgan_group_numel(self._ctx, grp)
            ents = []
            name = C.create_string_buffer(256)
            off = C.c_int64(); shp = (C.c_int * 4)(); nd = C.c_int()
            for i in range(L.lib.aclgan_tensor_count(self._ctx, grp)):
                L.check(L.lib.aclgan_tensor_info(self._ctx, grp, i, name, 256, C.byref(off), shp, C.byref(nd)))
                full = name.value.decode()
                net, key = full.split("/", 1)
                shape = tuple(shp[j] for j in range(nd.value))
                ents.append(dict(net=net, key=key, offset=off.value, shape=shape, numel=int(math.prod(shape))))
            self._tensors[grp] = ents
            self._param[grp] = torch.zeros(n, device=self.device)
            self._grad[grp] = torch.zeros(n, device=self.device)
            self._m[grp] = torch.zeros(n, device=self.device)
            self._v[grp] = torch.zeros(n, device=self.device)
            L.check(L.lib.aclgan_bind_params(self._ctx, grp, L.ptr(self._param[grp]), L.ptr(self._grad[grp]),
                                             L.ptr(self._m[grp]), L.ptr(self._v[grp])), "bind_params")
            if self.compute_dtype != "fp32":   # 16-bit weight packs, refreshed by the library from the fp32 master copy
                self._w16[grp] = torch.zeros(n, dtype=torch.int16, device=self.device)
                self._w16t[grp] = torch.zeros(n, dtype=torch.int16, device=self.device)
                L.check(L.lib.aclgan_bind_params16(self._ctx, grp, L.ptr(self._w16[grp]), L.ptr(self._w16t[grp])), "bind_params16")
        d = self.arch.gen_dim << self.arch.gen_n_downsample
        self._dummy_buffers = {}
        for net in ("gen_AB", "gen_BA"):   # AdaIN running_mean/var: never used, but in the state_dict (networks.py:488-489)
            b = OrderedDict()
            for r in range(self.arch.gen_n_res):
                for j in range(2):
                    b["dec.model.0.model.%d.model.%d.norm.running_mean" % (r, j)] = torch.zeros(d)
                    b["dec.model.0.model.%d.model.%d.norm.running_var" % (r, j)] = torch.ones(d)
            self._dummy_buffers[net] = b
        self.gen_AB, self.gen_BA = AdaINGen(self, "gen_AB"), AdaINGen(self, "gen_BA")
        self.dis_A, self.dis_B, self.dis_2 = MsImageDis(self, "dis_A"), MsImageDis(self, "dis_B"), MsImageDis(self, "dis_2")
        # fixed display noise (trainer.py:30-32)
        ds = int(hp["display_size"])
        self.z_1 = torch.randn(ds, self.style_dim, 1, 1).to(self.device)
        self.z_2 = torch.randn(ds, self.style_dim, 1, 1).to(self.device)
        self.z_3 = torch.randn(ds, self.style_dim, 1, 1).to(self.device)
        # optimizers (trainer.py:34-44): Adam + StepLR, one per group
        self._opt = {grp: dict(lr=float(hp["lr"]), beta1=float(hp["beta1"]), beta2=float(hp["beta2"]), eps=1e-8,
                               weight_decay=float(hp["weight_decay"]), steps=0) for grp in (L.GROUP_GEN, L.GROUP_DIS)}
        self._sched_calls = 0
        # weight init (trainer.py:48-52, utils.py:274-294)
        self._init_weights(hp.get("init", "kaiming"))
        self._ws = None
        self._ws_shape = None
        self._losses = torch.zeros(len(L.LOSS_NAMES), device=self.device)
        for n in L.LOSS_NAMES:
            setattr(self, n, torch.zeros((), device=self.device))
        self._zgen = None
        # fp16: dynamic loss scaling, state resident on the device (include/aclgan_hip.h: aclgan_bind_loss_scale)
        self._lscale = None
        if self.compute_dtype == "fp16":
            s0 = float(hp.get("loss_scale_init", 65536.0))
            self._lscale = torch.tensor([s0, 1.0 / s0, 0, 0, 0, 0, float(hp.get("loss_scale_growth_interval", 2000)), 0],
                                        dtype=torch.float32, device=self.device)
            L.check(L.lib.aclgan_bind_loss_scale(self._ctx, L.ptr(self._lscale)), "bind_loss_scale")
        # the scale each group's gradient buffers carry: a device-side copy of the live scale taken when that group's update starts
        # (grad_scale(grp)); the two groups are updated at different live scales whenever one of them overflowed in between
        self._gscale = None if self._lscale is None else torch.zeros(2, dtype=torch.float32, device=self.device)
        self._last_grp = None
        self._setup_data_parallel()

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                L.lib.aclgan_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # ---- plumbing: the nn.Module surface of the reference class (trainer.py:14) that callers can reach ----
    NETS = ("gen_AB", "gen_BA", "dis_A", "dis_B", "dis_2")

    def state_dict(self):
        """the reference trainer's own state_dict(): '<net>.<key>' for the five networks, in registration order
        (254 entries: tests/golden/state_dict_keys.txt)"""
        sd = OrderedDict()
        for n in self.NETS:
            for k, v in getattr(self, n).state_dict().items():
                sd[n + "." + k] = v
        return sd

    def load_state_dict(self, sd, strict=True):
        known = set()
        for n in self.NETS:
            sub = OrderedDict((k[len(n) + 1:], v) for k, v in sd.items() if k.startswith(n + "."))
            known.update(n + "." + k for k in sub)
            getattr(self, n).load_state_dict(sub, strict=strict)
        extra = [k for k in sd if k not in known]
        if strict and extra:
            raise L.AclganError("load_state_dict: unexpected keys %s" % extra[:5])

    def named_parameters(self):
        for n in self.NETS:
            for k, v in getattr(self, n).named_parameters():
                yield n + "." + k, v

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def zero_grad(self):
        for grp in (L.GROUP_GEN, L.GROUP_DIS):
            self._grad[grp].zero_()

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def _ref_key_order(self, net, pkeys, bkeys):
        # AdaIN blocks: norm buffers precede the conv parameters of the same Conv2dBlock
        out = []
        for k in pkeys:
            if k.endswith("conv.weight") and k.startswith("dec.model.0."):
                pre = k[: -len("conv.weight")]
                out += [pre + "norm.running_mean", pre + "norm.running_var"]
            out.append(k)
        assert set(out) == set(pkeys) | set(bkeys), (set(out) ^ (set(pkeys) | set(bkeys)))
        return out

    def _init_weights(self, kind):
        for grp, nets in ((L.GROUP_GEN, (self.gen_AB, self.gen_BA)), (L.GROUP_DIS, (self.dis_A, self.dis_B, self.dis_2))):
            for net in nets:
                for key, v in net.named_parameters():
                    if key.endswith(".weight"):
                        fan_in = int(math.prod(v.shape[1:]))
                        if grp == L.GROUP_DIS or kind == "gaussian":
                            w = torch.randn(tuple(v.shape)) * 0.02
                        elif kind == "kaiming":
                            w = torch.randn(tuple(v.shape)) * math.sqrt(2.0 / fan_in)
                        elif kind == "xavier":
                            fan_out = v.shape[0] * int(math.prod(v.shape[2:]))
                            w = torch.randn(tuple(v.shape)) * math.sqrt(2.0) * math.sqrt(2.0 / (fan_in + fan_out))
                        elif kind == "orthogonal":     # utils.py:285-286
                            w = torch.nn.init.orthogonal_(torch.empty(tuple(v.shape)), gain=math.sqrt(2.0))
                        elif kind == "default":        # utils.py:287-288: nn.Conv2d / nn.Linear's own reset_parameters() stays
                            w = torch.nn.init.kaiming_uniform_(torch.empty(tuple(v.shape)), a=math.sqrt(5.0))
                        else:
                            raise L.AclganError("Unsupported initialization: %s" % kind)
                        v.copy_(w.to(self.device))
                    elif key.endswith(".gamma"):
                        v.copy_(torch.rand(tuple(v.shape)).to(self.device))   # networks.py:517
                    else:
                        v.zero_()

    def _st(self):
        """the current stream of THIS trainer's device (not of torch's current device)"""
        return L.stream_ptr(self.device)

    def _ensure_workspace(self, B, H, W, forward_only=False):
        """Bind an arena large enough for one update (or, forward_only, for one encode / decode / discriminator
        forward: inference must not inherit the training step's shape constraints or its arena size)."""
        key = (B, H, W)
        have = self._ws_shape
        # (an update's need depends on the library's tuning switches -- lanes, batched transforms, kernel choices: the cached size is
        #  valid for one tuning epoch; after aclgan_tuning the next call sizes and binds again instead of failing with ACLGAN_ENOMEM)
        ep = C.c_longlong()
        L.check(L.lib.aclgan_tuning_get(b"epoch", C.byref(ep)), "tuning_get")
        if have is not None and have[0] >= B and have[1:3] == (H, W) and (forward_only or (have[3] and have[4] == ep.value)):
            return
        need = C.c_size_t()
        if forward_only:
            L.check(L.lib.aclgan_forward_workspace_bytes(self._ctx, B, H, W, C.byref(need)), "forward_workspace_bytes")
        else:
            L.check(L.lib.aclgan_workspace_bytes(self._ctx, B, H, W, C.byref(need)), "workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            try:
                self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            except torch.OutOfMemoryError:
                # arenas of trainers that are garbage but not yet collected (reference cycles through the bucket callbacks), and blocks
                # the caching allocator keeps reserved: release both and try once more before giving up
                import gc
                gc.collect(); torch.cuda.empty_cache()
                self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        L.check(L.lib.aclgan_bind_workspace(self._ctx, L.ptr(self._ws), self._ws.numel()), "bind_workspace")
        self._ws_shape = key + (not forward_only, ep.value)

    def loss_scale_state(self):
        """fp16 only: {'scale', 'clean_updates', 'skipped_gen', 'skipped_dis'} (one device->host copy)."""
        if self._lscale is None:
            return None
        v = self._lscale.cpu().tolist()
        return {"scale": v[0], "clean_updates": int(v[2]), "skipped_gen": int(v[4]), "skipped_dis": int(v[5])}

    def grad_scale(self, grp=None):
        """the factor a group's gradient buffers carry.  fp32 / bf16: 1.  fp16: the loss scale that group's last update ran with --
        the generator's and the discriminators' buffers differ whenever an overflow halved (or a clean run doubled) the live scale
        between the two updates.  grp: "gen" / "dis" (or GROUP_GEN / GROUP_DIS); None -> the group updated last.  A group that has not
        been updated yet reports the live scale (its buffers are zero).  One device->host copy, only when asked."""
        if self._lscale is None:
            return 1.0
        if grp is None:
            grp = self._last_grp
        elif isinstance(grp, str):
            grp = {"gen": L.GROUP_GEN, "dis": L.GROUP_DIS}[grp]
        live = float(self._lscale[0].item())
        if grp is None:
            return live
        own = float(self._gscale[grp].item())
        return own if own > 0 else live

    def _draw_z(self, B):
        # three draws from the CPU generator, in the reference's order (trainer.py:99-101); data-parallel ranks use
        # their own generator (seed + rank) so that shards do not share noise
        return [torch.randn(B, self.style_dim, 1, 1, generator=self._zgen) for _ in range(3)]

    def _current_lr(self, hp):
        if hp.get("lr_policy", "constant") == "step":   # StepLR (utils.py:263-271)
            return float(hp["lr"]) * float(hp["gamma"]) ** (self._sched_calls // int(hp["step_size"]))
        return float(hp["lr"])

    def _publish_losses(self, lo, hi):
        # fresh 0-d tensors per update, like the reference (trainer.py:136-165): a caller may keep them across steps
        vals = self._losses[lo:hi].clone()
        for i in range(lo, hi):
            setattr(self, L.LOSS_NAMES[i], vals[i - lo])

    def _update(self, which, x_a, x_b, hp, z):
        grp = L.GROUP_GEN if which == "gen" else L.GROUP_DIS
        x_a = x_a.to(self.device, torch.float32).contiguous()
        x_b = x_b.to(self.device, torch.float32).contiguous()
        B, Cc, H, W = x_a.shape
        if x_b.shape != x_a.shape or Cc != 3:
            raise L.AclganError("x_a / x_b must both be (B,3,H,W); got %s and %s" % (tuple(x_a.shape), tuple(x_b.shape)))
        if z is None:
            z = self._draw_z(B)
        zz = torch.stack([t.reshape(B, self.style_dim).to(torch.float32) for t in z]).to(self.device).contiguous()
        hpc = hparams_from_config(hp)
        with torch.cuda.device(self.device):
            det_now = bool(L.lib.aclgan_get_deterministic())
            if det_now != self.deterministic:      # the process-wide mode changed under us: scratch sizes depend on it
                self.deterministic = det_now
                self._ws_shape = None
                self._graphs = {}
            self._ensure_workspace(B, H, W)
            st = self._st()
            if self._gscale is not None:      # fp16: this update's gradients carry the scale that is live NOW (stream-ordered copy, no sync)
                self._gscale[grp:grp + 1].copy_(self._lscale[0:1])
                self._last_grp = grp
            fn = L.lib.aclgan_gen_update if which == "gen" else L.lib.aclgan_dis_update
            if self.hip_graph and self._reducer is None and self._run_graph(which, grp, fn, x_a, x_b, zz, B, H, W, hpc):
                pass        # zero_grad + update replayed from the captured graph
            else:
                L.check(L.lib.aclgan_zero_grad(self._ctx, grp, st), "zero_grad")   # opt.zero_grad() (trainer.py:91,248)
                if self._reducer is not None:
                    self._reducer.begin(grp)
                L.check(fn(self._ctx, L.ptr(x_a), L.ptr(x_b), L.ptr(zz), B, H, W, C.byref(hpc), L.ptr(self._losses), st), which + "_update")
            if getattr(self, "_sync_error", None) is not None:
                raise self._sync_error
            self._allreduce_grads(grp)
            o = self._opt[grp]
            o["steps"] += 1
            adam = L.Adam(self._current_lr(self._hp), o["beta1"], o["beta2"], o["eps"], o["weight_decay"])
            L.check(L.lib.aclgan_adam_step(self._ctx, grp, C.byref(adam), o["steps"], st), "adam_step")   # opt.step()
        if which == "gen":
            self._publish_losses(0, 12)
        else:
            self._publish_losses(12, 16)

    # ---- HIP-graph replay of an update (opt-in, see __init__) ----
    def _run_graph(self, which, grp, fn, x_a, x_b, zz, B, H, W, hpc):
        """True: the update ran as a graph replay.  False: run it eagerly (first call of a new key -- it also performs the
        library's one-time initialisation outside any capture -- or capture is unavailable)."""
        key = (B, H, W, bytes(hpc), self._ws.data_ptr(), self._ws.numel(), self.compute_dtype, self.deterministic)
        ent = self._graphs.get(which)
        if ent is None or ent["key"] != key:
            self._graphs[which] = {"key": key, "graph": None}
            return False
        if ent["graph"] is None:
            try:
                sx_a, sx_b, szz = torch.empty_like(x_a), torch.empty_like(x_b), torch.empty_like(zz)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    st = self._st()      # the capture stream
                    L.check(L.lib.aclgan_zero_grad(self._ctx, grp, st), "zero_grad (capture)")
                    L.check(fn(self._ctx, L.ptr(sx_a), L.ptr(sx_b), L.ptr(szz), B, H, W, C.byref(hpc), L.ptr(self._losses), st), which + "_update (capture)")
                ent.update(graph=g, x_a=sx_a, x_b=sx_b, zz=szz)
            except Exception as e:      # capture unsupported here: say so once and stay eager
                import warnings
                warnings.warn("aclgan_Trainer: HIP-graph capture failed (%r); running eagerly" % (e,))
                self.hip_graph = False
                self._graphs = {}
                return False
        ent["x_a"].copy_(x_a); ent["x_b"].copy_(x_b); ent["zz"].copy_(zz)
        ent["graph"].replay()
        return True

    # ---- data parallelism (not in the reference): one process per GPU, full replicas ----
    def _dist_world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        if dist.get_world_size() == 1 and os.environ.get("ACLGAN_BENCH_FORCE_DIST") != "1":
            return 0
        return dist.get_world_size()

    def _setup_data_parallel(self):
        """Called at the end of __init__ and resume(): replicas must START identical, whatever each rank's RNG did --
        broadcast rank 0's parameters and Adam state; then hook the engine's bucket callback so that each gradient
        bucket's all-reduce starts while the rest of the backward is still running (ddp.BucketReducer).
        z noise: every rank draws its own shard's z from a generator seeded with initial_seed() + rank."""
        self._reducer = None
        self._exposed_events, self._exposed_ms = [], 0.0
        world = self._dist_world()
        if not world:
            return
        import torch.distributed as dist
        from .ddp import BucketReducer, broadcast_flat
        for grp in (L.GROUP_GEN, L.GROUP_DIS):
            for buf in (self._param[grp], self._m[grp], self._v[grp]):
                broadcast_flat(buf)
        steps = torch.tensor([self._opt[0]["steps"], self._opt[1]["steps"], self._sched_calls], dtype=torch.int64, device=self.device)
        dist.broadcast(steps, 0)
        self._opt[0]["steps"], self._opt[1]["steps"], self._sched_calls = (int(v) for v in steps.tolist())
        self._zgen = torch.Generator().manual_seed(torch.initial_seed() + dist.get_rank())
        if os.environ.get("ACLGAN_DDP_OVERLAP", "1") != "0":
            self._reducer = BucketReducer(self._ctx, lambda g: self._grad[g], world)
        if self.hip_graph:
            # HIP-graph replay is single-GPU only: the bucket callbacks and the forward sync point are host code that starts
            # collectives from inside the update, which a captured graph cannot replay -- say so instead of silently ignoring the option
            import warnings
            warnings.warn("aclgan_Trainer: hip_graph=True is ignored in data-parallel runs (world size %d): updates run eagerly" % world)
            self.hip_graph = False
        # optional: the reference's global-batch semantics of the focus losses (config key ddp_global_focus / env
        # ACLGAN_DDP_GLOBAL_FOCUS=1): the 6 mask sums are all-reduced at a forward sync point of gen_update
        if self._hp.get("ddp_global_focus", False) or os.environ.get("ACLGAN_DDP_GLOBAL_FOCUS") == "1":
            def on_sync(user, ptr, n):
                try:
                    off = int(ptr) - self._ws.data_ptr()
                    t = self._ws[off: off + 4 * n].view(torch.float32)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)      # stream-ordered: the compute stream waits for it, the host does not
                except BaseException as e:   # noqa: BLE001  (must not unwind through the C frames)
                    self._sync_error = e
            self._sync_error = None
            self._sync_cb = L.SYNC_FN(on_sync)
            L.check(L.lib.aclgan_set_forward_sync(self._ctx, self._sync_cb, None, world), "set_forward_sync")

    def _allreduce_grads(self, grp):
        """Average the flat gradient buffer over ranks with RCCL before Adam.  With the bucket reducer the collectives
        were started from inside the backward (overlap) and are only waited for here; ACLGAN_DDP_OVERLAP=0 selects the
        plain post-backward bucketed all-reduce.  No-op when torch.distributed is not initialised."""
        world = self._dist_world()
        if not world:
            return
        # GPU-side exposure of the exchange: the compute stream does nothing between these two events except wait for the
        # collectives, so their distance is the part of the all-reduce that was NOT hidden behind the backward
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        if self._reducer is not None:
            self._reducer.finish(grp)
        else:
            from .ddp import allreduce_flat
            allreduce_flat(self._grad[grp], world)
        ev[1].record()
        self._exposed_events.append(ev)
        if len(self._exposed_events) > 4096:
            self.allreduce_exposed_ms()

    def allreduce_exposed_ms(self, reset=True):
        """total milliseconds the compute stream spent waiting for gradient collectives since the last call (synchronises)"""
        torch.cuda.synchronize(self.device)
        self._exposed_ms += sum(a.elapsed_time(b) for a, b in self._exposed_events)
        self._exposed_events = []
        v = self._exposed_ms
        if reset:
            self._exposed_ms = 0.0
        return v

    # ---- the hot path (trainer.py:90-170, 247-293) ----
    def gen_update(self, x_a, x_b, hyperparameters, z=None):
        self._update("gen", x_a, x_b, hyperparameters, z)

    def dis_update(self, x_a, x_b, hyperparameters, z=None):
        self._update("dis", x_a, x_b, hyperparameters, z)

    def update_learning_rate(self):   # trainer.py:295-299, called every iteration (train.py:101)
        self._sched_calls += 1

    def focus_translation(self, x_fg, x_bg, x_focus):
        """trainer.py:85-88 for sample() / test.py, on NCHW tensors, through the library's blend kernel.  x_fg / x_focus may
        be channel slices of one decoder output (only the batch stride has to be regular)."""
        B, Cc, H, W = x_fg.shape
        if Cc != 3 or x_focus.shape[1] != 1 or tuple(x_bg.shape) != (B, 3, H, W):
            raise L.AclganError("focus_translation: expected (B,3,H,W), (B,3,H,W), (B,1,H,W)")

        def dense(t):   # each sample's C*H*W block contiguous; any batch stride
            t = t.to(self.device, torch.float32)
            ok = t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W
            return t if ok else t.contiguous()
        fg, bg, fo = dense(x_fg), dense(x_bg), dense(x_focus)
        out = torch.empty(B, 3, H, W, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib.aclgan_focus_translation_nchw(L.ptr(fg), fg.stride(0), L.ptr(bg), bg.stride(0), L.ptr(fo), fo.stride(0),
                                                        L.ptr(out), B, H * W, self._st()), "focus_translation")
        return out

    def sample(self, x_a, x_b):
        """trainer.py:179-245: per-image eval forward; returns the reference's 9-tuple (focus branch) or 7-tuple (focus_loss == 0)."""
        x_a = x_a.to(self.device, torch.float32)
        x_b = x_b.to(self.device, torch.float32)
        if not (self.focus_lam > 0):
            return self._sample_plain(x_a, x_b)
        X_A, X_B, A_fake, B_fake, A2_fake, m_A, m_B, m_A2, m_rec, A_rec = [], [], [], [], [], [], [], [], [], []
        for i in range(x_a.size(0)):
            xa = x_a[i].unsqueeze(0)
            X_A.append(xa); X_B.append(x_b[i].unsqueeze(0))
            c_1, s_1 = self.gen_BA.encode(xa)
            img, mask = self.gen_BA.decode(c_1, self.z_1[i].unsqueeze(0)).split(3, 1)
            A_fake.append(self.focus_translation(img, xa, mask)); m_A.append(mask)
            img, mask = self.gen_BA.decode(c_1, s_1).split(3, 1)
            A_rec.append(img); m_rec.append(mask)
            c_2, _ = self.gen_AB.encode(xa)
            xb_img, mask = self.gen_AB.decode(c_2, self.z_2[i].unsqueeze(0)).split(3, 1)
            xb_img = self.focus_translation(xb_img, xa, mask)
            B_fake.append(xb_img); m_B.append(mask)
            c_3, _ = self.gen_BA.encode(xb_img)
            img, mask = self.gen_BA.decode(c_3, self.z_3[i].unsqueeze(0)).split(3, 1)
            A2_fake.append(self.focus_translation(img, xb_img, mask)); m_A2.append(mask)
        cat = torch.cat
        return (cat(X_A), cat(A_fake), cat(m_A), cat(B_fake), cat(m_B), cat(A2_fake), cat(m_A2), cat(A_rec), cat(m_rec))

    def _sample_plain(self, x_a, x_b):
        """trainer.py:216-230,238-245 (non-focus configuration).  Faithful to the reference including its quirk at :228-229: the B
        reconstruction encodes the WHOLE batch x_b inside the per-image loop, so x_B_recon comes back as B copies of the batch."""
        X_A, X_B, A_fake, B_fake, A2_fake, A_rec, B_rec = [], [], [], [], [], [], []
        for i in range(x_a.size(0)):
            xa = x_a[i].unsqueeze(0)
            X_A.append(xa); X_B.append(x_b[i].unsqueeze(0))
            c_1, s_1 = self.gen_BA.encode(xa)
            A_fake.append(self.gen_BA.decode(c_1, self.z_1[i].unsqueeze(0)))
            A_rec.append(self.gen_BA.decode(c_1, s_1))
            c_2, _ = self.gen_AB.encode(xa)
            x_B1 = self.gen_AB.decode(c_2, self.z_2[i].unsqueeze(0))
            B_fake.append(x_B1)
            c_3, _ = self.gen_BA.encode(x_B1)
            A2_fake.append(self.gen_BA.decode(c_3, self.z_3[i].unsqueeze(0)))
            c_4, s_4 = self.gen_AB.encode(x_b)
            B_rec.append(self.gen_AB.decode(c_4, s_4))
        cat = torch.cat
        return (cat(X_A), cat(A_fake), cat(B_fake), cat(A2_fake), cat(A_rec), cat(X_B), cat(B_rec))

    # ---- checkpoints (trainer.py:301-331, utils.py:211-220) ----
    def _opt_state_dict(self, grp):
        """torch.optim.Adam.state_dict() layout so that the reference can load optimizer.pt."""
        o = self._opt[grp]
        state = {}
        nets = (self.gen_AB, self.gen_BA) if grp == L.GROUP_GEN else (self.dis_A, self.dis_B, self.dis_2)
        idx = 0
        for net in nets:
            for e in net._entries:
                if o["steps"] > 0:
                    state[idx] = {"step": torch.tensor(float(o["steps"])),
                                  "exp_avg": net._view(e, self._m[grp]).contiguous().clone(),
                                  "exp_avg_sq": net._view(e, self._v[grp]).contiguous().clone()}
                idx += 1
        pg = {"lr": self._current_lr(self._hp), "betas": (o["beta1"], o["beta2"]), "eps": o["eps"], "weight_decay": o["weight_decay"],
              "amsgrad": False, "initial_lr": float(self._hp["lr"]), "params": list(range(idx))}
        return {"state": state, "param_groups": [pg]}

    def _load_opt_state_dict(self, grp, sd):
        o = self._opt[grp]
        nets = (self.gen_AB, self.gen_BA) if grp == L.GROUP_GEN else (self.dis_A, self.dis_B, self.dis_2)
        idx = 0
        steps = 0
        with torch.no_grad():
            for net in nets:
                for e in net._entries:
                    st = sd["state"].get(idx)
                    if st is not None:
                        net._view(e, self._m[grp]).copy_(st["exp_avg"].to(self.device))
                        net._view(e, self._v[grp]).copy_(st["exp_avg_sq"].to(self.device))
                        steps = int(float(st["step"]))
                    idx += 1
        o["steps"] = steps
        pg = sd["param_groups"][0]
        o.update(beta1=float(pg["betas"][0]), beta2=float(pg["betas"][1]), eps=float(pg["eps"]), weight_decay=float(pg["weight_decay"]))

    def save(self, snapshot_dir, iterations):
        gen_name = os.path.join(snapshot_dir, "gen_%08d.pt" % (iterations + 1))
        dis_name = os.path.join(snapshot_dir, "dis_%08d.pt" % (iterations + 1))
        opt_name = os.path.join(snapshot_dir, "optimizer.pt")
        cpu = lambda sd: OrderedDict((k, v.cpu()) for k, v in sd.items())  # noqa: E731
        torch.save({"AB": cpu(self.gen_AB.state_dict()), "BA": cpu(self.gen_BA.state_dict())}, gen_name)
        torch.save({"A": cpu(self.dis_A.state_dict()), "B": cpu(self.dis_B.state_dict()), "2": cpu(self.dis_2.state_dict())}, dis_name)
        opt = {"gen": self._opt_state_dict(L.GROUP_GEN), "dis": self._opt_state_dict(L.GROUP_DIS)}
        if self._lscale is not None:
            # fp16 only (an extra key the reference's loader ignores: trainer.py:313-316 reads 'gen' / 'dis'): the dynamic loss-scale state --
            # live scale, clean-update counter and the SKIPPED-update counts Adam's bias-correction step excludes.  Adam's per-tensor
            # 'step' above counts applied + skipped updates (the reference's own counter semantics: one per optimizer.step() call).
            opt["aclgan_loss_scale_state"] = self._lscale.cpu().clone()
        torch.save(opt, opt_name)

    @staticmethod
    def _get_model_list(dirname, key):   # utils.py:211-220
        if not os.path.exists(dirname):
            return None
        models = sorted(os.path.join(dirname, f) for f in os.listdir(dirname)
                        if os.path.isfile(os.path.join(dirname, f)) and key in f and ".pt" in f)
        return models[-1] if models else None

    def resume(self, checkpoint_dir, hyperparameters):
        last = self._get_model_list(checkpoint_dir, "gen")
        sd = torch.load(last, map_location="cpu")
        self.gen_AB.load_state_dict(sd["AB"]); self.gen_BA.load_state_dict(sd["BA"])
        iterations = int(last[-11:-3])
        last = self._get_model_list(checkpoint_dir, "dis")
        sd = torch.load(last, map_location="cpu")
        self.dis_A.load_state_dict(sd["A"]); self.dis_B.load_state_dict(sd["B"]); self.dis_2.load_state_dict(sd["2"])
        sd = torch.load(os.path.join(checkpoint_dir, "optimizer.pt"), map_location="cpu")
        self._load_opt_state_dict(L.GROUP_DIS, sd["dis"]); self._load_opt_state_dict(L.GROUP_GEN, sd["gen"])
        if self._lscale is not None and "aclgan_loss_scale_state" in sd:      # fp16: resume the loss scale and the skipped-update counts
            self._lscale.copy_(sd["aclgan_loss_scale_state"].to(self.device, torch.float32))
        # get_scheduler(..., iterations) (trainer.py:318-320, utils.py:263-271) builds StepLR(last_epoch=iterations).  The reference
        # pins torch 1.2.0 (acl-gan.yaml), whose _LRScheduler.__init__ ends with self.step(last_epoch): the scheduler resumes AT
        # last_epoch = iterations and (closed-form get_lr of that version) lr = lr0 * gamma ** (iterations // step_size).  torch >= 1.4
        # calls self.step() instead and would resume one epoch later (iterations + 1) -- the pinned version is the one mirrored here
        # (ACLGAN_RESUME_SCHED_PLUS1=1 selects the torch >= 1.4 behaviour).
        self._sched_calls = iterations + (1 if os.environ.get("ACLGAN_RESUME_SCHED_PLUS1") == "1" else 0)
        self._hp = hyperparameters
        self._setup_data_parallel()
        print("Resume from iteration %d" % iterations)
        return iterations
