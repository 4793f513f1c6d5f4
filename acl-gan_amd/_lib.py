"""ctypes binding of libaclgan_hip.so (C ABI: include/aclgan_hip.h).

The library is the product: there is NO CPU / PyTorch fallback.  If the shared object is missing
or fails to load, importing this module raises -- loudly -- instead of silently computing
something else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaclgan_hip.so")


class AclganError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libaclgan_hip.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C acl-gan_amd/csrc).  There is no fallback path." % LIB_PATH)

lib = C.CDLL(LIB_PATH)


class Arch(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "input_dim_a", "input_dim_b", "gen_dim", "gen_mlp_dim", "gen_style_dim", "gen_output_dim",
        "gen_n_downsample", "gen_n_res", "dis_dim", "dis_n_layer", "dis_num_scales")]


class HParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "gan_w", "gan_cw", "recon_x_w", "focus_loss", "focus_delta", "focus_upper", "focus_lower",
        "focus_epsilon", "alpha")]


class Adam(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay")]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "Hi", "Wi", "Ci", "Co", "k", "stride", "pad", "upsample", "act")]


class ImageDesc(C.Structure):
    _fields_ = [("src_offset", C.c_int64)] + [(n, C.c_int) for n in (
        "src_h", "src_w", "res_h", "res_w", "crop_y", "crop_x", "flip", "tab_x", "tab_y", "ksize_x", "ksize_y")]


ACT = {"none": 0, "relu": 1, "lrelu": 2, "tanh": 3}
NORM = {"none": 0, "in": 1, "adain": 2, "ln": 3}
DTYPE = {"fp32": 0, "bf16": 1, "fp16": 2}
GROUP_GEN, GROUP_DIS = 0, 1
NETS = {"gen_AB": 0, "gen_BA": 1, "dis_A": 2, "dis_B": 3, "dis_2": 4}
LOSS_NAMES = [
    "loss_gen_adv_A", "loss_gen_adv_B", "loss_gen_adv_2",
    "loss_gen_focus_B_size", "loss_gen_focus_B_digit", "loss_gen_focus_A_size", "loss_gen_focus_A_digit",
    "loss_gen_focus_A2_size", "loss_gen_focus_A2_digit", "loss_idt_A", "loss_idt_B", "loss_gen_total",
    "loss_dis_A", "loss_dis_B", "loss_dis_2", "loss_dis_total",
]

vp, ci, cf, i64, sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t
# aclgan_bucket_fn: void (*)(void* user, int group, int bucket, int64_t offset, int64_t numel)
BUCKET_FN = C.CFUNCTYPE(None, vp, ci, ci, i64, i64)
# aclgan_sync_fn: void (*)(void* user, float* sums_dev, int n)
SYNC_FN = C.CFUNCTYPE(None, vp, vp, ci)

# every symbol include/aclgan_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "aclgan_version": (ci, []),
    "aclgan_launch_count": (C.c_longlong, []),
    "aclgan_gemm_slices_f32": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "aclgan_winograd_filter_frag": (ci, [vp, vp, ci, ci, ci, vp]),
    "aclgan_conv3x3_winograd_fused": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "aclgan_gemm_slices_x3_scratch_bytes": (sz, [ci, ci, ci, ci]),
    "aclgan_gemm_slices_x3": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "aclgan_set_deterministic": (ci, [ci]),
    "aclgan_get_deterministic": (ci, []),
    "aclgan_last_error": (C.c_char_p, []),
    "aclgan_ctx_create": (ci, [C.POINTER(Arch), C.POINTER(vp)]),
    "aclgan_ctx_destroy": (None, [vp]),
    "aclgan_ctx_enable_capture": (ci, [vp]),
    "aclgan_warm_streams": (ci, [ci]),
    "aclgan_debug_capture_masks": (ci, [vp, vp, sz]),
    "aclgan_debug_mask_count": (ci, [vp]),
    "aclgan_debug_mask_info": (ci, [vp, ci, C.POINTER(ci), C.POINTER(C.c_longlong), C.POINTER(ci)]),
    "aclgan_group_numel": (i64, [vp, ci]),
    "aclgan_tensor_count": (ci, [vp, ci]),
    "aclgan_tensor_info": (ci, [vp, ci, ci, C.c_char_p, ci, C.POINTER(i64), C.POINTER(ci), C.POINTER(ci)]),
    "aclgan_bind_params": (ci, [vp, ci, vp, vp, vp, vp]),
    "aclgan_workspace_bytes": (ci, [vp, ci, ci, ci, C.POINTER(sz)]),
    "aclgan_bind_workspace": (ci, [vp, vp, sz]),
    "aclgan_forward_workspace_bytes": (ci, [vp, ci, ci, ci, C.POINTER(sz)]),
    "aclgan_step_algorithmic_bytes": (ci, [vp, ci, ci, ci, ci, C.POINTER(C.c_double)]),
    "aclgan_step_executed_flops": (ci, [vp, ci, ci, ci, ci, C.POINTER(C.c_double)]),
    "aclgan_gen_update": (ci, [vp, vp, vp, vp, ci, ci, ci, C.POINTER(HParams), vp, vp]),
    "aclgan_dis_update": (ci, [vp, vp, vp, vp, ci, ci, ci, C.POINTER(HParams), vp, vp]),
    "aclgan_set_grad_buckets": (ci, [vp, i64, BUCKET_FN, vp]),
    "aclgan_set_forward_sync": (ci, [vp, SYNC_FN, vp, ci]),
    "aclgan_focus_loss_global": (ci, [vp, i64, vp, i64, cf, cf, cf, cf, cf, vp, vp, vp, vp]),
    "aclgan_bucket_schedule": (ci, [vp, ci, ci, ci, ci, ci, C.POINTER(ci), ci, C.POINTER(ci)]),
    "aclgan_zero_grad": (ci, [vp, ci, vp]),
    "aclgan_adam_step": (ci, [vp, ci, C.POINTER(Adam), ci, vp]),
    "aclgan_gen_encode": (ci, [vp, ci, vp, ci, ci, ci, vp, vp, vp]),
    "aclgan_gen_decode": (ci, [vp, ci, vp, vp, ci, ci, ci, vp, vp]),
    "aclgan_dis_forward": (ci, [vp, ci, vp, ci, ci, ci, C.POINTER(vp), vp]),
    "aclgan_conv2d_fwd": (ci, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp]),
    "aclgan_conv2d_fwd_ws": (ci, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]),
    "aclgan_conv2d_fwd_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_conv2d_fwd_naive": (ci, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp]),
    "aclgan_set_compute_dtype": (ci, [vp, ci]),
    "aclgan_bind_params16": (ci, [vp, ci, vp, vp]),
    "aclgan_bind_loss_scale": (ci, [vp, vp]),
    "aclgan_conv16_eligible": (ci, [C.POINTER(ConvDesc), ci]),
    "aclgan_pack_weights16": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "aclgan_conv2d_fwd16": (ci, [C.POINTER(ConvDesc), ci, vp, vp, vp, vp, vp, vp, vp]),
    "aclgan_conv2d_fwd16_x16": (ci, [C.POINTER(ConvDesc), ci, vp, vp, vp, vp, vp, vp, vp]),
    "aclgan_conv2d_dgrad16": (ci, [C.POINTER(ConvDesc), ci, vp, vp, vp, vp, ci, vp, vp]),
    "aclgan_conv2d_wgrad16": (ci, [C.POINTER(ConvDesc), ci, vp, vp, vp, vp, vp, vp]),
    "aclgan_conv16s_ok": (ci, [C.POINTER(ConvDesc), ci]),
    "aclgan_conv2d_fwd16s": (ci, [C.POINTER(ConvDesc), ci, vp, vp, vp, vp, ci, vp]),
    "aclgan_set_tuning": (ci, [C.c_char_p, ci]),
    "aclgan_tuning": (ci, [C.c_char_p, ci, C.POINTER(ci)]),
    "aclgan_tuning_get": (ci, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "aclgan_check_workspace": (ci, [vp, ci, ci, ci]),
    "aclgan_conv2d_fwd16s_stats_chunk": (ci, [C.POINTER(ConvDesc)]),
    "aclgan_conv2d_fwd16s_stats": (ci, [C.POINTER(ConvDesc), ci, vp, vp, vp, vp, ci, vp, vp]),
    "aclgan_conv2d_dgrad16s_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_conv2d_dgrad16s": (ci, [C.POINTER(ConvDesc), ci, vp, vp, vp, ci, ci, vp, vp]),
    "aclgan_conv2d_wgrad16_st": (ci, [C.POINTER(ConvDesc), ci, vp, ci, vp, ci, vp, vp, vp, vp]),
    "aclgan_cast_storage": (ci, [vp, ci, vp, ci, i64, vp]),
    "aclgan_norm_fwd_st": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, C.POINTER(ci), vp]),
    "aclgan_norm_bwd_st": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, C.POINTER(ci), vp]),
    "aclgan_conv2d_fwd16_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_conv2d_dgrad16_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_conv2d_wgrad16_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_conv2d_dgrad": (ci, [C.POINTER(ConvDesc), vp, vp, vp, vp, ci, vp]),
    "aclgan_conv2d_dgrad_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_conv2d_wgrad": (ci, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp]),
    "aclgan_conv2d_wgrad_ws": (ci, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]),
    "aclgan_conv2d_wgrad_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_norm_fwd": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp]),
    "aclgan_conv2d_block_fwd": (ci, [C.POINTER(ConvDesc), ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int), vp]),
    "aclgan_conv2d_block_fwd_scratch_bytes": (sz, [C.POINTER(ConvDesc)]),
    "aclgan_norm_bwd": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp]),
    "aclgan_norm_scratch_bytes": (sz, [ci, ci, ci]),
    "aclgan_avgpool3s2_fwd": (ci, [ci, ci, ci, ci, vp, vp, vp]),
    "aclgan_avgpool3s2_bwd": (ci, [ci, ci, ci, ci, vp, vp, ci, vp]),
    "aclgan_adam_flat": (ci, [vp, vp, vp, vp, i64, C.POINTER(Adam), ci, vp]),
    "aclgan_linear_fwd": (ci, [ci, ci, ci, vp, vp, vp, ci, vp, vp]),
    "aclgan_mlp3_fwd": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "aclgan_linear_bwd": (ci, [ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp]),
    "aclgan_gap_fwd": (ci, [ci, ci, ci, vp, vp, vp]),
    "aclgan_gap_bwd": (ci, [ci, ci, ci, vp, vp, ci, vp]),
    "aclgan_focus_blend_fwd": (ci, [ci, ci, vp, vp, vp, vp, vp, vp]),
    "aclgan_focus_blend_bwd": (ci, [ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]),
    "aclgan_focus_translation_nchw": (ci, [vp, i64, vp, i64, vp, i64, vp, ci, ci, vp]),
    "aclgan_lsgan_loss": (ci, [vp, ci, cf, cf, vp, vp, cf, vp]),
    "aclgan_lsgan_loss_multi": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, vp]),
    "aclgan_l1_loss": (ci, [vp, ci, vp, i64, vp, vp, cf, ci, vp]),
    "aclgan_focus_loss_scratch_bytes": (sz, [i64]),
    "aclgan_focus_loss": (ci, [vp, i64, cf, cf, cf, cf, cf, vp, vp, vp, vp, vp]),
    "aclgan_nchw_to_nhwc": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "aclgan_nhwc_to_nchw": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "aclgan_image_resample_ksize": (ci, [ci, ci]),
    "aclgan_image_resample_coeffs": (ci, [ci, ci, C.POINTER(ci), C.POINTER(ci)]),
    "aclgan_image_batch_transform": (ci, [vp, C.POINTER(ImageDesc), vp, ci, vp, vp, ci, ci, vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the .so does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return lib.aclgan_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    if rc != 0:
        raise AclganError("%s failed (code %d): %s" % (what or "libaclgan_hip call", rc, last_error()))


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """the current HIP stream of `device` (default: the current device) as a void*"""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
