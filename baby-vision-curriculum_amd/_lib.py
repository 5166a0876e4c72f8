"""ctypes binding of libbvc_hip.so (C ABI declared in include/bvc.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BVC_LIB_PATH: another build of the same library (tools/: A/B of two builds on one box); the default is the in-tree build
LIB_PATH = os.environ.get("BVC_LIB_PATH") or os.path.join(_HERE, "libbvc_hip.so")

c_void_p, c_int, c_int64, c_float, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_char_p
c_double = ctypes.c_double


class BvcError(RuntimeError):
    pass


class VideoMAEConfigC(ctypes.Structure):
    _fields_ = [(n, c_int) for n in (
        "image_size", "patch_size", "num_channels", "num_frames", "tubelet_size",
        "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
        "decoder_hidden_size", "decoder_num_hidden_layers", "decoder_num_attention_heads",
        "decoder_intermediate_size")] + [("layer_norm_eps", c_float), ("decoder_norm_eps", c_float),
                                         ("norm_pix_loss", c_int)]


class VitConfigC(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("image_size", "patch_size", "num_channels", "num_frames", "tubelet_size", "embed_dim",
                                     "depth", "num_heads", "mlp_hidden")] + [("eps", c_float)]


class PredictorConfigC(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("seq_len", "embed_dim", "pred_dim", "depth", "num_heads", "mlp_hidden")] + [("eps", c_float)]


class PixelFormatC(ctypes.Structure):
    _fields_ = [("dtype", c_int), ("mean", c_float * 4), ("std", c_float * 4)]


def pixel_format(pixel_values, mean, std, channels):
    """None for f32 input (already normalised); a PixelFormatC for uint8 frames to be normalised on the GPU."""
    import torch
    if pixel_values.dtype != torch.uint8:
        return None
    def per_channel(v):
        v = list(v) if hasattr(v, "__len__") else [float(v)] * channels
        if len(v) != channels or channels > 4:
            raise ValueError("pixel mean / std need one value per channel (at most 4 channels)")
        return v + [0.0] * (4 - channels)
    f = PixelFormatC()
    f.dtype = 1
    f.mean = (c_float * 4)(*per_channel(mean))
    f.std = (c_float * 4)(*[x if i < channels else 1.0 for i, x in enumerate(per_channel(std))])
    return f


OPT_MAX_GROUPS = 8     # BVC_OPT_MAX_GROUPS (include/bvc.h)


class SgdGroupsC(ctypes.Structure):
    _fields_ = [("ngroups", c_int)] + [(n, c_float * OPT_MAX_GROUPS) for n in ("lr", "momentum", "dampening", "weight_decay")] + \
               [(n, c_int * OPT_MAX_GROUPS) for n in ("nesterov", "first_step", "maximize")]


class AdamGroupsC(ctypes.Structure):
    _fields_ = [("ngroups", c_int)] + [(n, ctypes.c_double * OPT_MAX_GROUPS) for n in ("lr", "beta1", "beta2", "eps", "weight_decay")] + \
               [(n, c_int * OPT_MAX_GROUPS) for n in ("decoupled", "maximize")]


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int),
        ("a_bytes", ctypes.c_uint32), ("b_bytes", ctypes.c_uint32),
        ("alpha", c_float), ("alpha_dev", c_void_p),
        ("epi", c_int), ("split_k", c_int),
        ("C", c_void_p), ("ldc", c_int), ("C2", c_void_p),
        ("bias", c_void_p), ("resid", c_void_p), ("aux", c_void_p), ("ldaux", c_int),
        ("rowtok", c_void_p), ("pos", c_void_p), ("labels", c_void_p), ("partial", c_void_p),
        ("rin", c_int), ("rout", c_int),
        ("rowsum", c_void_p),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("ln_eps", c_float),
        ("ln_x", c_void_p), ("ln_part", c_void_p), ("ln_dgamma", c_void_p), ("ln_dbeta", c_void_p),
    ]


BUCKET_FN = ctypes.CFUNCTYPE(None, c_int64, c_int64, c_void_p)

# every symbol include/bvc.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "bvc_last_error": (c_char_p, []),
    "bvc_version": (c_char_p, []),
    "bvc_set_option": (c_int, [c_char_p, c_int]),
    "bvc_get_option": (c_int, [c_char_p]),
    "bvc_videomae_param_count": (c_int, [ctypes.POINTER(VideoMAEConfigC)]),
    "bvc_videomae_param_numel": (c_int64, [ctypes.POINTER(VideoMAEConfigC)]),
    "bvc_videomae_param_info": (c_int, [ctypes.POINTER(VideoMAEConfigC), c_int, ctypes.c_char_p, c_int,
                                        ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), ctypes.POINTER(c_int),
                                        ctypes.POINTER(c_int64)]),
    "bvc_videomae_create": (c_int, [ctypes.POINTER(VideoMAEConfigC), c_int, c_int, ctypes.POINTER(c_void_p)]),
    "bvc_videomae_destroy": (None, [c_void_p]),
    "bvc_videomae_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bvc_videomae_forward_px": (c_int, [c_void_p, c_void_p, ctypes.POINTER(PixelFormatC), c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "bvc_videomae_backward": (c_int, [c_void_p, c_void_p, c_void_p, BUCKET_FN, c_void_p, c_void_p]),
    "bvc_videomae_tap": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, ctypes.POINTER(c_int64), c_void_p]),
    "bvc_videomae_encoder_param_numel": (c_int64, [ctypes.POINTER(VideoMAEConfigC)]),
    "bvc_videomae_encoder_create": (c_int, [ctypes.POINTER(VideoMAEConfigC), c_int, ctypes.POINTER(c_void_p)]),
    "bvc_videomae_encoder_destroy": (None, [c_void_p]),
    "bvc_videomae_encode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "bvc_videomae_encode_px": (c_int, [c_void_p, c_void_p, ctypes.POINTER(PixelFormatC), c_int, c_void_p, c_void_p, c_void_p, c_float,
                                       c_void_p, c_void_p, c_void_p]),
    "bvc_vit_param_count": (c_int, [ctypes.POINTER(VitConfigC)]),
    "bvc_vit_param_numel": (c_int64, [ctypes.POINTER(VitConfigC)]),
    "bvc_vit_param_info": (c_int, [ctypes.POINTER(VitConfigC), c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_int64),
                                   ctypes.POINTER(c_int64), ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]),
    "bvc_vit_create": (c_int, [ctypes.POINTER(VitConfigC), c_int, ctypes.POINTER(c_void_p)]),
    "bvc_vit_destroy": (None, [c_void_p]),
    "bvc_vit_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "bvc_vit_forward_px": (c_int, [c_void_p, c_void_p, ctypes.POINTER(PixelFormatC), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "bvc_vit_backward": (c_int, [c_void_p, c_void_p, c_void_p, BUCKET_FN, c_void_p, c_void_p]),
    "bvc_predictor_param_count": (c_int, [ctypes.POINTER(PredictorConfigC)]),
    "bvc_predictor_param_numel": (c_int64, [ctypes.POINTER(PredictorConfigC)]),
    "bvc_predictor_param_info": (c_int, [ctypes.POINTER(PredictorConfigC), c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_int64),
                                         ctypes.POINTER(c_int64), ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]),
    "bvc_predictor_create": (c_int, [ctypes.POINTER(PredictorConfigC), c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "bvc_predictor_destroy": (None, [c_void_p]),
    "bvc_predictor_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "bvc_predictor_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bvc_predictor_backward_cb": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, BUCKET_FN, c_void_p, c_void_p]),
    "bvc_op_target_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "bvc_op_smooth_l1_workspace": (c_int, [c_int64]),
    "bvc_op_smooth_l1_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "bvc_op_smooth_l1_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "bvc_op_ema": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "bvc_op_token_mean": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bvc_op_token_mean_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bvc_op_gemm": (c_int, [ctypes.POINTER(GemmDesc), c_int, c_int, c_int, c_int, c_void_p]),
    "bvc_op_gemm_num_tiles": (c_int, [ctypes.POINTER(GemmDesc), c_int]),
    "bvc_op_gemm_kernel": (c_int, [ctypes.POINTER(GemmDesc), c_int, c_int, c_int, c_int, ctypes.c_char_p, c_int]),
    "bvc_op_gemm_plan_dw": (c_int, [ctypes.POINTER(GemmDesc), c_int]),
    "bvc_op_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bvc_op_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bvc_op_attention_bwd_part": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bvc_op_layernorm_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_float, c_void_p]),
    "bvc_op_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bvc_op_layernorm_bwd_workspace": (c_int64, [c_int, c_int]),
    "bvc_op_colsum_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "bvc_op_cast_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "bvc_op_row_normalize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "bvc_op_row_normalize_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bvc_op_nce_finalize": (c_int, [c_void_p, c_int, c_float, c_int64, c_void_p, c_void_p, c_void_p]),
    "bvc_op_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "bvc_videomae_shadow": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64)]),
    "bvc_vit_shadow": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64)]),
    "bvc_predictor_shadow": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64)]),
    "bvc_op_adam_prepare": (c_int, [c_void_p, c_double, c_double, c_double, c_void_p, c_void_p]),
    "bvc_op_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double, c_double, c_double,
                                 c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "bvc_op_row_ln_selected": (c_int, [c_int, c_int, c_int, c_int]),
    "bvc_op_sgd_step_segments": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_void_p]),
    "bvc_op_adam_step_segments": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "bvc_op_nonfinite_check": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "bvc_op_mask_index": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bvc_op_gather_patches": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bvc_op_pixel_labels": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bvc_comm_unique_id": (c_int, [c_void_p]),
    "bvc_comm_init": (c_int, [c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "bvc_comm_destroy": (c_int, [c_void_p]),
    "bvc_comm_stream": (c_void_p, [c_void_p]),
    "bvc_comm_rank": (c_int, [c_void_p]),
    "bvc_comm_world": (c_int, [c_void_p]),
    "bvc_comm_library": (c_char_p, []),
    "bvc_allreduce_bucket": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "bvc_comm_wait": (c_int, [c_void_p, c_void_p]),
    "bvc_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "bvc_allreduce": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "bvc_broadcast": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
}

_lib = None


def lib():
    """Load libbvc_hip.so once; raise loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BvcError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().bvc_last_error()
        raise BvcError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def set_option(name, value):
    """bvc_set_option (include/bvc.h): "gemm8" -1 / 0 / 1, "dw_overlap" 0 / 1, "row_ln" -1 / 0 / 1.  Returns the previous value."""
    old = lib().bvc_get_option(name.encode())
    check(lib().bvc_set_option(name.encode(), int(value)), "bvc_set_option")
    return old


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
