"""JEPA encoder / predictor with the reference's interface (pretraining/predictive/vision_transformer.py,
pretrain_jepa.py:84-124, tensors.py:53-71); the arithmetic runs in libbvc_hip.so.

Reference seam (pretrain_jepa.py):
  :84-124  get_model -> vit.__dict__[model_name](...), vit_predictor(sequence_shape, embed_dim, ..., num_heads=encoder.num_heads)
  :258     target_encoder = copy.deepcopy(encoder)
  :386-392 h = target_encoder(imgs); h = F.layer_norm(h, (D,)); h = apply_masks(h, masks_pred); repeat_interleave_batch
  :395-396 z = encoder(imgs, masks_enc); z = predictor(z, masks_enc, masks_pred)
  :400     F.smooth_l1_loss(z, h)
  :431-432 param_k.data.mul_(m).add_((1.-m) * param_q.detach().data)
State-dict keys equal the reference's (pos_embed, patch_embed.proj.*, blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,
mlp.fc2}.*, norm.*; mask_token, predictor_pos_embed, predictor_embed.*, predictor_blocks.N.*, predictor_norm.*, predictor_proj.*).
"""
from __future__ import annotations

import ctypes
import math
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .flat import FlatParamModule, query_layout


# ----------------------------------------------------------------------------- small helpers with the reference's names
def apply_masks(x, masks):
    """tensors.py:53-62 - keep the tokens listed in each mask, concatenated over masks on the batch dim."""
    all_x = []
    for m in masks:
        mask_keep = m.unsqueeze(-1).repeat(1, 1, x.size(-1))
        all_x += [torch.gather(x, dim=1, index=mask_keep)]
    return torch.cat(all_x, dim=0)


def repeat_interleave_batch(x, B, repeat):
    """tensors.py:65-71"""
    N = len(x) // B
    return torch.cat([torch.cat([x[i * B:(i + 1) * B] for _ in range(repeat)], dim=0) for i in range(N)], dim=0)


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def positional_encoding_3d(sequence_shape, channels_out):
    """PositionalEncoding3D (vision_transformer.py:29-78) evaluated on a (T, H, W) grid -> (1, T*H*W, channels_out).
    Per axis ch = 2*ceil(C/6) channels (made even), inv_freq = 10000^(-arange(0,ch,2)/ch), sin/cos interleaved; the three
    axis embeddings are concatenated [time, row, col] and truncated to C."""
    ch = int(np.ceil(channels_out / 6) * 2)
    if ch % 2:
        ch += 1
    inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))

    def emb(n):
        s = torch.einsum("i,j->ij", torch.arange(n).float(), inv_freq)
        return torch.flatten(torch.stack((s.sin(), s.cos()), dim=-1), -2, -1)

    x, y, z = sequence_shape
    e = torch.zeros((x, y, z, ch * 3))
    e[..., :ch] = emb(x)[:, None, None, :]
    e[..., ch:2 * ch] = emb(y)[None, :, None, :]
    e[..., 2 * ch:] = emb(z)[None, None, :, :]
    return e[..., :channels_out].reshape(1, -1, channels_out).float()


def _as_idx(mask, device):
    return mask.to(device=device, dtype=torch.int32).contiguous()


def _init_like_reference(name, shape, init_std, fan_in_conv, layer_scale):
    """VisionTransformer.__init__: trunc_normal(std) Linear weights, zero biases, LayerNorm 1/0, then fix_init_weight
    divides attn.proj / mlp.fc2 weights by sqrt(2 * layer_id) (:357-376).  The Conv3d keeps torch's default init."""
    if name.endswith("patch_embed.proj.weight"):
        bound = 1.0 / math.sqrt(fan_in_conv)
        return torch.empty(shape).uniform_(-bound, bound)
    if name.endswith("patch_embed.proj.bias"):
        bound = 1.0 / math.sqrt(fan_in_conv)
        return torch.empty(shape).uniform_(-bound, bound)
    if len(shape) == 2:
        w = trunc_normal_(torch.empty(shape), std=init_std)
        if layer_scale and (name.endswith("attn.proj.weight") or name.endswith("mlp.fc2.weight")):
            w.div_(math.sqrt(2.0 * (int(name.split(".")[1]) + 1)))
        return w
    if name.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
        return torch.ones(shape)
    return torch.zeros(shape)


# ----------------------------------------------------------------------------- encoder
class _PatchEmbedInfo:
    def __init__(self, num_patches, patch_size, tubelet_size, img_size):
        self.num_patches, self.patch_size, self.tubelet_size, self.img_size = num_patches, patch_size, tubelet_size, img_size


class _EncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, imgs, idx):
        ctx.model = model
        out = model._run_forward(imgs, idx)
        ctx.stamp = model._stamp_forward()
        return out

    @staticmethod
    def backward(ctx, dout):
        ctx.model._check_generation(ctx.stamp)
        ctx.model._run_backward(dout)
        return None, None, None, None


class VisionTransformer(FlatParamModule):
    _shadow_fn = "bvc_vit_shadow"
    """Vision Transformer (vision_transformer.py:293-418); forward(x, masks=None) with x (B, T, C, H, W)."""

    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_frames=1, tubelet_size=1, embed_dim=768,
                 predictor_embed_dim=384, depth=12, predictor_depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=None, init_std=0.02,
                 norm_eps=1e-6, **kwargs):
        super().__init__()
        if not qkv_bias or qk_scale is not None or drop_rate or attn_drop_rate or drop_path_rate:
            raise ValueError("only qkv_bias=True, default scale and zero dropout / drop-path are implemented (the reference's settings)")
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.sequence_shape = (num_frames // tubelet_size, img_size[0] // patch_size, img_size[0] // patch_size)
        self.in_chans, self.num_frames = in_chans, num_frames
        self._pe_info = _PatchEmbedInfo(int(np.prod(self.sequence_shape)), patch_size, tubelet_size, img_size[0])
        self.init_std = init_std
        self._cfg = _lib.VitConfigC(img_size[0], patch_size, in_chans, num_frames, tubelet_size, embed_dim, depth, num_heads,
                                    int(embed_dim * mlp_ratio), float(norm_eps))
        L = _lib.lib()
        layout, numel = query_layout(L.bvc_vit_param_count, L.bvc_vit_param_numel, L.bvc_vit_param_info, self._cfg)
        pos = positional_encoding_3d(self.sequence_shape, embed_dim)
        fan_in = in_chans * tubelet_size * patch_size * patch_size

        def init(name, shape):
            if name == "pos_embed":
                return pos.clone()
            return _init_like_reference(name, shape, init_std, fan_in, layer_scale=True)

        self._init_flat(layout, numel, init, frozen=("pos_embed",))
        self._modules["patch_embed"].num_patches = self._pe_info.num_patches
        self._ctx, self._ctx_key = None, None
        self.pixel_mean, self.pixel_std = 0.5, 0.25      # normalisation applied on the GPU when the input is uint8 frames

    @property
    def num_patches(self):
        return self._pe_info.num_patches

    def _get_ctx(self, batch):
        dev = self._flat.device.index
        if self._ctx is not None and self._ctx_key[1] == dev and self._ctx_key[0] >= batch:
            return self._ctx
        self._free_ctx()
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().bvc_vit_create(ctypes.byref(self._cfg), batch, ctypes.byref(h)), "bvc_vit_create")
        self._ctx, self._ctx_key = h, (batch, dev)
        return h

    def _free_ctx(self):
        if getattr(self, "_ctx", None) is not None:
            _lib.lib().bvc_vit_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._free_ctx()
        except Exception:
            pass

    def _run_forward(self, imgs, idx):
        B = imgs.shape[0]
        N = idx.shape[1] if idx is not None else self.num_patches
        h = self._get_ctx(B)
        out = torch.empty((B, N, self.embed_dim), dtype=torch.float32, device=imgs.device)
        fmt = _lib.pixel_format(imgs, self.pixel_mean, self.pixel_std, self.in_chans)
        self._shadow_vouch(h)
        _lib.check(_lib.lib().bvc_vit_forward_px(h, imgs.data_ptr(), ctypes.byref(fmt) if fmt is not None else None,
                                                 idx.data_ptr() if idx is not None else None, B, N, self._flat.data_ptr(),
                                                 out.data_ptr(), _lib.current_stream_ptr()), "bvc_vit_forward")
        self._shadow_established(h)
        self._live = (imgs, idx)
        return out

    def _run_backward(self, dout):
        target, accumulate = self._grad_target()
        d = dout.detach().to(torch.float32).contiguous()
        cb = self._bucket_callback(accumulate)
        self._library_backward("bvc_vit_backward",
                               _lib.lib().bvc_vit_backward(self._ctx, d.data_ptr(), target.data_ptr(), cb, None, _lib.current_stream_ptr()))
        self._publish_grads(target, accumulate)
        self._live = None

    def forward(self, x, masks=None):
        if not x.is_cuda:
            raise _lib.BvcError("VisionTransformer runs on a GPU only (libbvc_hip.so has no CPU path)")
        B, T, C, H, W = x.shape
        if (T, C, H, W) != (self.num_frames, self.in_chans, self._pe_info.img_size, self._pe_info.img_size):
            raise ValueError("input shape does not match the model (B, T, C, H, W)")
        idx = None
        if masks is not None:
            if not isinstance(masks, list):
                masks = [masks]
            if len(masks) != 1:
                raise NotImplementedError("one context mask per call (the reference's collator uses nenc=1)")
            idx = _as_idx(masks[0], x.device)
        self._ensure_flat(x.device)
        imgs = x.detach()
        imgs = (imgs if imgs.dtype == torch.uint8 else imgs.to(torch.float32)).contiguous()    # uint8: normalised on the GPU
        anchor = self._param("norm.weight")
        if torch.is_grad_enabled() and anchor.requires_grad:
            return _EncFn.apply(anchor, self, imgs, idx)
        out = self._run_forward(imgs, idx)
        self._stamp_forward()         # a pending backward of an earlier forward must not run on these activations
        return out


# ----------------------------------------------------------------------------- predictor
class _PredFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, model, idx_ctx, idx_pred, anchor):
        ctx.model = model
        ctx.z_dtype = z.dtype
        out = model._run_forward(z, idx_ctx, idx_pred)
        ctx.stamp = model._stamp_forward()
        return out

    @staticmethod
    def backward(ctx, dout):
        ctx.model._check_generation(ctx.stamp)
        dz = ctx.model._run_backward(dout)
        return dz.to(ctx.z_dtype), None, None, None, None


class VisionTransformerPredictor(FlatParamModule):
    _shadow_fn = "bvc_predictor_shadow"
    """vision_transformer.py:421-535; forward(x, masks_x, masks) -> (len(masks) * B, N_pred, embed_dim)."""

    def __init__(self, sequence_shape, embed_dim=768, predictor_embed_dim=384, depth=6, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=None,
                 init_std=0.02, norm_eps=1e-6, **kwargs):
        super().__init__()
        if not qkv_bias or qk_scale is not None or drop_rate or attn_drop_rate or drop_path_rate:
            raise ValueError("only qkv_bias=True, default scale and zero dropout / drop-path are implemented")
        self.embed_dim, self.predictor_embed_dim, self.num_heads = embed_dim, predictor_embed_dim, num_heads
        self.sequence_shape = tuple(sequence_shape)
        self.num_patches = int(np.prod(sequence_shape))
        self.init_std = init_std
        self._cfg = _lib.PredictorConfigC(self.num_patches, embed_dim, predictor_embed_dim, depth, num_heads,
                                          int(predictor_embed_dim * mlp_ratio), float(norm_eps))
        L = _lib.lib()
        layout, numel = query_layout(L.bvc_predictor_param_count, L.bvc_predictor_param_numel, L.bvc_predictor_param_info, self._cfg)
        pos = positional_encoding_3d(self.sequence_shape, predictor_embed_dim)

        def init(name, shape):
            if name == "predictor_pos_embed":
                return pos.clone()
            if name == "mask_token":
                return trunc_normal_(torch.zeros(shape), std=init_std)
            n2 = name.replace("predictor_blocks.", "blocks.").replace("predictor_norm.", "norm.")
            return _init_like_reference(n2, shape, init_std, 1, layer_scale=True)

        self._init_flat(layout, numel, init, frozen=("predictor_pos_embed",))
        self._ctx, self._ctx_key = None, None

    def _get_ctx(self, B, nsets, tokens):
        dev = self._flat.device.index
        k = self._ctx_key
        if self._ctx is not None and k[3] == dev and k[0] >= B and k[1] >= nsets and k[2] >= tokens:
            return self._ctx
        self._free_ctx()
        h = ctypes.c_void_p()
        tokens = max(tokens, min(2 * self.num_patches, tokens + 64))     # head-room: token counts vary from step to step
        _lib.check(_lib.lib().bvc_predictor_create(ctypes.byref(self._cfg), B, nsets, tokens, ctypes.byref(h)), "bvc_predictor_create")
        self._ctx, self._ctx_key = h, (B, nsets, tokens, dev)
        return h

    def _free_ctx(self):
        if getattr(self, "_ctx", None) is not None:
            _lib.lib().bvc_predictor_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._free_ctx()
        except Exception:
            pass

    def _run_forward(self, z, idx_ctx, idx_pred):
        B, Nc = idx_ctx.shape
        nsets, _, Np = idx_pred.shape
        h = self._get_ctx(B, nsets, Nc + Np)
        zf = z.detach().to(torch.float32).contiguous()
        out = torch.empty((nsets * B, Np, self.embed_dim), dtype=torch.float32, device=z.device)
        self._shadow_vouch(h)
        _lib.check(_lib.lib().bvc_predictor_forward(h, zf.data_ptr(), idx_ctx.data_ptr(), idx_pred.data_ptr(), B, Nc, nsets, Np,
                                                    self._flat.data_ptr(), out.data_ptr(), _lib.current_stream_ptr()),
                   "bvc_predictor_forward")
        self._shadow_established(h)
        self._live = (zf, idx_ctx, idx_pred)
        return out

    def _run_backward(self, dout):
        zf = self._live[0]
        target, accumulate = self._grad_target()
        d = dout.detach().to(torch.float32).contiguous()
        dz = torch.empty_like(zf)
        cb = self._bucket_callback(accumulate)      # per-block gradient ranges, tail first, for the data-parallel wrapper
        self._library_backward("bvc_predictor_backward",
                               _lib.lib().bvc_predictor_backward_cb(self._ctx, d.data_ptr(), target.data_ptr(), dz.data_ptr(), cb, None,
                                                                    _lib.current_stream_ptr()))
        self._publish_grads(target, accumulate)
        self._live = None
        return dz

    def forward(self, x, masks_x, masks):
        assert (masks is not None) and (masks_x is not None), 'Cannot run predictor without mask indices'
        if not x.is_cuda:
            raise _lib.BvcError("VisionTransformerPredictor runs on a GPU only (libbvc_hip.so has no CPU path)")
        if not isinstance(masks_x, list):
            masks_x = [masks_x]
        if not isinstance(masks, list):
            masks = [masks]
        if len(masks_x) != 1:
            raise NotImplementedError("one context mask per call (the reference's collator uses nenc=1)")
        self._ensure_flat(x.device)
        idx_ctx = _as_idx(masks_x[0], x.device)
        idx_pred = torch.stack([_as_idx(m, x.device) for m in masks], dim=0).contiguous()
        if x.shape[0] != idx_ctx.shape[0] or x.shape[1] != idx_ctx.shape[1]:
            raise ValueError("context tokens and context mask disagree")
        anchor = self._param("predictor_norm.weight")
        if torch.is_grad_enabled() and (x.requires_grad or anchor.requires_grad):
            return _PredFn.apply(x, self, idx_ctx, idx_pred, anchor)
        out = self._run_forward(x, idx_ctx, idx_pred)
        self._stamp_forward()
        return out


# ----------------------------------------------------------------------------- factories (vision_transformer.py:538-590)
def vit_predictor(**kwargs):
    return VisionTransformerPredictor(mlp_ratio=4, qkv_bias=True, norm_eps=1e-6, **kwargs)


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True, norm_eps=1e-6, **kwargs)


def vit_base(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, norm_eps=1e-6, **kwargs)


def vit_large(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True, norm_eps=1e-6, **kwargs)


VIT_EMBED_DIMS = {'vit_small': 384, 'vit_base': 768, 'vit_large': 1024}
_FACTORIES = {'vit_small': vit_small, 'vit_base': vit_base, 'vit_large': vit_large, 'vit_predictor': vit_predictor}


def get_model(device, patch_size=16, tubelet_size=1, num_frames=1, model_name='vit_base', image_size=224, pred_depth=6,
              pred_emb_dim=384):
    """pretrain_jepa.py:84-124: builds both modules, then re-draws every Linear with trunc_normal(std=0.02) (which also
    overwrites fix_init_weight's depth rescale) and resets LayerNorms."""
    encoder = _FACTORIES[model_name](img_size=[image_size], patch_size=patch_size, num_frames=num_frames, tubelet_size=tubelet_size)
    predictor = vit_predictor(sequence_shape=encoder.sequence_shape, embed_dim=encoder.embed_dim,
                              predictor_embed_dim=pred_emb_dim, depth=pred_depth, num_heads=encoder.num_heads)
    for mod in (encoder, predictor):
        for name, p in mod.named_parameters():
            if name.startswith("patch_embed."):
                continue                      # init_weights only matches nn.Linear / nn.LayerNorm
            if p.dim() == 2:
                trunc_normal_(p.data, std=0.02)
            elif p.dim() == 1 and name.endswith("bias"):
                p.data.zero_()
            elif p.dim() == 1 and "norm" in name:
                p.data.fill_(1.0)
    encoder.to(device)
    predictor.to(device)
    return encoder, predictor


def _decay_split(module):
    """(weights, no-decay tensors) of a module by the reference's rule (helper.py:124-140): a tensor is excluded from weight decay
    when its name contains 'bias' or it is 1-D (LayerNorm scales; the frozen pos_embed and the mask token are 3-D and stay)."""
    decay, plain = [], []
    for n, p in module.named_parameters():
        (plain if ("bias" in n or p.dim() == 1) else decay).append(p)
    return decay, plain


def init_opt(encoder, predictor, iterations_per_epoch, start_lr, ref_lr, momentum, warmup, num_epochs, wd=1e-6, final_wd=1e-6,
             final_lr=0.0, use_bfloat16=False, ipe_scale=1.25):
    """pretraining/predictive/helper.py:108-165 (called at pretrain_jepa.py:274): FOUR parameter groups - encoder weights,
    predictor weights, then the encoder's and the predictor's biases / 1-D tensors with ``weight_decay`` 0 and the ``WD_exclude``
    mark - under SGD(lr=ref_lr, weight_decay=wd, momentum, nesterov=True); the reference leaves both schedulers None and builds a
    GradScaler when ``use_bfloat16``.  Here the optimiser is the fused one (two launches per step: one per flat buffer, whatever
    the groups) and the scaler is the one whose inf check is a single read per flat gradient buffer."""
    from . import amp, optim
    enc_w, enc_b = _decay_split(encoder)
    pred_w, pred_b = _decay_split(predictor)
    groups = [{"params": enc_w}, {"params": pred_w},
              {"params": enc_b, "WD_exclude": True, "weight_decay": 0},
              {"params": pred_b, "WD_exclude": True, "weight_decay": 0}]
    optimizer = optim.SGD(groups, lr=ref_lr, weight_decay=wd, momentum=momentum, nesterov=True)
    scaler = amp.GradScaler("cuda") if use_bfloat16 else None
    return optimizer, scaler, None, None


# ----------------------------------------------------------------------------- fused pieces of train_step
def select_targets(h, masks_pred, eps=1e-5):
    """forward_target's tail (pretrain_jepa.py:387-392) in one kernel: F.layer_norm(h, (D,)) without affine, the rows the
    prediction masks keep, ordered mask-major then sample (= apply_masks + repeat_interleave_batch with one context mask)."""
    B, L, D = h.shape
    idx = torch.stack([_as_idx(m, h.device) for m in masks_pred], dim=0).contiguous()
    nsets, _, Np = idx.shape
    hf = h.detach().to(torch.float32).contiguous()
    out = torch.empty((nsets * B, Np, D), dtype=torch.float32, device=h.device)
    _lib.check(_lib.lib().bvc_op_target_select(hf.data_ptr(), idx.data_ptr(), out.data_ptr(), nsets, B, Np, L, D, float(eps),
                                               _lib.current_stream_ptr()), "bvc_op_target_select")
    return out


_TARGET_STREAMS = {}


def forward_target_async(target_encoder, imgs, masks_pred, eps=1e-5):
    """forward_target of pretrain_jepa.py:384-392 (target encoder under no_grad, F.layer_norm, the prediction masks' rows), enqueued on a
    second HIP stream so that it runs BESIDE the context encoder and the predictor: nothing consumes the targets before the loss, and at the
    reference's 16 samples per GPU neither branch fills 256 CUs alone (ViT-B step -9 %, ViT-L -5 %, profiles/r05_z_jepa_overlap_*.txt).
    Same kernels, same results.  Returns join(): call it where the targets are needed - it makes the current stream wait for the side
    stream and returns h.  The side stream first waits for the current one (the EMA update of the last step wrote these parameters there)."""
    dev = imgs.device
    main = torch.cuda.current_stream(dev)
    side = _TARGET_STREAMS.get(dev.index)
    if side is None:
        side = _TARGET_STREAMS[dev.index] = torch.cuda.Stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side), torch.no_grad():
        h = select_targets(target_encoder(imgs), masks_pred, eps)

    def join():
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(side)
        h.record_stream(cur)
        return h
    return join


class _SmoothL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, h):
        zf, hf = z.detach().to(torch.float32).contiguous(), h.detach().to(torch.float32).contiguous()
        n = zf.numel()
        L = _lib.lib()
        ws = torch.empty(L.bvc_op_smooth_l1_workspace(n), dtype=torch.float32, device=z.device)
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        _lib.check(L.bvc_op_smooth_l1_fwd(zf.data_ptr(), hf.data_ptr(), n, ws.data_ptr(), loss.data_ptr(), _lib.current_stream_ptr()),
                   "bvc_op_smooth_l1_fwd")
        ctx.save_for_backward(zf, hf)
        ctx.z_dtype = z.dtype
        return loss

    @staticmethod
    def backward(ctx, gout):
        zf, hf = ctx.saved_tensors
        g = gout.detach().to(torch.float32).contiguous()
        dz = torch.empty_like(zf)
        _lib.check(_lib.lib().bvc_op_smooth_l1_bwd(zf.data_ptr(), hf.data_ptr(), g.data_ptr(), zf.numel(), dz.data_ptr(),
                                                   _lib.current_stream_ptr()), "bvc_op_smooth_l1_bwd")
        return dz.to(ctx.z_dtype), None


def smooth_l1_loss(z, h):
    """F.smooth_l1_loss(z, h) (beta = 1, mean) as used at pretrain_jepa.py:400."""
    if z.shape != h.shape:
        raise ValueError("smooth_l1_loss: shape mismatch")
    return _SmoothL1.apply(z, h)


@torch.no_grad()
def ema_update(encoder, target_encoder, m):
    """pretrain_jepa.py:431-432 over the flat buffers: one kernel instead of a Python loop over 150 tensors."""
    q, k = encoder.flat_parameters(), target_encoder.flat_parameters()
    _lib.check(_lib.lib().bvc_op_ema(k.data_ptr(), q.data_ptr(), k.numel(), float(m), _lib.current_stream_ptr()), "bvc_op_ema")
    if hasattr(target_encoder, "_shadow_invalidate"):
        target_encoder._shadow_invalidate()       # written through a raw pointer: the context's bf16 copy no longer matches


class _TokenMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, N, D = x.shape
        xf = x.detach().to(torch.float32).contiguous()
        out = torch.empty((B, D), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().bvc_op_token_mean(xf.data_ptr(), B, N, D, out.data_ptr(), _lib.current_stream_ptr()), "bvc_op_token_mean")
        ctx.shape = (B, N, D)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, D = ctx.shape
        gf = g.detach().to(torch.float32).contiguous()
        dx = torch.empty((B, N, D), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().bvc_op_token_mean_bwd(gf.data_ptr(), B, N, D, dx.data_ptr(), _lib.current_stream_ptr()),
                   "bvc_op_token_mean_bwd")
        return dx


def token_mean(x):
    """`x.mean(1)` over the token axis of (B, N, D) encoder output (benchmarks/compute_embeddings_jepa.py:242), one kernel."""
    if not x.is_cuda or x.dim() != 3 or x.shape[-1] % 4:
        raise _lib.BvcError("token_mean needs a CUDA (B, N, D) tensor with D % 4 == 0")
    return _TokenMean.apply(x)
