"""VideoMAE pre-training model object with the interface the reference's entry point uses.

Reference seam (pretraining/generative/pretrain_videomae.py):
  :43-64   get_config / get_model  -> transformers.VideoMAEConfig / VideoMAEForPreTraining
  :66-70   load_state_dict(ckpt['model_state_dict'])        (transformers key names)
  :170-176 model.config.image_size / patch_size / num_frames / tubelet_size
  :178-181 .to(rank), DDP(model), .parameters(), .train(), .eval()
  :301-302 outputs = xmodel(inputs, bool_masked_pos=m); outputs.loss
  :312     scaler.scale(loss).backward()
The arithmetic (HF:531-671) runs in libbvc_hip.so; this file only owns parameters, the flat
parameter / gradient buffers and the autograd bridge.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .flat import FlatParamModule, query_layout


class VideoMAEConfig:
    """The fields of transformers.VideoMAEConfig that the path reads; unknown kwargs are kept as attributes."""

    def __init__(self, image_size=224, patch_size=16, num_channels=3, num_frames=16, tubelet_size=2,
                 hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, qkv_bias=True,
                 use_mean_pooling=True, decoder_num_attention_heads=6, decoder_hidden_size=384,
                 decoder_num_hidden_layers=4, decoder_intermediate_size=1536, norm_pix_loss=True, **kwargs):
        if hidden_act != "gelu":
            raise ValueError("only hidden_act='gelu' (exact erf GELU) is implemented")
        if not qkv_bias:
            raise ValueError("qkv_bias=False is not implemented")
        self.image_size, self.patch_size, self.num_channels = image_size, patch_size, num_channels
        self.num_frames, self.tubelet_size = num_frames, tubelet_size
        self.hidden_size, self.num_hidden_layers = hidden_size, num_hidden_layers
        self.num_attention_heads, self.intermediate_size = num_attention_heads, intermediate_size
        self.hidden_act, self.layer_norm_eps, self.initializer_range = hidden_act, layer_norm_eps, initializer_range
        self.qkv_bias, self.use_mean_pooling = qkv_bias, use_mean_pooling
        self.decoder_num_attention_heads, self.decoder_hidden_size = decoder_num_attention_heads, decoder_hidden_size
        self.decoder_num_hidden_layers, self.decoder_intermediate_size = decoder_num_hidden_layers, decoder_intermediate_size
        self.norm_pix_loss = norm_pix_loss
        self.decoder_norm_eps = 1e-5   # nn.LayerNorm default used by decoder.norm (HF:484)
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def seq_length(self):
        g = self.image_size // self.patch_size
        return (self.num_frames // self.tubelet_size) * g * g

    def to_c(self) -> _lib.VideoMAEConfigC:
        return _lib.VideoMAEConfigC(
            self.image_size, self.patch_size, self.num_channels, self.num_frames, self.tubelet_size,
            self.hidden_size, self.num_hidden_layers, self.num_attention_heads, self.intermediate_size,
            self.decoder_hidden_size, self.decoder_num_hidden_layers, self.decoder_num_attention_heads,
            self.decoder_intermediate_size, float(self.layer_norm_eps), float(self.decoder_norm_eps),
            int(bool(self.norm_pix_loss)))


@dataclass
class VideoMAEForPreTrainingOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None


def param_layout(config: VideoMAEConfig):
    """[(state-dict key, offset, shape)] in flat-buffer order, as libbvc_hip.so defines it."""
    L = _lib.lib()
    return query_layout(L.bvc_videomae_param_count, L.bvc_videomae_param_numel, L.bvc_videomae_param_info, config.to_c())


class _Step(torch.autograd.Function):
    """loss = step(pixels, mask); backward fills the flat gradient buffer and hands out views as .grad."""

    @staticmethod
    def forward(ctx, anchor, model, pixels, mask, want_logits):
        ctx.model = model
        loss, logits = model._run_forward(pixels, mask, want_logits)
        ctx.stamp = model._stamp_forward()
        ctx.mark_non_differentiable(logits) if logits is not None else None
        return loss, logits

    @staticmethod
    def backward(ctx, grad_loss, _grad_logits):
        ctx.model._check_generation(ctx.stamp)
        ctx.model._run_backward(grad_loss)
        return None, None, None, None, None


class VideoMAEForPreTraining(FlatParamModule):
    _shadow_fn = "bvc_videomae_shadow"
    """Drop-in for transformers.VideoMAEForPreTraining on the pre-training path (same state-dict keys)."""

    def __init__(self, config: VideoMAEConfig):
        super().__init__()
        self.config = config
        layout, numel = param_layout(config)
        std = config.initializer_range

        def init(name, shape):
            # transformers' _init_weights: normal(0, initializer_range) for Linear/Conv3d weights, zero biases,
            # LayerNorm weight 1 / bias 0; mask_token is created as zeros (HF:515)
            if len(shape) >= 2 and name != "mask_token":
                return torch.empty(shape).normal_(0.0, std)
            if name.endswith("layernorm_before.weight") or name.endswith("layernorm_after.weight") or name == "decoder.norm.weight":
                return torch.ones(shape)
            return torch.zeros(shape)

        self._init_flat(layout, numel, init)
        self._ctx = None
        self._ctx_key = None
        self.strict_mask_check = False
        self._nmask_cache = {}
        # uint8 pixel_values are normalised on the GPU as (u / 255 - mean) / std, the loader's ToTensor + Normalize
        # (homeview.py:221-230 uses 0.5 / 0.25 for every channel); f32 pixel_values are taken as already normalised
        self.pixel_mean, self.pixel_std = 0.5, 0.25

    # ---- library context
    def _get_ctx(self, batch, nmask):
        key = (batch, nmask, self._flat.device.index)
        if self._ctx is not None and self._ctx_key[1] == nmask and self._ctx_key[2] == key[2] and self._ctx_key[0] >= batch:
            return self._ctx
        self._free_ctx()
        h = ctypes.c_void_p()
        cc = self.config.to_c()
        _lib.check(_lib.lib().bvc_videomae_create(ctypes.byref(cc), batch, nmask, ctypes.byref(h)), "bvc_videomae_create")
        self._ctx, self._ctx_key = h, key
        return h

    def _free_ctx(self):
        if self._ctx is not None:
            _lib.lib().bvc_videomae_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._free_ctx()
        except Exception:
            pass

    # ---- step
    def _num_masked(self, mask):
        """Masked tokens per clip.  One host sync on the first call per mask shape; afterwards the cached count is VERIFIED
        asynchronously: every call leaves `count(row 0) != cached` and `rows differ` flags in a pinned word that the next call
        reads (a ratio change at the same shape - a validation phase, a curriculum stage - raises on the following step
        instead of training on NaN losses that GradScaler silently skips)."""
        key = tuple(mask.shape)
        pend = getattr(self, "_nmask_pending", None)
        if pend is not None:
            ev, flag, pkey = pend
            ev.synchronize()                    # the previous step's check: long done, no stall
            self._nmask_pending = None
            if int(flag[0]) != 0:
                self._nmask_cache.pop(pkey, None)
                raise ValueError("bool_masked_pos: the number of masked patches per clip changed (or differs between clips) "
                                 "without a new model object; every clip must mask the same number of patches")
        if self.strict_mask_check or key not in self._nmask_cache:
            counts = mask.sum(dim=1)
            n = int(counts[0])       # one host sync, first call per shape only
            if not bool((counts == n).all()):
                raise ValueError("every clip must have the same number of masked patches")
            self._nmask_cache[key] = n
            return n
        n = self._nmask_cache[key]
        if not hasattr(self, "_nmask_flag"):
            self._nmask_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        bad = (mask.sum(dim=1) != n).any().to(torch.int32).reshape(1)
        self._nmask_flag.copy_(bad, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(mask.device))
        self._nmask_pending = (ev, self._nmask_flag, key)
        return n

    def _run_forward(self, pixels, mask, want_logits):
        cfg = self.config
        fmt = _lib.pixel_format(pixels, self.pixel_mean, self.pixel_std, cfg.num_channels)
        B = pixels.shape[0]
        nmask = self._num_masked(mask)
        h = self._get_ctx(B, nmask)
        loss = torch.empty((), dtype=torch.float32, device=pixels.device)
        logits = None
        if want_logits:
            pd = cfg.num_channels * cfg.tubelet_size * cfg.patch_size ** 2
            logits = torch.empty((B, nmask, pd), dtype=torch.float32, device=pixels.device)
        self._shadow_vouch(h)
        _lib.check(_lib.lib().bvc_videomae_forward_px(
            h, pixels.data_ptr(), ctypes.byref(fmt) if fmt is not None else None, mask.data_ptr(), B, self._flat.data_ptr(),
            loss.data_ptr(), logits.data_ptr() if logits is not None else None, _lib.current_stream_ptr()), "bvc_videomae_forward")
        self._shadow_established(h)
        self._live = (pixels, mask)   # keep the borrowed inputs alive until backward
        return loss, logits

    def _run_backward(self, grad_loss):
        target, accumulate = self._grad_target()
        g = grad_loss.detach().to(dtype=torch.float32).contiguous()
        cb = self._bucket_callback(accumulate)
        self._library_backward("bvc_videomae_backward",
                               _lib.lib().bvc_videomae_backward(self._ctx, g.data_ptr(), target.data_ptr(), cb, None, _lib.current_stream_ptr()))
        self._publish_grads(target, accumulate)
        self._live = None

    def forward(self, pixel_values, bool_masked_pos=None, output_logits=False, **kwargs):
        if bool_masked_pos is None:
            raise ValueError("One must provided a boolean mask ")
        if not pixel_values.is_cuda:
            raise _lib.BvcError("VideoMAEForPreTraining runs on a GPU only (libbvc_hip.so has no CPU path)")
        cfg = self.config
        B, T, C, H, W = pixel_values.shape
        if C != cfg.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
        if H != cfg.image_size or W != cfg.image_size or T != cfg.num_frames:
            raise ValueError(f"Input size ({T}x{H}*{W}) doesn't match model ({cfg.num_frames}x{cfg.image_size}*{cfg.image_size}).")
        if tuple(bool_masked_pos.shape) != (B, cfg.seq_length):
            raise ValueError(f"bool_masked_pos must have shape {(B, cfg.seq_length)}")
        self._ensure_flat(pixel_values.device)
        pixels = pixel_values.detach()
        pixels = (pixels if pixels.dtype == torch.uint8 else pixels.to(dtype=torch.float32)).contiguous()
        mask = bool_masked_pos.to(device=pixels.device, dtype=torch.bool).contiguous()
        anchor = self._param(self._names[0])
        if torch.is_grad_enabled() and anchor.requires_grad:
            loss, logits = _Step.apply(anchor, self, pixels, mask, output_logits)
        else:
            loss, logits = self._run_forward(pixels, mask, output_logits)
            self._stamp_forward()     # a pending backward of an earlier forward must not run on these activations
        return VideoMAEForPreTrainingOutput(loss=loss, logits=logits)

    # ---- parity probes
    def tap(self, name):
        """f32 copy of a saved activation of the last forward ('embed', 'enc<i>', 'x_full', 'dec<i>', 'labels')."""
        cfg = self.config
        cap = max(self._ctx_key[0] * cfg.seq_length * max(cfg.hidden_size, cfg.decoder_hidden_size),
                  self._ctx_key[0] * self._ctx_key[1] * cfg.num_channels * cfg.tubelet_size * cfg.patch_size ** 2)
        buf = torch.empty(cap, dtype=torch.float32, device=self._flat.device)
        n = ctypes.c_int64()
        _lib.check(_lib.lib().bvc_videomae_tap(self._ctx, name.encode(), buf.data_ptr(), cap, ctypes.byref(n),
                                               _lib.current_stream_ptr()), "bvc_videomae_tap")
        return buf[: n.value]


@dataclass
class ImageClassifierOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None
    last_hidden_state: Optional[torch.Tensor] = None


class VideoMAEForVideoClassification(FlatParamModule):
    """Encoder-only inference model of the embedding benchmark (benchmarks/compute_embeddings_videomae.py:78-96,253-264).

    Same sub-module tree / state-dict keys as transformers.VideoMAEForVideoClassification: ``videomae.embeddings.*``,
    ``videomae.encoder.*`` (so ``adapt_videomae``'s ``target.videomae.embeddings.load_state_dict(source.videomae.embeddings
    .state_dict())`` works against a pre-training model), ``fc_norm.*`` and, for ``num_labels > 0``, ``classifier.*``.
    ``forward(pixel_values).logits`` = classifier(fc_norm(mean over all tokens of the encoder output)); the reference uses
    ``num_labels=0`` (classifier = Identity), i.e. the logits ARE the embedding.  Forward only (no autograd).
    """

    def __init__(self, config: VideoMAEConfig):
        super().__init__()
        if not getattr(config, "use_mean_pooling", True):
            raise ValueError("only use_mean_pooling=True (the reference's setting) is implemented")
        self.config = config
        self.num_labels = int(getattr(config, "num_labels", 2))
        full, _ = param_layout(config)
        cc = config.to_c()
        numel = int(_lib.lib().bvc_videomae_encoder_param_numel(ctypes.byref(cc)))
        layout = [e for e in full if e[0].startswith("videomae.")]
        assert sum(int(torch.Size(e[2]).numel()) for e in layout) == numel
        std = config.initializer_range

        def init(name, shape):
            if len(shape) >= 2:
                return torch.empty(shape).normal_(0.0, std)
            if name.endswith("layernorm_before.weight") or name.endswith("layernorm_after.weight"):
                return torch.ones(shape)
            return torch.zeros(shape)

        self._init_flat(layout, numel, init)
        self.fc_norm = nn.LayerNorm(config.hidden_size)    # as in HF: default eps 1e-5, not config.layer_norm_eps
        self.classifier = nn.Linear(config.hidden_size, self.num_labels) if self.num_labels > 0 else nn.Identity()
        if self.num_labels > 0:
            nn.init.normal_(self.classifier.weight, 0.0, std)
            nn.init.zeros_(self.classifier.bias)
        self._ctx = None
        self._ctx_key = None
        self.pixel_mean, self.pixel_std = 0.5, 0.25      # for uint8 pixel_values, as in VideoMAEForPreTraining

    def _get_ctx(self, batch):
        key = (batch, self._flat.device.index)
        if self._ctx is not None and self._ctx_key[1] == key[1] and self._ctx_key[0] >= batch:
            return self._ctx
        self._free_ctx()
        h = ctypes.c_void_p()
        cc = self.config.to_c()
        _lib.check(_lib.lib().bvc_videomae_encoder_create(ctypes.byref(cc), batch, ctypes.byref(h)), "bvc_videomae_encoder_create")
        self._ctx, self._ctx_key = h, key
        return h

    def _free_ctx(self):
        if self._ctx is not None:
            _lib.lib().bvc_videomae_encoder_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._free_ctx()
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, pixel_values=None, labels=None, output_last_hidden_state=False, **kwargs):
        if labels is not None:
            raise NotImplementedError("the classification loss / fine-tuning path is outside the pre-training hot path")
        if pixel_values is None or not pixel_values.is_cuda:
            raise _lib.BvcError("VideoMAEForVideoClassification runs on a GPU only (libbvc_hip.so has no CPU path)")
        cfg = self.config
        B, T, C, H, W = pixel_values.shape
        if C != cfg.num_channels or H != cfg.image_size or W != cfg.image_size or T != cfg.num_frames:
            raise ValueError(f"Input size ({T}x{C}x{H}*{W}) doesn't match model ({cfg.num_frames}x{cfg.num_channels}x{cfg.image_size}*{cfg.image_size}).")
        dev = pixel_values.device
        self._ensure_flat(dev)
        pixels = pixel_values.detach()
        pixels = (pixels if pixels.dtype == torch.uint8 else pixels.to(dtype=torch.float32)).contiguous()
        fmt = _lib.pixel_format(pixels, self.pixel_mean, self.pixel_std, cfg.num_channels)
        h = self._get_ctx(B)
        w = self.fc_norm.weight.detach().to(device=dev, dtype=torch.float32).contiguous()
        b = self.fc_norm.bias.detach().to(device=dev, dtype=torch.float32).contiguous()
        pooled = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device=dev)
        tokens = torch.empty((B, cfg.seq_length, cfg.hidden_size), dtype=torch.float32, device=dev) if output_last_hidden_state else None
        _lib.check(_lib.lib().bvc_videomae_encode_px(
            h, pixels.data_ptr(), ctypes.byref(fmt) if fmt is not None else None, B, self._flat.data_ptr(), w.data_ptr(), b.data_ptr(),
            float(self.fc_norm.eps),
            tokens.data_ptr() if tokens is not None else None, pooled.data_ptr(), _lib.current_stream_ptr()), "bvc_videomae_encode")
        logits = self.classifier(pooled)
        return ImageClassifierOutput(logits=logits, last_hidden_state=tokens)


def get_config(image_size, args):
    """pretrain_videomae.py:43-58 (only architecture='base' exists in the reference)."""
    if getattr(args, "architecture", "base") != "base":
        raise ValueError("only architecture='base' is defined by the reference")
    return VideoMAEConfig(image_size=image_size, patch_size=16, num_channels=3, num_frames=args.num_frames,
                          tubelet_size=args.tubelet_size, hidden_size=768, num_hidden_layers=12,
                          num_attention_heads=12, intermediate_size=3072, initializer_range=0.02,
                          use_mean_pooling=True, decoder_num_attention_heads=6, decoder_hidden_size=384,
                          decoder_num_hidden_layers=4, decoder_intermediate_size=1536, norm_pix_loss=True)


def get_model(image_size, args):
    """pretrain_videomae.py:61-64"""
    return VideoMAEForPreTraining(get_config(image_size, args))
