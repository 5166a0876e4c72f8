"""CPU oracle for the VideoMAE pre-training step  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 restatement of the arithmetic the reference's
VideoMAE step runs (the reference delegates it to the third-party package
`transformers`, class VideoMAEForPreTraining; pinned here to transformers 5.15.0,
see SURVEY.md section 8c).  It exists to *check* the HIP path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product package never does.

Parity pinning: `oracle/make_golden.py` (run in the build container, where
`transformers` is importable) loads the same deterministic weights into
transformers' VideoMAEForPreTraining and into this restatement and writes the
fixtures under `tests/golden/`; `tests/test_oracle_golden.py` re-checks this file
against those fixtures on every run.  The reference repository itself holds no
tests or golden vectors for this path.

Reference call sites followed (paths relative to the reference checkout):
  pretraining/generative/pretrain_videomae.py:43-58   config (ViT-B enc, 4-layer dec)
  pretraining/generative/pretrain_videomae.py:292-304 forward_loss closure
  pretraining/generative/mask.py:3-24                 TubeMaskingGenerator
  pretraining/generative/loggingtools.py:98-119       grad_logger probes
HF = transformers/models/videomae/modeling_videomae.py (5.15.0):
  HF:80-91   sinusoid table            HF:109-124 embeddings (+pos, drop masked)
  HF:157-177 tube patch embed Conv3d   HF:181-206 eager attention
  HF:339-357 pre-LN layer              HF:492-503 decoder (slice, LN, head)
  HF:566-582 enc->dec glue             HF:588-664 pixel targets + MSE
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


@dataclasses.dataclass
class OracleConfig:
    """Mirror of the VideoMAEConfig fields the path reads (pretrain_videomae.py:51-57)."""
    image_size: int = 224
    patch_size: int = 16
    num_channels: int = 3
    num_frames: int = 16
    tubelet_size: int = 2
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    decoder_hidden_size: int = 384
    decoder_num_hidden_layers: int = 4
    decoder_num_attention_heads: int = 6
    decoder_intermediate_size: int = 1536
    layer_norm_eps: float = 1e-12      # HF default, used by every VideoMAELayer LN
    decoder_norm_eps: float = 1e-5     # nn.LayerNorm default for decoder.norm (HF:484)
    norm_pix_loss: bool = True

    @property
    def grid(self):
        return (self.num_frames // self.tubelet_size,
                self.image_size // self.patch_size,
                self.image_size // self.patch_size)

    @property
    def seq_len(self):
        g = self.grid
        return g[0] * g[1] * g[2]

    @property
    def patch_dim(self):
        return self.num_channels * self.tubelet_size * self.patch_size ** 2


BASE = OracleConfig()
TINY = OracleConfig(image_size=64, patch_size=16, num_frames=4, tubelet_size=2,
                    hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                    intermediate_size=256, decoder_hidden_size=64,
                    decoder_num_hidden_layers=1, decoder_num_attention_heads=1,
                    decoder_intermediate_size=128)


# ----------------------------------------------------------------------------- params
def param_shapes(cfg: OracleConfig) -> "Dict[str, tuple]":
    """State-dict keys and shapes, in the order transformers 5.15.0 registers them."""
    D, Dd = cfg.hidden_size, cfg.decoder_hidden_size
    out: Dict[str, tuple] = {}
    out["mask_token"] = (1, 1, Dd)
    pe = "videomae.embeddings.patch_embeddings.projection."
    out[pe + "weight"] = (D, cfg.num_channels, cfg.tubelet_size, cfg.patch_size, cfg.patch_size)
    out[pe + "bias"] = (D,)

    def layer(prefix, d, inter):
        for nm in ("query", "key", "value"):
            out[f"{prefix}attention.attention.{nm}.weight"] = (d, d)
            out[f"{prefix}attention.attention.{nm}.bias"] = (d,)
        out[f"{prefix}attention.output.dense.weight"] = (d, d)
        out[f"{prefix}attention.output.dense.bias"] = (d,)
        out[f"{prefix}intermediate.dense.weight"] = (inter, d)
        out[f"{prefix}intermediate.dense.bias"] = (inter,)
        out[f"{prefix}output.dense.weight"] = (d, inter)
        out[f"{prefix}output.dense.bias"] = (d,)
        out[f"{prefix}layernorm_before.weight"] = (d,)
        out[f"{prefix}layernorm_before.bias"] = (d,)
        out[f"{prefix}layernorm_after.weight"] = (d,)
        out[f"{prefix}layernorm_after.bias"] = (d,)

    for i in range(cfg.num_hidden_layers):
        layer(f"videomae.encoder.layer.{i}.", D, cfg.intermediate_size)
    out["encoder_to_decoder.weight"] = (Dd, D)
    for i in range(cfg.decoder_num_hidden_layers):
        layer(f"decoder.decoder_layers.{i}.", Dd, cfg.decoder_intermediate_size)
    out["decoder.norm.weight"] = (Dd,)
    out["decoder.norm.bias"] = (Dd,)
    out["decoder.head.weight"] = (cfg.patch_dim, Dd)
    out["decoder.head.bias"] = (cfg.patch_dim,)
    return out


def make_params(cfg: OracleConfig, seed: int = 0, perturb: bool = True) -> "Dict[str, torch.Tensor]":
    """Deterministic weights shared by oracle, fixtures and the HIP path's tests.

    Tensor i (state-dict order) is drawn from its own torch.Generator seeded with
    1000*seed + i.  Matrices follow the reference's init, normal(0, 0.02)
    (transformers/modeling_utils.py _init_weights); with `perturb` the biases, LN
    affine parameters and mask token get small non-trivial values too so that a
    dropped bias or LN term cannot hide behind a zero.
    """
    out = {}
    for i, (k, shp) in enumerate(param_shapes(cfg).items()):
        g = torch.Generator().manual_seed(1000 * seed + i)
        if len(shp) >= 2 and k != "mask_token":
            t = torch.randn(shp, generator=g) * 0.02
        elif "layernorm" in k and k.endswith("weight") or k == "decoder.norm.weight":
            t = torch.ones(shp) + (torch.randn(shp, generator=g) * 0.05 if perturb else 0)
        else:
            t = torch.randn(shp, generator=g) * 0.02 if perturb else torch.zeros(shp)
        out[k] = t.float().contiguous()
    return out


# ----------------------------------------------------------------------------- inputs
def tube_mask(grid, mask_ratio: float, rng: np.random.RandomState) -> np.ndarray:
    """Restatement of TubeMaskingGenerator.__call__ (pretraining/generative/mask.py:17-24).

    One per-frame mask with int(ratio * H*W) ones is shuffled and tiled over the
    temporal slots.  `rng` replaces the reference's global (unseeded) numpy RNG.
    """
    frames, h, w = grid
    per_frame = h * w
    n_mask = int(mask_ratio * per_frame)
    m = np.hstack([np.zeros(per_frame - n_mask), np.ones(n_mask)])
    rng.shuffle(m)
    return np.tile(m, (frames, 1)).flatten()


def synthetic_batch(cfg: OracleConfig, batch: int, seed: int, mask_ratio: float = 0.9):
    """Synthetic clips shaped like the reference loader's output.

    uint8 ~ U{0..255} frames through (x/255 - 0.5)/0.25  (homeview.py:218-231),
    tube masks from a seeded RandomState.
    """
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (batch, cfg.num_frames, cfg.num_channels, cfg.image_size, cfg.image_size),
                       generator=g, dtype=torch.uint8)
    pixels = (u8.float() / 255.0 - 0.5) / 0.25
    rng = np.random.RandomState(seed)
    mask = np.stack([tube_mask(cfg.grid, mask_ratio, rng) for _ in range(batch)]).astype(bool)
    return pixels, torch.from_numpy(mask)


# ----------------------------------------------------------------------------- model
def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """HF:80-91.  float64 numpy, cast to float32 at the end like torch.FloatTensor(np)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000.0, 2 * (j // 2) / d_hid)
    table = angle.copy()
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.from_numpy(table).float()


def _layer(x, p, prefix, heads, eps, taps, tapname):
    """One pre-LN transformer layer (HF:339-357, attention HF:181-257, MLP HF:299-322)."""
    B, N, D = x.shape
    d = D // heads
    a = prefix + "attention.attention."
    h = F.layer_norm(x, (D,), p[prefix + "layernorm_before.weight"], p[prefix + "layernorm_before.bias"], eps)
    q = F.linear(h, p[a + "query.weight"], p[a + "query.bias"]).view(B, N, heads, d).transpose(1, 2)
    k = F.linear(h, p[a + "key.weight"], p[a + "key.bias"]).view(B, N, heads, d).transpose(1, 2)
    v = F.linear(h, p[a + "value.weight"], p[a + "value.bias"]).view(B, N, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5)
    pr = torch.softmax(s, dim=-1)
    ctx = torch.matmul(pr, v).transpose(1, 2).reshape(B, N, D)
    x = x + F.linear(ctx, p[prefix + "attention.output.dense.weight"], p[prefix + "attention.output.dense.bias"])
    h = F.layer_norm(x, (D,), p[prefix + "layernorm_after.weight"], p[prefix + "layernorm_after.bias"], eps)
    h = F.gelu(F.linear(h, p[prefix + "intermediate.dense.weight"], p[prefix + "intermediate.dense.bias"]))
    x = x + F.linear(h, p[prefix + "output.dense.weight"], p[prefix + "output.dense.bias"])
    if taps is not None:
        taps[tapname] = x
    return x


def pixel_labels(cfg: OracleConfig, pixel_values: torch.Tensor, bool_masked_pos: torch.Tensor) -> torch.Tensor:
    """HF:588-661: un-normalise with the ImageNet constants, patchify to (B, L, ts*p*p, C),
    normalise each patch per channel with the unbiased variance, keep the masked rows."""
    B, T, C, H, W = pixel_values.shape
    ts, ps = cfg.tubelet_size, cfg.patch_size
    mean = torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=pixel_values.dtype)[None, None, :, None, None]
    std = torch.tensor(IMAGENET_DEFAULT_STD, dtype=pixel_values.dtype)[None, None, :, None, None]
    frames = pixel_values * std + mean if C == 3 else pixel_values
    frames = frames.view(B, T // ts, ts, C, H // ps, ps, W // ps, ps).permute(0, 1, 4, 6, 2, 5, 7, 3).contiguous()
    frames = frames.view(B, cfg.seq_len, ts * ps * ps, C)
    if cfg.norm_pix_loss:
        frames = (frames - frames.mean(dim=-2, keepdim=True)) / (
            frames.var(dim=-2, unbiased=True, keepdim=True).sqrt() + 1e-6)
    patches = frames.view(B, cfg.seq_len, ts * ps * ps * C)
    return patches[bool_masked_pos].reshape(B, -1, ts * ps * ps * C)


def forward(cfg: OracleConfig, p: "Dict[str, torch.Tensor]", pixel_values: torch.Tensor,
            bool_masked_pos: torch.Tensor, taps: Optional[dict] = None):
    """VideoMAEForPreTraining.forward (HF:531-671) in fp32.  Returns (loss, logits, labels).

    `taps`, when given, receives the per-layer activations the parity tests compare:
    'embed' (visible tokens after pos-emb), 'enc{i}', 'x_full', 'dec{i}', 'logits', 'labels'.
    """
    B, T, C, H, W = pixel_values.shape
    D, Dd = cfg.hidden_size, cfg.decoder_hidden_size
    L = cfg.seq_len
    # HF:164-177  tube patch embedding, token = t'*h*w + y'*w + x'
    w = p["videomae.embeddings.patch_embeddings.projection.weight"]
    x = F.conv3d(pixel_values.permute(0, 2, 1, 3, 4), w,
                 p["videomae.embeddings.patch_embeddings.projection.bias"],
                 stride=(cfg.tubelet_size, cfg.patch_size, cfg.patch_size))
    x = x.flatten(2).transpose(1, 2)
    # HF:114-122  + sinusoid, keep visible tokens (row-major ascending token order)
    x = x + sinusoid_table(L, D)[None]
    x = x[~bool_masked_pos].reshape(B, -1, D)
    if taps is not None:
        taps["embed"] = x
    for i in range(cfg.num_hidden_layers):
        x = _layer(x, p, f"videomae.encoder.layer.{i}.", cfg.num_attention_heads, cfg.layer_norm_eps, taps, f"enc{i}")
    # HF:566-582  (no final encoder LN in pre-training: use_mean_pooling=True, HF:406-409)
    x = F.linear(x, p["encoder_to_decoder.weight"])
    pos = sinusoid_table(L, Dd)[None].expand(B, -1, -1)
    pos_vis = pos[~bool_masked_pos].reshape(B, -1, Dd)
    pos_msk = pos[bool_masked_pos].reshape(B, -1, Dd)
    x = torch.cat([x + pos_vis, p["mask_token"] + pos_msk], dim=1)
    if taps is not None:
        taps["x_full"] = x
    for i in range(cfg.decoder_num_hidden_layers):
        x = _layer(x, p, f"decoder.decoder_layers.{i}.", cfg.decoder_num_attention_heads, cfg.layer_norm_eps, taps, f"dec{i}")
    n_mask = pos_msk.shape[1]
    x = x[:, -n_mask:]
    x = F.layer_norm(x, (Dd,), p["decoder.norm.weight"], p["decoder.norm.bias"], cfg.decoder_norm_eps)
    logits = F.linear(x, p["decoder.head.weight"], p["decoder.head.bias"])
    with torch.no_grad():
        labels = pixel_labels(cfg, pixel_values, bool_masked_pos)
    loss = F.mse_loss(logits, labels)
    if taps is not None:
        taps["logits"] = logits
        taps["labels"] = labels
    return loss, logits, labels


def encode(cfg: OracleConfig, p: "Dict[str, torch.Tensor]", pixel_values: torch.Tensor, fc_norm_w=None, fc_norm_b=None,
           fc_norm_eps: float = 1e-5):
    """Encoder-only inference = VideoMAEForVideoClassification(num_labels=0).forward in fp32
    (benchmarks/compute_embeddings_videomae.py:78-96,253-264): every token, no mask, encoder, mean over tokens, fc_norm.
    Returns (embedding (B, hidden), last_hidden_state (B, L, hidden))."""
    D, L = cfg.hidden_size, cfg.seq_len
    x = F.conv3d(pixel_values.permute(0, 2, 1, 3, 4), p["videomae.embeddings.patch_embeddings.projection.weight"],
                 p["videomae.embeddings.patch_embeddings.projection.bias"],
                 stride=(cfg.tubelet_size, cfg.patch_size, cfg.patch_size)).flatten(2).transpose(1, 2)
    x = x + sinusoid_table(L, D)[None]
    for i in range(cfg.num_hidden_layers):
        x = _layer(x, p, f"videomae.encoder.layer.{i}.", cfg.num_attention_heads, cfg.layer_norm_eps, None, "")
    pooled = x.mean(1)
    if fc_norm_w is not None:
        pooled = F.layer_norm(pooled, (D,), fc_norm_w, fc_norm_b, fc_norm_eps)
    return pooled, x


GRAD_PROBES = (  # loggingtools.py:107-116  (grad-EFL, grad-ELL, grad-DLL)
    "videomae.embeddings.patch_embeddings.projection.weight",
    "encoder_to_decoder.weight",
    "decoder.head.weight",
)


def step(cfg: OracleConfig, params: "Dict[str, torch.Tensor]", pixel_values, bool_masked_pos,
         grad_scale: float = 1.0, taps: Optional[dict] = None):
    """Forward + backward of one batch.  Returns (loss, grads dict) with grads multiplied
    by `grad_scale` (what GradScaler feeds into backward, pretrain_videomae.py:312)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss, _, _ = forward(cfg, p, pixel_values, bool_masked_pos, taps)
    (loss * grad_scale).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    return loss.detach(), grads


def sgd_nesterov_step(params, grads, bufs, lr=0.1, momentum=0.9, weight_decay=0.0):
    """torch.optim.SGD(nesterov=True) update (pretrain_videomae.py:187-189), restated:
    buf = m*buf + g (buf = g on the first step); p -= lr*(g + m*buf)."""
    for k in params:
        g = grads[k]
        if weight_decay:
            g = g + weight_decay * params[k]
        if k not in bufs:
            bufs[k] = g.clone()
        else:
            bufs[k].mul_(momentum).add_(g)
        params[k].sub_(lr * (g + momentum * bufs[k]))
