"""CPU oracle for the JEPA rows (encoder, predictor, targets, loss, EMA)  --  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32 restatement of pretraining/predictive/:
  vision_transformer.py:29-78    PositionalEncoding3D (baked into pos_embed / predictor_pos_embed)
  vision_transformer.py:186-231  Attention / Block (fused qkv, softmax(q k^T d^-1/2) v, proj, MLP with exact GELU, pre-LN)
  vision_transformer.py:234-261  PatchEmbed (Conv3d, stride = kernel)
  vision_transformer.py:378-402  VisionTransformer.forward
  vision_transformer.py:494-535  VisionTransformerPredictor.forward
  tensors.py:53-71               apply_masks / repeat_interleave_batch
  pretrain_jepa.py:383-402       forward_target / forward_context / smooth-L1 loss
  pretrain_jepa.py:426-432       EMA update
Pinned by tests/golden/jepa_*.json, written by oracle/make_golden.py from the reference's own modules (imported in the
build container with the same weights).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


@dataclasses.dataclass
class JepaConfig:
    image_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_frames: int = 2
    tubelet_size: int = 1
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    pred_dim: int = 384
    pred_depth: int = 6
    eps: float = 1e-6

    @property
    def sequence_shape(self):
        g = self.image_size // self.patch_size
        return (self.num_frames // self.tubelet_size, g, g)

    @property
    def seq_len(self):
        s = self.sequence_shape
        return s[0] * s[1] * s[2]


VIT_B = JepaConfig()                                                    # slurmscripts/predictive: ViT-B, 2 frames, tubelet 1
TINY = JepaConfig(image_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=2, pred_dim=64, pred_depth=1)   # head dims 64 / 32
# predictor heads of 24 dims, as vit_large gives (16 heads on 384 dims, vision_transformer.py:447,572-576): 8 heads on 192 dims
TINY_HD24 = JepaConfig(image_size=64, patch_size=16, embed_dim=512, depth=1, num_heads=8, pred_dim=192, pred_depth=2)
VIT_L = JepaConfig(embed_dim=1024, depth=24, num_heads=16)


def positional_encoding_3d(sequence_shape, channels_out):
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


def _block_shapes(prefix, d, inter):
    return {f"{prefix}norm1.weight": (d,), f"{prefix}norm1.bias": (d,), f"{prefix}attn.qkv.weight": (3 * d, d),
            f"{prefix}attn.qkv.bias": (3 * d,), f"{prefix}attn.proj.weight": (d, d), f"{prefix}attn.proj.bias": (d,),
            f"{prefix}norm2.weight": (d,), f"{prefix}norm2.bias": (d,), f"{prefix}mlp.fc1.weight": (inter, d),
            f"{prefix}mlp.fc1.bias": (inter,), f"{prefix}mlp.fc2.weight": (d, inter), f"{prefix}mlp.fc2.bias": (d,)}


def encoder_shapes(cfg: JepaConfig) -> "Dict[str, tuple]":
    D = cfg.embed_dim
    out = {"pos_embed": (1, cfg.seq_len, D),
           "patch_embed.proj.weight": (D, cfg.in_chans, cfg.tubelet_size, cfg.patch_size, cfg.patch_size),
           "patch_embed.proj.bias": (D,)}
    for i in range(cfg.depth):
        out.update(_block_shapes(f"blocks.{i}.", D, int(D * cfg.mlp_ratio)))
    out["norm.weight"] = (D,)
    out["norm.bias"] = (D,)
    return out


def predictor_shapes(cfg: JepaConfig) -> "Dict[str, tuple]":
    D, Dp = cfg.embed_dim, cfg.pred_dim
    out = {"mask_token": (1, 1, Dp), "predictor_pos_embed": (1, cfg.seq_len, Dp),
           "predictor_embed.weight": (Dp, D), "predictor_embed.bias": (Dp,)}
    for i in range(cfg.pred_depth):
        out.update(_block_shapes(f"predictor_blocks.{i}.", Dp, int(Dp * cfg.mlp_ratio)))
    out.update({"predictor_norm.weight": (Dp,), "predictor_norm.bias": (Dp,),
                "predictor_proj.weight": (D, Dp), "predictor_proj.bias": (D,)})
    return out


def make_params(shapes, cfg: JepaConfig, seed: int):
    """Deterministic weights: normal(0, 0.02) matrices, small non-trivial biases / LN / mask token, the real sinusoid tables."""
    out = {}
    for i, (k, shp) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(7000 * (seed + 1) + i)
        if k in ("pos_embed", "predictor_pos_embed"):
            t = positional_encoding_3d(cfg.sequence_shape, shp[-1])
        elif len(shp) >= 2 and k != "mask_token":
            t = torch.randn(shp, generator=g) * 0.02
        elif k.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
            t = 1 + torch.randn(shp, generator=g) * 0.05
        else:
            t = torch.randn(shp, generator=g) * 0.02
        out[k] = t.float().contiguous()
    return out


def _block_bf16(x, p, prefix, heads, eps, pol):
    """`_block` under a bf16-operand policy (videomae_oracle_bf16.Policy): operands of every product rounded as the policy says,
    f32 accumulation, f32 LayerNorm / softmax / residual stream; the same custom autograd functions as the VideoMAE oracle."""
    from . import videomae_oracle_bf16 as vb
    B, N, D = x.shape
    d = D // heads
    h = F.layer_norm(x, (D,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    qkv = vb._Round.apply(vb.linear(h, p[prefix + "attn.qkv.weight"], p[prefix + "attn.qkv.bias"], pol), pol)
    qkv = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    y = vb._Attention.apply(qkv[0], qkv[1], qkv[2], pol).transpose(1, 2).reshape(B, N, D)
    x = x + vb.linear(y, p[prefix + "attn.proj.weight"], p[prefix + "attn.proj.bias"], pol)
    h = F.layer_norm(x, (D,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    h = vb._Gelu.apply(vb.linear(h, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"], pol), pol)
    return x + vb.linear(h, p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"], pol, round_dx=False)


def _lin(x, w, b, pol):
    if pol is None:
        return F.linear(x, w, b)
    from . import videomae_oracle_bf16 as vb
    return vb.linear(x, w, b, pol)


def _block(x, p, prefix, heads, eps, pol=None):
    if pol is not None:
        return _block_bf16(x, p, prefix, heads, eps, pol)
    B, N, D = x.shape
    d = D // heads
    h = F.layer_norm(x, (D,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    qkv = F.linear(h, p[prefix + "attn.qkv.weight"], p[prefix + "attn.qkv.bias"]).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(dim=-1)
    y = (a @ v).transpose(1, 2).reshape(B, N, D)
    x = x + F.linear(y, p[prefix + "attn.proj.weight"], p[prefix + "attn.proj.bias"])
    h = F.layer_norm(x, (D,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"]))
    return x + F.linear(h, p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"])


def apply_masks(x, masks):
    return torch.cat([torch.gather(x, 1, m.unsqueeze(-1).repeat(1, 1, x.size(-1))) for m in masks], dim=0)


def repeat_interleave_batch(x, B, repeat):
    N = len(x) // B
    return torch.cat([torch.cat([x[i * B:(i + 1) * B] for _ in range(repeat)], dim=0) for i in range(N)], dim=0)


def encoder_forward(cfg: JepaConfig, p, imgs, masks=None, pol=None):
    """imgs (B, T, C, H, W); masks: list of (B, N) int64 index tensors or None.  pol: bf16-operand policy (None = fp32)."""
    if pol is None:
        x = F.conv3d(imgs.permute(0, 2, 1, 3, 4), p["patch_embed.proj.weight"], p["patch_embed.proj.bias"],
                     stride=(cfg.tubelet_size, cfg.patch_size, cfg.patch_size)).flatten(2).transpose(1, 2)
    else:        # the same convolution as a product over patches (k order c, dt, dy, dx = the Conv3d weight's layout)
        B, T, C, H, W = imgs.shape
        ts, ps = cfg.tubelet_size, cfg.patch_size
        pt = imgs.permute(0, 2, 1, 3, 4).reshape(B, C, T // ts, ts, H // ps, ps, W // ps, ps)
        pt = pt.permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, -1, C * ts * ps * ps)
        x = _lin(pt, p["patch_embed.proj.weight"].reshape(cfg.embed_dim, -1), p["patch_embed.proj.bias"], pol)
    x = x + p["pos_embed"]
    if masks is not None:
        x = apply_masks(x, masks)
    for i in range(cfg.depth):
        x = _block(x, p, f"blocks.{i}.", cfg.num_heads, cfg.eps, pol)
    return F.layer_norm(x, (cfg.embed_dim,), p["norm.weight"], p["norm.bias"], cfg.eps)


def predictor_forward(cfg: JepaConfig, p, x, masks_x, masks, pol=None):
    B = len(x) // len(masks_x)
    x = _lin(x, p["predictor_embed.weight"], p["predictor_embed.bias"], pol)
    x = x + apply_masks(p["predictor_pos_embed"].repeat(B, 1, 1), masks_x)
    n_ctx = x.shape[1]
    pos = repeat_interleave_batch(apply_masks(p["predictor_pos_embed"].repeat(B, 1, 1), masks), B, repeat=len(masks_x))
    pred = p["mask_token"].repeat(pos.size(0), pos.size(1), 1) + pos
    x = torch.cat([x.repeat(len(masks), 1, 1), pred], dim=1)
    for i in range(cfg.pred_depth):
        x = _block(x, p, f"predictor_blocks.{i}.", cfg.num_heads, cfg.eps, pol)
    x = F.layer_norm(x, (cfg.pred_dim,), p["predictor_norm.weight"], p["predictor_norm.bias"], cfg.eps)
    return _lin(x[:, n_ctx:], p["predictor_proj.weight"], p["predictor_proj.bias"], pol)


def targets(cfg: JepaConfig, tgt_p, imgs, masks_enc, masks_pred, pol=None):
    with torch.no_grad():
        h = encoder_forward(cfg, tgt_p, imgs, pol=pol)
        h = F.layer_norm(h, (h.size(-1),))
        B = len(h)
        return repeat_interleave_batch(apply_masks(h, masks_pred), B, repeat=len(masks_enc))


def step(cfg: JepaConfig, enc_p, pred_p, tgt_p, imgs, masks_enc, masks_pred, grad_scale=1.0, pol=None):
    """train_step's forward/backward (pretrain_jepa.py:383-418).  Returns loss, encoder grads, predictor grads, z, h.
    pol: a videomae_oracle_bf16.Policy - the same step with bf16 OPERANDS (what the reference computes under CUDA autocast,
    pretrain_jepa.py:404-405, and what the build computes); None = the fp32 step."""
    ep = {k: v.detach().clone().requires_grad_(k != "pos_embed") for k, v in enc_p.items()}
    pp = {k: v.detach().clone().requires_grad_(k != "predictor_pos_embed") for k, v in pred_p.items()}
    h = targets(cfg, tgt_p, imgs, masks_enc, masks_pred, pol)
    z_ctx = encoder_forward(cfg, ep, imgs, masks_enc, pol)
    z = predictor_forward(cfg, pp, z_ctx, masks_enc, masks_pred, pol)
    loss = F.smooth_l1_loss(z, h)
    (loss * grad_scale).backward()
    ge = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in ep.items()}
    gp = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in pp.items()}
    return loss.detach(), ge, gp, z.detach(), h


def ema(target, online, m):
    for k in target:
        target[k].mul_(m).add_((1.0 - m) * online[k])


def synthetic_inputs(cfg: JepaConfig, B, seed, n_ctx, n_pred, n_sets=4):
    """imgs like the loader's output and index masks shaped like the collator's after update_masks (mask.py:21-38):
    context indices from temporal slot 0, the prediction sets from the last slot, ascending like torch.nonzero gives."""
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, cfg.num_frames, cfg.in_chans, cfg.image_size, cfg.image_size), generator=g, dtype=torch.uint8)
    imgs = (u8.float() / 255.0 - 0.5) / 0.25
    per = cfg.sequence_shape[1] * cfg.sequence_shape[2]
    last = (cfg.sequence_shape[0] - 1) * per
    enc = torch.stack([torch.sort(torch.randperm(per, generator=g)[:n_ctx]).values for _ in range(B)])
    preds = [torch.stack([torch.sort(torch.randperm(per, generator=g)[:n_pred]).values for _ in range(B)]) + last for _ in range(n_sets)]
    return imgs, [enc], preds
