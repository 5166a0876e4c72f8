"""bf16-OPERAND mode of the CPU oracle for the VideoMAE pre-training step  --  TEST INFRASTRUCTURE ONLY.

`videomae_oracle.py` restates the reference's step in fp32 (what the reference computes on a CPU).  On a GPU the reference runs the
same step under `torch.autocast(dtype=bfloat16)` (pretraining/generative/pretrain_videomae.py:306-308): every matrix product takes
bf16 operands and accumulates in f32, LayerNorm / softmax / the loss stay in f32 (dtype column).  This file
restates the step with THAT operand policy, switchable per operand family, so that a parity report can say how much of a
deviation from the fp32 step ANY bf16-operand run shows and how much is a particular build's own choice:

  Policy.weights    weight operands of every product rounded to bf16 (autocast's weight cast; the build's bf16 weight shadow)
  Policy.acts       forward activation operands rounded: LayerNorm outputs, patches, q / k / v, attention probabilities, the
                    attention context, the GELU output
  Policy.grads      backward operands rounded: every dY that enters a product, dS, the dX / dq / dk / dv outputs
  Policy.gelu_grad  the saved gelu'(pre) rounded to bf16 (the build's forward epilogue stores it in bf16)

Everything else is f32 exactly as in `videomae_oracle.forward`: the residual stream, LayerNorm and softmax arithmetic, biases, the
accumulation of every product (f32 matmul of the rounded operands), weight gradients, the loss.  With every switch off the
functions below reproduce `videomae_oracle.step` to f32 round-off (tests/test_oracle_bf16.py).

Where this differs from CUDA autocast proper: autocast also rounds the OUTPUT of every linear layer to bf16 (so its residual
stream picks up a rounding per layer) and produces weight gradients in bf16; the build under test keeps both in f32.
`Policy.autocast_outputs` adds those two roundings; tests/test_oracle_bf16.py pins a single linear layer in that mode against
`torch.autocast("cpu", dtype=torch.bfloat16)` (forward bits and both gradients), which is where the two policies coincide.

Only `tests/`, `tools/` report scripts and `oracle/make_golden.py` import this file; the product package never does.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import videomae_oracle as vo


@dataclasses.dataclass(frozen=True)
class Policy:
    weights: bool = True
    acts: bool = True
    grads: bool = True
    gelu_grad: bool = True
    autocast_outputs: bool = False      # + linear outputs and weight gradients rounded (CUDA / CPU autocast proper)

    def label(self):
        on = [k for k in ("weights", "acts", "grads", "gelu_grad", "autocast_outputs") if getattr(self, k)]
        return "+".join(on) if on else "f32"


F32 = Policy(False, False, False, False)
BUILD = Policy()                         # the operand policy of libbvc_hip.so ("Data layout in HBM")


def _r(t, on=True):
    return t.to(torch.bfloat16).float() if on else t


class _Linear(torch.autograd.Function):
    """y = r(x) r(w)^T + b with f32 accumulation; backward takes r(dy): dx = r(dy) r(w), dw = r(dy)^T r(x), db = sum r(dy)."""

    @staticmethod
    def forward(ctx, x, w, b, pol, round_dx):
        xq, wq = _r(x, pol.acts), _r(w, pol.weights)
        ctx.save_for_backward(xq, wq)
        ctx.pol, ctx.round_dx, ctx.has_b = pol, round_dx, b is not None
        y = xq @ wq.t()
        if b is not None:
            y = y + b
        return _r(y, pol.autocast_outputs)

    @staticmethod
    def backward(ctx, dy):
        xq, wq = ctx.saved_tensors
        pol = ctx.pol
        dyq = _r(dy, pol.grads)
        d2 = dyq.reshape(-1, dyq.shape[-1])
        dx = dyq @ wq
        if ctx.round_dx:
            dx = _r(dx, pol.grads)
        dw = _r(d2.t() @ xq.reshape(-1, xq.shape[-1]), pol.autocast_outputs)
        db = d2.sum(0) if ctx.has_b else None
        return dx, dw, db, None, None


def linear(x, w, b, pol, round_dx=True):
    return _Linear.apply(x, w, b, pol, round_dx)


class _Round(torch.autograd.Function):
    """r(x) forward, r(dy) backward: a tensor the build stores in bf16 (qkv, the attention context)."""

    @staticmethod
    def forward(ctx, x, pol):
        ctx.pol = pol
        return _r(x, pol.acts)

    @staticmethod
    def backward(ctx, dy):
        return _r(dy, ctx.pol.grads), None


class _Attention(torch.autograd.Function):
    """softmax(q k^T d^-1/2) v per head, operands bf16, scores / softmax / row statistics f32 (HF:181-206, SDPA HF:239-252).
    Backward = the flash-attention identities the build uses: dV = P^T dO, dP = dO V^T, delta = rowsum(dO * O),
    dS = P (dP - delta) d^-1/2, dQ = dS K, dK = dS^T Q, with P (f32, recomputed) rounded where it is an MFMA operand."""

    @staticmethod
    def forward(ctx, q, k, v, pol):
        d = q.shape[-1]
        s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        p = torch.softmax(s, dim=-1)
        o = _r(torch.matmul(_r(p, pol.acts), v), pol.acts)
        ctx.save_for_backward(q, k, v, p, o)
        ctx.pol = pol
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, o = ctx.saved_tensors
        pol = ctx.pol
        d = q.shape[-1]
        do = _r(do, pol.grads)
        dv = _r(torch.matmul(_r(p, pol.acts).transpose(-1, -2), do), pol.grads)
        dp = torch.matmul(do, v.transpose(-1, -2))
        delta = (do * o).sum(-1, keepdim=True)
        ds = _r(p * (dp - delta) * (d ** -0.5), pol.grads)
        dq = _r(torch.matmul(ds, k), pol.grads)
        dk = _r(torch.matmul(ds.transpose(-1, -2), q), pol.grads)
        return dq, dk, dv, None


class _Gelu(torch.autograd.Function):
    """act = r(gelu(pre)); the derivative is SAVED (rounded when Policy.gelu_grad) and the backward product's f32 result is
    multiplied by it before the one rounding of d pre (the build's fc1 / dX-fc2 epilogues)."""

    @staticmethod
    def forward(ctx, pre, pol):
        act = F.gelu(pre)
        cdf = 0.5 * (1.0 + torch.erf(pre * 0.7071067811865476))
        gp = cdf + pre * torch.exp(-0.5 * pre * pre) * 0.3989422804014327
        ctx.save_for_backward(_r(gp, pol.gelu_grad))
        ctx.pol = pol
        return _r(act, pol.acts)

    @staticmethod
    def backward(ctx, dact):
        (gp,) = ctx.saved_tensors
        return _r(dact * gp, ctx.pol.grads), None


def _layer(x, p, prefix, heads, eps, pol, taps, tapname):
    B, N, D = x.shape
    d = D // heads
    a = prefix + "attention.attention."
    h = F.layer_norm(x, (D,), p[prefix + "layernorm_before.weight"], p[prefix + "layernorm_before.bias"], eps)

    def proj(nm):
        y = linear(h, p[a + nm + ".weight"], p[a + nm + ".bias"], pol)
        return _Round.apply(y, pol).view(B, N, heads, d).transpose(1, 2)
    ctxv = _Attention.apply(proj("query"), proj("key"), proj("value"), pol).transpose(1, 2).reshape(B, N, D)
    x = x + linear(ctxv, p[prefix + "attention.output.dense.weight"], p[prefix + "attention.output.dense.bias"], pol)
    h = F.layer_norm(x, (D,), p[prefix + "layernorm_after.weight"], p[prefix + "layernorm_after.bias"], eps)
    pre = linear(h, p[prefix + "intermediate.dense.weight"], p[prefix + "intermediate.dense.bias"], pol)
    act = _Gelu.apply(pre, pol)
    # dX of fc2 is NOT rounded on its own: the build multiplies the f32 accumulator by gelu' and rounds once (_Gelu.backward)
    x = x + linear(act, p[prefix + "output.dense.weight"], p[prefix + "output.dense.bias"], pol, round_dx=False)
    if taps is not None:
        taps[tapname] = x
    return x


def forward(cfg: vo.OracleConfig, p: "Dict[str, torch.Tensor]", pixel_values, bool_masked_pos, pol: Policy = BUILD,
            taps: Optional[dict] = None):
    """`videomae_oracle.forward` with the operand policy `pol`.  Returns (loss, logits, labels)."""
    B, T, C, H, W = pixel_values.shape
    D, Dd, L = cfg.hidden_size, cfg.decoder_hidden_size, cfg.seq_len
    ts, ps = cfg.tubelet_size, cfg.patch_size
    # HF:164-177 as a product over patches (k order c, dt, dy, dx = the Conv3d weight's own layout)
    patches = pixel_values.permute(0, 2, 1, 3, 4).reshape(B, C, T // ts, ts, H // ps, ps, W // ps, ps)
    patches = patches.permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, L, C * ts * ps * ps)
    w = p["videomae.embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
    x = linear(patches, w, p["videomae.embeddings.patch_embeddings.projection.bias"], pol)
    x = x + vo.sinusoid_table(L, D)[None]
    x = x[~bool_masked_pos].reshape(B, -1, D)
    if taps is not None:
        taps["embed"] = x
    for i in range(cfg.num_hidden_layers):
        x = _layer(x, p, f"videomae.encoder.layer.{i}.", cfg.num_attention_heads, cfg.layer_norm_eps, pol, taps, f"enc{i}")
    x = linear(x, p["encoder_to_decoder.weight"], None, pol)
    pos = vo.sinusoid_table(L, Dd)[None].expand(B, -1, -1)
    pos_vis = pos[~bool_masked_pos].reshape(B, -1, Dd)
    pos_msk = pos[bool_masked_pos].reshape(B, -1, Dd)
    x = torch.cat([x + pos_vis, p["mask_token"] + pos_msk], dim=1)
    if taps is not None:
        taps["x_full"] = x
    for i in range(cfg.decoder_num_hidden_layers):
        x = _layer(x, p, f"decoder.decoder_layers.{i}.", cfg.decoder_num_attention_heads, cfg.layer_norm_eps, pol, taps, f"dec{i}")
    n_mask = pos_msk.shape[1]
    x = x[:, -n_mask:]
    x = F.layer_norm(x, (Dd,), p["decoder.norm.weight"], p["decoder.norm.bias"], cfg.decoder_norm_eps)
    logits = linear(x, p["decoder.head.weight"], p["decoder.head.bias"], pol)
    with torch.no_grad():
        labels = vo.pixel_labels(cfg, pixel_values, bool_masked_pos)
    loss = F.mse_loss(logits.float(), labels)
    if taps is not None:
        taps["logits"] = logits
        taps["labels"] = labels
    return loss, logits, labels


def step(cfg: vo.OracleConfig, params, pixel_values, bool_masked_pos, pol: Policy = BUILD, grad_scale: float = 1.0,
         taps: Optional[dict] = None):
    """Forward + backward under `pol`; gradients are f32 (multiplied by `grad_scale`, like `videomae_oracle.step`)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss, _, _ = forward(cfg, p, pixel_values, bool_masked_pos, pol, taps)
    (loss * grad_scale).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    return loss.detach(), grads


def probe_norms(grads):
    """The three grad_logger norms (loggingtools.py:107-116): grad-EFL, grad-ELL, grad-DLL."""
    return [float(grads[k].double().norm()) for k in vo.GRAD_PROBES]
