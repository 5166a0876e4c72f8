"""SimCLR projection head and loss with the reference's interface (pretraining/contrastive/pretrain_simclr.py).

  :71-77   _adapt_model_simclr : model.fc = Linear(p, p) -> ReLU -> Linear(p, p)     -> ProjectionHead (keys fc.0.*, fc.2.*)
  :86-91   get_special_matrix  : tridiagonal positive mask                             -> same
  :114-128 info_nce_loss       : cos-sim / T, logsumexp over ALL negatives - mean(pos) -> same signature, HIP kernels
  :284-292 mask construction                                                           -> make_masks
The trunk (torchvision ResNet in the reference) is not part of this path; any module producing (2B, p) features works.
The (2B x 2B) similarity matrix is never materialised in f32: it is a bf16 MFMA GEMM of the L2-normalised rows whose
epilogue reduces the loss; backward re-runs the product to emit bf16 d loss / d sim and multiplies it back.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, _ops


def get_special_matrix(n):
    x = np.zeros((n, n), dtype=np.int64)
    i = np.arange(n - 1)
    x[i, i + 1] = 1
    x[i + 1, i] = 1
    return x


def make_masks(batch_size, device):
    """(pos_mask, neg_mask) exactly as pretrain_simclr.py:284-292 builds them (mask_size = 2 * batch_size)."""
    n = batch_size * 2
    self_mask = torch.eye(n, dtype=torch.bool, device=device)
    pos_mask = torch.tensor(get_special_matrix(n), dtype=torch.bool, device=device)
    neg_mask = torch.ones_like(pos_mask)
    neg_mask[pos_mask | self_mask] = False
    return pos_mask, neg_mask


_checked_masks = {}


def _check_masks(masks, n):
    key = (id(masks[0]), id(masks[1]), n)
    if key in _checked_masks:
        return
    pos, neg = make_masks(n // 2, masks[0].device)
    if masks[0].shape != (n, n) or not (torch.equal(masks[0], pos) and torch.equal(masks[1], neg)):
        raise NotImplementedError("info_nce_loss: only the reference's masks (tridiagonal positives, all other "
                                  "off-diagonal entries negative) are implemented")
    _checked_masks[key] = True


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, temperature):
        n, p = feats.shape
        if n % 8 or p % 64:
            raise ValueError("info_nce_loss: needs 2*batch % 8 == 0 and feature width % 64 == 0")
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        f = feats.detach().float().contiguous()
        fn = torch.empty((n, p), dtype=torch.bfloat16, device=f.device)
        inv = torch.empty(n, dtype=torch.float32, device=f.device)
        _lib.check(L.bvc_op_row_normalize(f.data_ptr(), fn.data_ptr(), inv.data_ptr(), n, p, 1e-8, st), "row_normalize")
        d = _ops.gemm_desc(fn, fn, n, n, p, _ops.EPI["NCE"], None, ldc=n, alpha=1.0 / temperature)
        nt = _ops.num_tiles(d)
        partial = torch.empty(2 * nt, dtype=torch.float32, device=f.device)
        d.partial = partial.data_ptr()
        _ops.gemm(d, _ops.NT)
        loss = torch.empty((), dtype=torch.float32, device=f.device)
        stats = torch.empty(2, dtype=torch.float32, device=f.device)
        npos = 2 * (n - 1)
        _lib.check(L.bvc_op_nce_finalize(partial.data_ptr(), nt, 1.0 / temperature, npos, loss.data_ptr(), stats.data_ptr(), st),
                   "nce_finalize")
        ctx.save_for_backward(f, fn, inv, stats)
        ctx.temperature = temperature
        return loss

    @staticmethod
    def backward(ctx, gout):
        f, fn, inv, stats = ctx.saved_tensors
        n, p = f.shape
        T = ctx.temperature
        ldp = (n + 63) // 64 * 64                      # contraction length of the second product, zero padded
        P = torch.zeros((n, ldp), dtype=torch.bfloat16, device=f.device)
        _ops.gemm(_ops.gemm_desc(fn, fn, n, n, p, _ops.EPI["NCE_BWD"], P, ldc=ldp, alpha=1.0 / T, labels=stats), _ops.NT)
        # d loss / d fn = (2 / T) * P fn   (P is symmetric); the upstream gradient rides in as a device scalar.
        # fn enters this product as hi + lo (two bf16 terms, ~16 mantissa bits): the normalisation backward below keeps only the
        # component of d fn ORTHOGONAL to fn, and for the strongly correlated rows a trunk produces (cosines of 0.99) that is a
        # small difference of large terms - with fn rounded to one bf16 term the feature gradient was 1.7e-2 ... 3.4e-2 off the f32
        # result (round 3's "S3 d features" = 2.0e-2), with the split 1e-3; rounding P itself accounts for ~1e-3
        # (tools/debug/nce_grad_error_sources.py; the second pass costs one more product of the loss's own size).
        g = gout.detach().float().contiguous()
        dfn = torch.empty((n, p), dtype=torch.float32, device=f.device)
        _ops.gemm(_ops.gemm_desc(P, fn, n, p, ldp, _ops.EPI["F32"], dfn, alpha=2.0 / T, alpha_dev=g), _ops.NN)
        fn_lo = _ops.cast_bf16(f * inv.unsqueeze(1) - fn.float())
        _ops.gemm(_ops.gemm_desc(P, fn_lo, n, p, ldp, _ops.EPI["RESID"], dfn, alpha=2.0 / T, alpha_dev=g, resid=dfn), _ops.NN)
        df = torch.empty_like(dfn)
        _lib.check(_lib.lib().bvc_op_row_normalize_bwd(f.data_ptr(), inv.data_ptr(), dfn.data_ptr(), df.data_ptr(), n, p,
                                                       _lib.current_stream_ptr()), "row_normalize_bwd")
        return df, None


def info_nce_loss(temperature, masks, feats, mode='train'):
    """Same call as the reference: criterion = partial(info_nce_loss, temperature, masks); loss = criterion(pred)."""
    if not feats.is_cuda:
        raise _lib.BvcError("info_nce_loss runs on a GPU only (libbvc_hip.so has no CPU path)")
    _check_masks(masks, feats.shape[0])
    return _InfoNCE.apply(feats, float(temperature)).to(feats.dtype if feats.dtype == torch.float32 else torch.float32)


class _Head(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        n, pin = x.shape
        pout = w1.shape[0]
        xb, w1b, w2b = _ops.cast_bf16(x.detach()), _ops.cast_bf16(w1.detach()), _ops.cast_bf16(w2.detach())
        h = torch.empty((n, pout), dtype=torch.bfloat16, device=x.device)
        _ops.gemm(_ops.gemm_desc(xb, w1b, n, pout, pin, _ops.EPI["RELU"], h, bias=b1.detach().float()), _ops.NT)
        out = torch.empty((n, pout), dtype=torch.float32, device=x.device)
        _ops.gemm(_ops.gemm_desc(h, w2b, n, pout, pout, _ops.EPI["F32"], out, bias=b2.detach().float()), _ops.NT)
        ctx.save_for_backward(xb, w1b, w2b, h)
        ctx.x_dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        xb, w1b, w2b, h = ctx.saved_tensors
        n, pin = xb.shape
        pout = w1b.shape[0]
        dev = xb.device
        dob = _ops.cast_bf16(dout)
        dw2 = torch.empty((pout, pout), dtype=torch.float32, device=dev)
        db2 = torch.zeros(pout, dtype=torch.float32, device=dev)
        _ops.gemm(_ops.gemm_desc(dob, h, pout, pout, n, _ops.EPI["F32"], dw2, rowsum=db2), _ops.TN)
        dh = torch.empty((n, pout), dtype=torch.bfloat16, device=dev)
        _ops.gemm(_ops.gemm_desc(dob, w2b, n, pout, pout, _ops.EPI["DRELU"], dh, aux=h), _ops.NN)
        dw1 = torch.empty((pout, pin), dtype=torch.float32, device=dev)
        db1 = torch.zeros(pout, dtype=torch.float32, device=dev)
        _ops.gemm(_ops.gemm_desc(dh, xb, pout, pin, n, _ops.EPI["F32"], dw1, rowsum=db1), _ops.TN)
        dx = torch.empty((n, pin), dtype=torch.float32, device=dev)
        _ops.gemm(_ops.gemm_desc(dh, w1b, n, pin, pout, _ops.EPI["F32"], dx), _ops.NN)
        return dx.to(ctx.x_dtype), dw1, db1, dw2, db2


class ProjectionHead(nn.Module):
    """Linear(n_features, n_out) -> ReLU -> Linear(n_out, n_out) with nn.Sequential's parameter names (0.*, 2.*)."""

    def __init__(self, n_features, n_out):
        super().__init__()
        ref = nn.Sequential(nn.Linear(n_features, n_out), nn.ReLU(), nn.Linear(n_out, n_out))   # torch's default init
        for idx in ("0", "2"):
            holder = nn.Module()
            holder.weight = nn.Parameter(ref[int(idx)].weight.detach().clone())
            holder.bias = nn.Parameter(ref[int(idx)].bias.detach().clone())
            self.add_module(idx, holder)

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.BvcError("ProjectionHead runs on a GPU only (libbvc_hip.so has no CPU path)")
        if x.shape[-1] % 64 or self._modules["0"].weight.shape[0] % 64:
            raise ValueError("ProjectionHead: feature widths must be multiples of 64")
        m0, m2 = self._modules["0"], self._modules["2"]
        return _Head.apply(x.reshape(-1, x.shape[-1]), m0.weight, m0.bias, m2.weight, m2.bias)


def _adapt_model_simclr(model, n_features, n_out):
    model.fc = ProjectionHead(n_features, n_out)
    _ = model.float()
    return model


class SimCLRViT(nn.Module):
    """BASELINE config 5 ("SimCLR ViT-B, global batch 4096, embedding all-gather"), assembled from reference parts as
    SURVEY §8 states, because the reference's own get_model only adapts torchvision ResNets (pretrain_simclr.py:71-84):
    trunk = the reference's video ViT with one frame (pretraining/predictive/vision_transformer.py VisionTransformer,
    num_frames=1, tubelet 1), features = mean over tokens, ``fc`` = the SimCLR head of pretrain_simclr.py:71-77.
    ``forward(x)`` takes the (2B, C, H, W) view the reference's forward_loss builds (:322-324) and returns (2B, p)."""

    # the head's ordinary parameters sit BEHIND the flat trunk: autograd has accumulated their gradients when the trunk's backward
    # starts, so the data-parallel wrapper sends them under it (ddp.py)
    _bvc_loose_before_flat = True

    def __init__(self, model_name="vit_base", image_size=224, patch_size=16, pred_emb_dim=None):
        super().__init__()
        from . import jepa
        self.trunk = jepa.__dict__[model_name](img_size=[image_size], patch_size=patch_size, num_frames=1, tubelet_size=1)
        d = self.trunk.embed_dim
        self.fc = ProjectionHead(d, pred_emb_dim or d)

    def forward(self, x):
        from . import jepa
        tokens = self.trunk(x.unsqueeze(1))                      # (2B, N, D)
        return self.fc(jepa.token_mean(tokens))


def global_info_nce_loss(temperature, masks, feats_local):
    """info_nce_loss over the rows of EVERY rank: AllGather (pretraining/predictive/distributed.py:49-76: forward all_gather + cat,
    backward all_reduce then own slice) in front of the reference's loss.  `masks` are make_masks(global_batch)."""
    from .distributed import AllGather
    return info_nce_loss(temperature, masks, AllGather.apply(feats_local))
