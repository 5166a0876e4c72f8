"""CPU oracle for the SimCLR rows (projection head + loss)  --  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32 restatement of pretraining/contrastive/pretrain_simclr.py:
  :71-77   _adapt_model_simclr   fc = Linear(p, p) -> ReLU -> Linear(p, p)
  :86-91   get_special_matrix    tridiagonal 0/1 matrix
  :114-128 info_nce_loss         cosine similarity / T; logsumexp over every negative entry of the matrix (one scalar,
                                 boolean indexing flattens) minus the mean of the positive entries
  :284-292 masks                 pos = tridiagonal, neg = everything else off the diagonal
Pinned by tests/golden/simclr_*.json, written by oracle/make_golden.py from the reference's own functions (imported in the
build container).  The torchvision trunk of the reference is absent offline: parity for it is unpinned.
"""
import numpy as np
import torch
import torch.nn.functional as F


def get_special_matrix(n):
    return np.asarray([[1 if i == j + 1 or i == j - 1 else 0 for j in range(n)] for i in range(n)])


def make_masks(batch_size):
    n = batch_size * 2
    self_mask = torch.eye(n, dtype=torch.bool)
    pos_mask = torch.tensor(get_special_matrix(n), dtype=torch.bool)
    neg_mask = torch.ones_like(pos_mask)
    neg_mask[pos_mask | self_mask] = False
    return pos_mask, neg_mask


def info_nce_loss(temperature, masks, feats):
    cos_sim = F.cosine_similarity(feats[:, None, :], feats[None, :, :], dim=-1) / temperature
    pos_mask, neg_mask = masks
    pos_part = -cos_sim[pos_mask]
    neg_part = torch.logsumexp(cos_sim[neg_mask], dim=-1)
    return (neg_part + pos_part).mean()


def info_nce_loss_lowmem(temperature, batch_size, feats, eps=1e-8):
    """The same loss without the (n, n, p) broadcast product of F.cosine_similarity (550 GB at 8192 rows x 2048): rows
    normalised as cosine_similarity does (each norm clamped at eps), one (n, n) product, the reference's masks applied by
    index arithmetic.  Checked against info_nce_loss (and through it against the reference's function) up to 512 rows by
    oracle/make_golden.py and tests/test_simclr_oracle.py."""
    n = 2 * batch_size
    fn = feats / feats.norm(dim=1, keepdim=True).clamp_min(eps)
    cos = fn @ fn.t() / temperature
    i = torch.arange(n)
    d = (i[:, None] - i[None, :]).abs()
    pos = cos[d == 1]
    neg = cos[d > 1]
    return torch.logsumexp(neg, dim=-1) - pos.mean()


def head_forward_bf16_operands(x, w1, b1, w2, b2):
    """head_forward with every GEMM operand rounded to bf16 (x, both weights, the hidden activation) and f32 accumulation: what
    a bf16-MFMA implementation computes up to summation order.  The ReLU gates are then decided by the same numbers the
    device sees, which separates operand rounding from gate flips in the gradient comparison (tests/test_gpu_simclr.py)."""
    r = lambda t: t.to(torch.bfloat16).to(torch.float32)   # noqa: E731
    h = F.relu(F.linear(r(x), r(w1), b1))
    return F.linear(r(h), r(w2), b2)


def head_forward(x, w1, b1, w2, b2):
    return F.linear(F.relu(F.linear(x, w1, b1)), w2, b2)


def synthetic_features(n, p, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, p, generator=g)


def head_params(p_in, p_out, seed):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / np.sqrt(p_in)
    return {"0.weight": (torch.rand(p_out, p_in, generator=g) * 2 - 1) * k, "0.bias": (torch.rand(p_out, generator=g) * 2 - 1) * k,
            "2.weight": (torch.rand(p_out, p_out, generator=g) * 2 - 1) / np.sqrt(p_out),
            "2.bias": (torch.rand(p_out, generator=g) * 2 - 1) / np.sqrt(p_out)}
