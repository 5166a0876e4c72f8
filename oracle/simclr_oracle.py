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
