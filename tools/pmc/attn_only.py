"""Runs the decoder-shape attention kernels a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G
L = G.L
B, N, H, HD = int(os.environ.get("BVC_BATCH", "64")), 1568, 6, 64
D = HD * H
qkv = G.bf16_randn(B * N, 3 * D)
ctx = torch.zeros(B * N, D, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B * H, N, device="cuda")
dctx = G.bf16_randn(B * N, D, seed=2)
dqkv = torch.zeros_like(qkv)
delta = torch.zeros(B * H, N, device="cuda")
for _ in range(3):
    L.check(L.lib().bvc_op_attention_fwd(G.ptr(qkv), G.ptr(ctx), G.ptr(lse), B, N, H, HD, G.stream()))
    L.check(L.lib().bvc_op_attention_bwd(G.ptr(qkv), G.ptr(ctx), G.ptr(dctx), G.ptr(lse), G.ptr(delta), G.ptr(dqkv), B, N, H, HD, G.stream()))
torch.cuda.synchronize()
