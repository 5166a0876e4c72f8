"""Attention forward / backward at the step's shapes (BVC_BATCH clips): microseconds per launch, HIP events on the launch stream.
Run once per library build (BVC_LIB_PATH selects another build) on the same box for an A/B."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

L = G.L
dev = "cuda"


def t(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


Bc = int(os.environ.get("BVC_BATCH", "64"))
for (B, N, H, HD) in [(Bc, 160, 12, 64), (Bc, 1568, 6, 64), (4 * 16, 125, 16, 32)]:
    D = HD * H
    qkv = G.bf16_randn(B * N, 3 * D)
    ctx = torch.zeros(B * N, D, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B * H, N, device=dev)
    dctx = G.bf16_randn(B * N, D, seed=2)
    dqkv = torch.zeros_like(qkv)
    delta = torch.zeros(B * H, N, device=dev)
    f = lambda: L.check(L.lib().bvc_op_attention_fwd(G.ptr(qkv), G.ptr(ctx), G.ptr(lse), B, N, H, HD, G.stream()))
    b = lambda: L.check(L.lib().bvc_op_attention_bwd(G.ptr(qkv), G.ptr(ctx), G.ptr(dctx), G.ptr(lse), G.ptr(delta), G.ptr(dqkv), B, N, H, HD, G.stream()))
    f(); b(); torch.cuda.synchronize()
    tf = statistics.median([t(f) for _ in range(5)])
    tb = statistics.median([t(b) for _ in range(5)])
    flops = 4.0 * B * H * N * N * HD
    print(f"attn B{B} N{N} H{H} d{HD}: fwd {tf:8.1f} us {flops / tf / 1e6:7.1f} TF | bwd {tb:8.1f} us {2.5 * flops / tb / 1e6:7.1f} TF (5 products)", flush=True)
