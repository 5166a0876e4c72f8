"""Experiments build: attention kernels with 256-row blocks (eight waves; BVC_ATTN_NW8=1) against the 128-row product form, same process,
interleaved rounds.  A block stages every K / V (or Q / dO) tile of its head: with twice the rows per block the L2 -> LDS fill per query halves.
Outputs are compared first (the same arithmetic per row: equal bit for bit except where the ragged-last-block split differs)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

L = G.L
dev = "cuda"


def t(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def setnw(on):
    if on:
        os.environ["BVC_ATTN_NW8"] = "1"
    else:
        os.environ.pop("BVC_ATTN_NW8", None)


Bc = int(os.environ.get("BVC_BATCH", "256"))
rounds = int(os.environ.get("BVC_ROUNDS", "5"))
print(f"tools/ab/attn_nw_ab.py at BVC_BATCH={Bc}, {rounds} interleaved rounds, median [min-max] us")
for (B, N, H, HD) in [(Bc, 1568, 6, 64), (Bc, 1536, 6, 64), (Bc, 160, 12, 64), (Bc, 196, 12, 64)]:
    D = HD * H
    qkv = G.bf16_randn(B * N, 3 * D)
    dctx = G.bf16_randn(B * N, D, seed=2)
    outs = {}
    for on in (False, True):
        setnw(on)
        ctx = torch.zeros(B * N, D, device=dev, dtype=torch.bfloat16)
        lse = torch.zeros(B * H, N, device=dev)
        dqkv = torch.zeros_like(qkv)
        delta = torch.zeros(B * H, N, device=dev)
        L.check(L.lib().bvc_op_attention_fwd(G.ptr(qkv), G.ptr(ctx), G.ptr(lse), B, N, H, HD, G.stream()))
        L.check(L.lib().bvc_op_attention_bwd(G.ptr(qkv), G.ptr(ctx), G.ptr(dctx), G.ptr(lse), G.ptr(delta), G.ptr(dqkv), B, N, H, HD, G.stream()))
        torch.cuda.synchronize()
        outs[on] = (ctx, lse, dqkv)
    for name, a, b in zip(("ctx", "lse", "dqkv"), outs[False], outs[True]):
        d = (a.float() - b.float()).abs().max().item()
        print(f"  N={N}: {name}: max |128-row - 256-row| = {d:.3e} (max |value| {a.float().abs().max().item():.3e})")
    ctx, lse, dqkv = outs[False]
    delta = torch.zeros(B * H, N, device=dev)
    f = lambda: L.check(L.lib().bvc_op_attention_fwd(G.ptr(qkv), G.ptr(ctx), G.ptr(lse), B, N, H, HD, G.stream()))
    b = lambda: L.check(L.lib().bvc_op_attention_bwd(G.ptr(qkv), G.ptr(ctx), G.ptr(dctx), G.ptr(lse), G.ptr(delta), G.ptr(dqkv), B, N, H, HD, G.stream()))
    res = {(k, on): [] for k in "fb" for on in (False, True)}
    for _ in range(rounds):
        for on in (False, True):
            setnw(on)
            f(); b()
            res[("f", on)].append(t(f))
            res[("b", on)].append(t(b))
    setnw(False)
    flops = 4.0 * B * H * N * N * HD
    for k, nm, mult in (("f", "fwd", 1.0), ("b", "bwd", 2.5)):
        parts = []
        for on in (False, True):
            v = res[(k, on)]
            m = statistics.median(v)
            parts.append(f"{'256-row' if on else '128-row'} {m:8.1f} [{min(v):7.1f}-{max(v):7.1f}] {mult * flops / m / 1e6:6.1f} TF")
        print(f"attn B{B} N{N} H{H} d{HD} {nm}: " + " | ".join(parts), flush=True)
