"""Experiments build: unit order of a layer's grouped weight-gradient launch - tiles row-major inside a K split (default) vs the shorter
side fastest (BVC_G8_TN_SHORT_FAST=1), with the plan the step uses (plan_dw).  Usage: python tools/ab/dw_order_ab.py [clips]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_util as G  # noqa: E402

dev = "cuda"
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def t(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for tag, M, D, I in (("enc", Bc * 160, 768, 3072), ("dec", Bc * 1568, 384, 1536)):
    dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
    dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
    dqkv = G.bf16_randn(M, 3 * D, seed=11)
    outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
    bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
    flops = 2.0 * M * (D * I * 2 + D * D * 4)
    ds = [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0]), G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1]),
          G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2]), G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3])]
    tile, split = G.bvc._ops.plan_dw(ds)
    res = {}
    for rnd in range(5):
        for walk in ("row-major", "short-fast", "short-fast", "row-major"):
            if walk == "short-fast":
                os.environ["BVC_G8_TN_SHORT_FAST"] = "1"
            else:
                os.environ.pop("BVC_G8_TN_SHORT_FAST", None)
            G.run_gemm(ds, G.TN, tile)
            torch.cuda.synchronize()
            res.setdefault(walk, []).append(t(lambda: G.run_gemm(ds, G.TN, tile)))
    os.environ.pop("BVC_G8_TN_SHORT_FAST", None)
    print(f"{tag} dW group, {Bc} clips, tile config {tile}, split {split}: " +
          " | ".join(f"{w} {statistics.median(v):8.1f} us [{min(v):7.1f} .. {max(v):7.1f}] {flops / statistics.median(v) / 1e6:7.1f} TF" for w, v in res.items()), flush=True)
