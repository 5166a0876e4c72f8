"""Experiments build (tools/gpu_check.sh exp): the grouped weight-gradient launch of one layer on the 256-row persistent kernel with the
K splits slowest (default) vs fastest (BVC_G8_TN_LEGACY_WALK=1) in the unit order, at BVC_BATCH clips."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"


def t(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


Bc = int(os.environ.get("BVC_BATCH", "256"))
for tag, M, D, I, splits in (("enc", Bc * 160, 768, 3072, (2, 4)), ("dec", Bc * 1568, 384, 1536, (6, 12))):
    dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
    dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
    dqkv = G.bf16_randn(M, 3 * D, seed=11)
    outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
    bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
    flops = 2.0 * M * (D * I * 2 + D * D * 4)
    for split in splits:
        ds = [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),
              G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),
              G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),
              G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]
        res = {}
        for rnd in range(3):
            for walk in ("split-slowest", "split-fastest"):
                if walk == "split-fastest":
                    os.environ["BVC_G8_TN_LEGACY_WALK"] = "1"
                else:
                    os.environ.pop("BVC_G8_TN_LEGACY_WALK", None)
                G.run_gemm(ds, G.TN, 10)
                torch.cuda.synchronize()
                res.setdefault(walk, []).append(t(lambda: G.run_gemm(ds, G.TN, 10)))
        os.environ.pop("BVC_G8_TN_LEGACY_WALK", None)
        print(f"{tag} dW group, {Bc} clips, 256x256 tiles, split {split}: " +
              " | ".join(f"{w} {statistics.median(v):8.1f} us {flops / statistics.median(v) / 1e6:7.1f} TF" for w, v in res.items()), flush=True)
