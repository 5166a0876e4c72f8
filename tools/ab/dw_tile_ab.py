"""Same-process A/B of the weight-gradient group of one layer on 256 x 256 tiles (tile config 10) and on 128 x 384 tiles (12), per
K split, at BVC_BATCH clips: decoder widths (384 / 1536: multiples of 384, not of 256) and encoder widths (768 / 3072: both).
Interleaved rounds, median; algorithmic TFLOP/s = 2 M (2 D I + 4 D D) per launch."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"
Bc = int(os.environ.get("BVC_BATCH", "256"))
rounds = int(os.environ.get("BVC_ROUNDS", "5"))


def time_once(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"tools/ab/dw_tile_ab.py at BVC_BATCH={Bc}, {rounds} interleaved rounds, medians")
for tag, M, D, I, combos in (("dec", Bc * 1568, 384, 1536, [(10, 6), (10, 12), (11, 4), (12, 7), (12, 14), (12, 4), (12, 3)]),
                             ("enc", Bc * 160, 768, 3072, [(10, 2), (10, 4), (12, 2), (12, 1)])):
    dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
    dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
    dqkv = G.bf16_randn(M, 3 * D, seed=11)
    outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
    bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
    flops = 2.0 * M * (D * I * 2 + D * D * 4)

    def mk(split):
        return [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),
                G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),
                G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),
                G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]
    res = {c: [] for c in combos}
    for tile, split in combos:
        G.run_gemm(mk(split), G.TN, tile)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for tile, split in combos:
            ds = mk(split)
            res[(tile, split)].append(time_once(lambda: G.run_gemm(ds, G.TN, tile)))
    for (tile, split), v in res.items():
        us = statistics.median(v)
        print(f"{tag} dW group  tile{tile} split {split:2d}: {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
