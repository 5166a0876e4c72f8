"""Decoder-layer weight gradients (D = 384: not a multiple of 256) at BVC_BATCH clips: the four products as ONE grouped launch on
256 x 256 tiles (what plan_dw picks today) against TWO launches - the products whose output width is D on 256 x 128 tiles (384 = 3 x 128,
no padded column tile), the other two on 256 x 256."""
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
M, D, I = Bc * 1568, 384, 1536
dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
dqkv = G.bf16_randn(M, 3 * D, seed=11)
outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
flops = 2.0 * M * (D * I * 2 + D * D * 4)


def descs(split):
    return [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),     # fc2  [D x I]
            G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),     # fc1  [I x D]
            G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),     # proj [D x D]
            G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]  # qkv [3D x D]


res = {}
for rnd in range(3):
    one = descs(6)
    res.setdefault("one launch, 256x256 split 6", []).append(t(lambda: G.run_gemm(one, G.TN, 10)))
    for s1, s2 in ((7, 14), (6, 12), (7, 7)):
        a, b = descs(s1), descs(s2)
        g1, g2 = [a[1], a[3]], [b[0], b[2]]

        def two():
            G.run_gemm(g1, G.TN, 11)
            G.run_gemm(g2, G.TN, 10)
        res.setdefault(f"two launches: fc1+qkv 256x128 split {s1}, fc2+proj 256x256 split {s2}", []).append(t(two))
        g2b = [b[0], b[2]]

        def two11():
            G.run_gemm(g1, G.TN, 11)
            G.run_gemm(g2b, G.TN, 11)
        res.setdefault(f"two launches, both 256x128: split {s1} / {s2}", []).append(t(two11))
for k, v in res.items():
    m = statistics.median(v)
    print(f"dec dW layer, {Bc} clips: {k:78s} {m:8.1f} us {flops / m / 1e6:7.1f} TF", flush=True)
