"""Same-process A/B of the A-stationary kernel (gemm_as.hip, tile configs 15 - 18) against the 256 x 256 persistent kernel (tile config 10) on the
K = 384 products of the VideoMAE decoder.  Usage: python tools/ab/as_ab.py [clips] [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_util as G   # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
M, K = B * 1568, 384


def ev_time(fn, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"{B} clips: M = {M}, K = {K}; us per launch, median [min .. max] over {ROUNDS} interleaved rounds (A and A' = the same launch twice)")
print(f"{'product':16s} {'A: gemm8 (tile 10)':>24s} {'A-prime':>24s} {'15: ovl, early DMA':>24s} {'16: no ovl, early':>24s} {'17: ovl, late DMA':>24s} {'18: no ovl, late':>24s}  15/A  16/A  17/A  18/A")
for name, N, epi in (("dec qkv", 1152, "BF16"), ("dec fc1+GELU", 1536, "GELU")):
    A = G.bf16_randn(M, K, seed=1)
    W = G.bf16_randn(N, K, scale=0.05, seed=2)
    bias = torch.zeros(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == "GELU" else None
    d = G.gemm_desc(A, W, M, N, K, G.EPI[epi], C, bias=bias, C2=C2)
    kname = G.bvc._ops.gemm_kernel_name(d, G.NT)
    fns = [lambda: G.run_gemm([d], G.NT, tile_cfg=10), lambda: G.run_gemm([d], G.NT, tile_cfg=10), lambda: G.run_gemm([d], G.NT, tile_cfg=15),
           lambda: G.run_gemm([d], G.NT, tile_cfg=16), lambda: G.run_gemm([d], G.NT, tile_cfg=17), lambda: G.run_gemm([d], G.NT, tile_cfg=18)]
    for f in fns:
        ev_time(f, 2)
    t = [[] for _ in fns]
    for r in range(ROUNDS):
        for i in (0, 2, 3, 4, 5, 1, 1, 5, 4, 3, 2, 0):
            t[i].append(ev_time(fns[i]))

    def fmt(a):
        a = np.array(a)
        return f"{np.median(a):8.1f} [{a.min():6.1f} ..{a.max():7.1f}]"
    m = [np.median(x) for x in t]
    print(f"{name:16s} " + " ".join(f"{fmt(x):>24s}" for x in t) + "  " + " ".join(f"{m[i] / m[0]:5.3f}" for i in (2, 3, 4, 5)), flush=True)
    del A, W, C, C2
    torch.cuda.empty_cache()
