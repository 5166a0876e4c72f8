"""Time vs K for the decoder-shaped GEMMs (fixed M, N): T(K) = a + b K separates the per-tile fixed cost (prologue latency,
epilogue, launch) from the main-loop rate.  Same process, one MI355X.  -> gpurun_out/ksweep.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = "cuda"
lines = []


def emit(s):
    print(s, flush=True)
    lines.append(s)


def run(M, N, K, layout, epi, tile):
    if layout == G.NT:
        A, B = G.bf16_randn(M, K), G.bf16_randn(N, K)
    else:
        A, B = G.bf16_randn(M, K), G.bf16_randn(K, N)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    kw = {}
    if epi == "GELU":
        kw = dict(C2=torch.zeros(M, N, device=dev, dtype=torch.bfloat16), bias=torch.zeros(N, device=dev))
    if epi == "DGELU":
        kw = dict(aux=G.bf16_randn(M, N, seed=3))
    d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **kw)
    if tile >= 3 and K == 384:      # correctness of the experimental kernel against the 128x128 one (same operands)
        C0 = torch.zeros_like(C)
        G.run_gemm([G.gemm_desc(A, B, M, N, K, G.EPI[epi], C0, **kw)], layout, 0, -1)
        G.run_gemm([d], layout, tile, -1)
        torch.cuda.synchronize()
        emit(f"   tile{tile} vs tile0 at K={K}: max abs diff {float((C.float() - C0.float()).abs().max()):.3e}, ref max {float(C0.float().abs().max()):.2f}")
    return timeit(lambda: G.run_gemm([d], layout, tile, -1), iters=10, warm=2)


Bc = int(os.environ.get("BVC_BATCH", "64"))
M = Bc * 1568
for N, layout, epi in ((1152, G.NT, "BF16"), (1536, G.NT, "GELU"), (1536, G.NN, "DGELU"), (384, G.NT, "BF16")):
    for tile in ((0, 6, 9) if N != 384 else (0, 6, 9)):
        row = []
        for K in ((128, 256, 384, 768, 1536) if tile in (6, 9) else (64, 128, 256, 384, 768, 1536)):
            ms = run(M, N, K, layout, epi, tile)
            row.append((K, ms))
        (k0, t0), (k1, t1) = row[0], row[-1]
        slope = (t1 - t0) / (k1 - k0)
        icpt = t0 - slope * k0
        tiles = -(-M // (256 if tile in (3, 8) else 64 if tile == 2 else 128)) * -(-N // (64 if tile in (1, 2) else 128))
        emit(f"M={M} N={N} {['NT','NN','TN'][layout]} {epi:5s} tile{tile}: " + " ".join(f"K{k}={ms*1e3:.0f}us" for k, ms in row) +
             f" | intercept {icpt*1e3:.0f}us slope {2.0*M*N/slope/1e9:.0f}TF marginal | {tiles} tiles")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "ksweep.txt"), "w").write("\n".join(lines) + "\n")
