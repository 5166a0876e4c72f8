"""K-sweep / epilogue probe: separates per-K-step cost from per-tile fixed cost for the decoder-shaped GEMM."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_util as G
from tools.microbench import timeit
dev = "cuda"
M, N = 25088, 1152
for tile in (0, 1):
    for epi, dt in (("BF16", torch.bfloat16), ("F32", torch.float32)):
        line = []
        for K in (64, 128, 384, 768, 1536):
            A, B = G.bf16_randn(M, K), G.bf16_randn(N, K)
            C = torch.zeros(M, N, device=dev, dtype=dt)
            d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C)
            ms = timeit(lambda: G.run_gemm([d], G.NT, tile, 2))
            line.append(f"K={K}: {ms*1e3:6.1f}us")
        print(f"tile{tile} {epi:5s} " + "  ".join(line), flush=True)
# GELU epilogue cost
K = 384
A, B = G.bf16_randn(M, K), G.bf16_randn(1536, K)
bias = torch.zeros(1536, device=dev)
pre = torch.zeros(M, 1536, device=dev, dtype=torch.bfloat16); act = torch.zeros_like(pre)
d1 = G.gemm_desc(A, B, M, 1536, K, G.EPI["BF16"], pre, bias=bias)
d2 = G.gemm_desc(A, B, M, 1536, K, G.EPI["GELU"], pre, C2=act, bias=bias)
print("fc1 BF16 %.1fus  GELU %.1fus" % (timeit(lambda: G.run_gemm([d1], G.NT, 0, 2))*1e3, timeit(lambda: G.run_gemm([d2], G.NT, 0, 2))*1e3))
# pure copy bandwidth reference
x = torch.empty(64 * 1024 * 1024, device=dev, dtype=torch.float32); y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x)); print("copy 256MB->256MB: %.1fus = %.2f TB/s" % (ms*1e3, 2*x.numel()*4/ms/1e9))
