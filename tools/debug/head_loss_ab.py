"""head + MSE product (EPI_LOSS, M = B x 1408, N = 1536, K = 384) per tile config, BVC_BATCH clips."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G
dev = "cuda"
B = int(os.environ.get("BVC_BATCH", "256"))
M, N, K = B * 1408, 1536, 384
A, W = G.bf16_randn(M, K, seed=1), G.bf16_randn(N, K, seed=2, scale=0.02)
bias, labels = torch.zeros(N, device=dev), torch.randn(M, N, device=dev)
diff = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
part = torch.zeros(1 << 17, device=dev)
d = G.gemm_desc(A, W, M, N, K, G.EPI["LOSS"], diff, bias=bias, labels=labels, partial=part)
def t(tile, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): G.run_gemm([d], G.NT, tile)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tiles = [-1, 0, 10, 11]
res = {k: [] for k in tiles}
for k in tiles: G.run_gemm([d], G.NT, k)
torch.cuda.synchronize()
for _ in range(5):
    for k in tiles: res[k].append(t(k))
for k in tiles:
    us = statistics.median(res[k]); print(f"head+MSE B={B} tile {k:3d}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF {(2*(M*K+N*K)+6.0*M*N)/us/1e3:7.1f} GB/s")
