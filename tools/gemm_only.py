"""Runs one GEMM per layout/tile a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_util as G
M, N, K = 2560, 3072, 768
for layout in (G.NT, G.NN, G.TN):
    for tile in (0, 2):
        if layout == G.NT: A, B = G.bf16_randn(M, K), G.bf16_randn(N, K)
        elif layout == G.NN: A, B = G.bf16_randn(M, K), G.bf16_randn(K, N)
        else: A, B = G.bf16_randn(K, M), G.bf16_randn(K, N)
        C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        d = G.gemm_desc(A, B, M, N, K, G.EPI["BF16"], C)
        for _ in range(3):
            G.run_gemm([d], layout, tile, 2)
torch.cuda.synchronize()
