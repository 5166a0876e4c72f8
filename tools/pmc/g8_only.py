"""A handful of launches of the 256-row persistent GEMM for rocprofv3 --pmc passes (tools/gpu_check.sh pmc_g8)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"
cases = [("square8192", G.NT, 8192, 8192, 8192, 10), ("square8192_bn128", G.NT, 8192, 8192, 8192, 11), ("dec_qkv", G.NT, 64 * 1568, 1152, 384, 10),
         ("enc_fc1", G.NT, 64 * 160, 3072, 768, 10), ("square8192_128x128", G.NT, 8192, 8192, 8192, 0),
         ("dec_qkv256", G.NT, 256 * 1568, 1152, 384, 10), ("enc_qkv256", G.NT, 256 * 160, 2304, 768, 10), ("enc_dxfc1_256", G.NN, 256 * 160, 768, 3072, 10)]
only = os.environ.get("BVC_G8_CASE")
for name, lay, M, N, K, tile in cases:
    if only and name != only:
        continue
    A, B = G.bf16_randn(M, K), (G.bf16_randn(N, K, seed=1) if lay == G.NT else G.bf16_randn(K, N, seed=1))
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    d = G.gemm_desc(A, B, M, N, K, G.EPI["BF16"], C)
    for _ in range(4):
        G.run_gemm([d], lay, tile)
    torch.cuda.synchronize()
    print(name, "done", flush=True)
