"""A few launches of the decoder-sized products (B=64) through the product path (persistent kernel) and through the per-tile
kernel, for rocprofv3 --pmc passes (tools/gpu_check.sh pmc_gemm)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G
M = 64 * 1568
for N, K, layout, epi in ((1152, 384, G.NT, "BF16"), (1536, 384, G.NT, "GELU"), (384, 1536, G.NN, "BF16")):
    A = G.bf16_randn(M, K)
    B = G.bf16_randn(N, K) if layout == G.NT else G.bf16_randn(K, N)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(C2=torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), bias=torch.zeros(N, device="cuda")) if epi == "GELU" else {}
    d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **kw)
    for env in (None, "1"):
        if env:
            os.environ["BVC_GEMM_NO_PERSIST"] = "1"
        for _ in range(3):
            G.run_gemm([d], layout, -1, -1)
        os.environ.pop("BVC_GEMM_NO_PERSIST", None)
torch.cuda.synchronize()
