"""GELU epilogue of the persistent GEMM against an f64 reference: C2 = gelu(pre), C = gelu'(pre) with pre = A W^T + bias (bf16 operands,
exact-erf GELU, HF:299-308).  Prints, for the library in BVC_LIB_PATH (or the product library): relative L2 and maximum absolute error of
both outputs, and how many outputs are off the bf16 rounding of the reference by more than one / two bf16 steps.  The reference's inputs
are the bf16 operands, so what is left is the accumulation order, the GELU arithmetic (A&S 7.1.26 or the table) and the bf16 rounding."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"


def main():
    print(f"tools/ab/gelu_tab_check.py, library: {os.environ.get('BVC_LIB_PATH', 'product')}")
    for M, N, K, scale in ((4096, 3072, 768, 0.05), (8192, 1536, 384, 0.12), (2048, 3072, 768, 0.3)):
        A, W = G.bf16_randn(M, K, seed=1), (G.bf16_randn(N, K, seed=2).float() * scale).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) * 0.5
        C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        C2 = torch.zeros_like(C)
        d = G.gemm_desc(A, W, M, N, K, G.EPI["GELU"], C, C2=C2, bias=bias)
        G.run_gemm([d], G.NT, 10)
        torch.cuda.synchronize()
        pre = A.double() @ W.double().t() + bias.double()
        Phi = 0.5 * (1.0 + torch.special.erf(pre / math.sqrt(2.0)))
        phi = torch.exp(-0.5 * pre * pre) / math.sqrt(2.0 * math.pi)
        for name, got, want in (("gelu ", C2, pre * Phi), ("gelu'", C, Phi + pre * phi)):
            g = got.double()
            rel = float((g - want).norm() / want.norm())
            mx = float((g - want).abs().max())
            wb = want.to(torch.bfloat16)
            step = (wb.double().abs() * 2.0 ** -8).clamp_min(1e-30)        # >= one bf16 step at that magnitude
            off = (g - wb.double()).abs() / step
            print(f"  M={M} N={N} K={K} pre std {float(pre.std()):.2f}  {name}: rel L2 {rel:.3e}  max abs {mx:.3e}  "
                  f"outputs off the rounded reference by > 1 step: {int((off > 1.001).sum())}, > 2 steps: {int((off > 2.001).sum())} of {g.numel()}")


if __name__ == "__main__":
    main()
