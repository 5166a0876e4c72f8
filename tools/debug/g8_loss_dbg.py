"""Where does the LOSS epilogue of gemm8 (tile 10 / 11) differ from the per-tile kernel once a workgroup walks several units?"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G
dev = "cuda"
for tile in (10, 11):
    for M, N, K in ((45056, 1536, 384), (8192, 1536, 384), (65536, 512, 384), (131072, 256, 128)):
        A, B = G.bf16_randn(M, K, seed=1), G.bf16_randn(N, K, seed=2)
        bias, labels = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
        def run(t):
            C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            part = torch.zeros(1 << 16, device=dev)
            G.run_gemm([G.gemm_desc(A, B, M, N, K, G.EPI["LOSS"], C, bias=bias, labels=labels, partial=part)], G.NT, t, -1)
            torch.cuda.synchronize()
            return C, part
        r, rp = run(0)
        g, gp = run(tile)
        bad = (g.float() != r.float()).nonzero()
        print(f"tile{tile} M={M} N={N} K={K}: {bad.shape[0]} wrong elements; partial sum {float(gp.sum()):.6e} vs {float(rp.sum()):.6e}")
        if bad.shape[0]:
            rows, cols = bad[:, 0], bad[:, 1]
            print("   rows mod 256:", sorted(set((rows % 256).tolist()))[:40])
            print("   row tiles:", sorted(set((rows // 256).tolist()))[:40])
            print("   cols:", sorted(set(cols.tolist()))[:64])
            print("   sample got/ref:", [(float(g[i, j]), float(r[i, j])) for i, j in bad[:6].tolist()])
