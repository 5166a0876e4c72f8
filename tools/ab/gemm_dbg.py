"""Where does the per-tile fixed cost of the short-K decoder GEMMs go?  Same-process experiments through BVC_GEMM_DEBUG
(1 = drop the bf16 stores, 2 = odd resident slots start ~3 us late).  -> gpurun_out/gemm_dbg.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = "cuda"
lines = []
M = int(os.environ.get("BVC_BATCH", "64")) * 1568
for N, K, layout, epi in ((1152, 384, G.NT, "BF16"), (1152, 1536, G.NT, "BF16"), (1536, 384, G.NT, "GELU"), (384, 1536, G.NN, "BF16"),
                          (1152, 64, G.NT, "BF16")):
    A = G.bf16_randn(M, K)
    B = G.bf16_randn(N, K) if layout == G.NT else G.bf16_randn(K, N)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    kw = {}
    if epi == "GELU":
        kw = dict(C2=torch.zeros(M, N, device=dev, dtype=torch.bfloat16), bias=torch.zeros(N, device=dev))
    if epi == "DGELU":
        kw = dict(aux=G.bf16_randn(M, N, seed=3))
    d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **kw)
    row = []
    for dbg in (0, 1, 1 + 8, 1 + 32, 1 + 16, 1 + 16 + 32, 16):
        os.environ["BVC_GEMM_DEBUG"] = str(dbg)
        row.append(timeit(lambda: G.run_gemm([d], layout, 0, -1), iters=10, warm=2) * 1e3)
    os.environ.pop("BVC_GEMM_DEBUG", None)
    s = (f"M={M} N={N} K={K} {['NT','NN','TN'][layout]} {epi:5s}: normal {row[0]:.0f}us | no stores {row[1]:.0f} | no stores, A refills only {row[2]:.0f} | "
         f"no stores, no refills {row[3]:.0f} | no stores, no MFMA {row[4]:.0f} | no stores/MFMA/refills {row[5]:.0f} | no MFMA (loads + stores) {row[6]:.0f}")
    print(s, flush=True)
    lines.append(s)
# persistent kernel (tile config 6): plain vs s_setprio(1) over the MFMA bursts (BVC_GEMM_DEBUG=64)
for N, K, layout, epi in ((1152, 384, G.NT, "BF16"), (1536, 384, G.NT, "GELU"), (1536, 384, G.NN, "DGELU"), (1152, 1536, G.NT, "BF16")):
    A = G.bf16_randn(M, K)
    B = G.bf16_randn(N, K) if layout == G.NT else G.bf16_randn(K, N)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    kw = {}
    if epi == "GELU":
        kw = dict(C2=torch.zeros(M, N, device=dev, dtype=torch.bfloat16), bias=torch.zeros(N, device=dev))
    if epi == "DGELU":
        kw = dict(aux=G.bf16_randn(M, N, seed=3))
    d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **kw)
    row = []
    for dbg in (0, 64, 0, 64):
        os.environ["BVC_GEMM_DEBUG"] = str(dbg)
        row.append(timeit(lambda: G.run_gemm([d], layout, 6, -1), iters=10, warm=2) * 1e3)
    os.environ.pop("BVC_GEMM_DEBUG", None)
    s = f"persistent M={M} N={N} K={K} {['NT','NN','TN'][layout]} {epi:5s}: plain {row[0]:.0f} / {row[2]:.0f} us | setprio {row[1]:.0f} / {row[3]:.0f} us"
    print(s, flush=True)
    lines.append(s)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_dbg.txt"), "w").write("\n".join(lines) + "\n")
