"""Experiments build: the A-stationary kernel with its stores dropped at the descriptor (BVC_GEMM_DEBUG=1) against the real thing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_util as G   # noqa: E402

dev = torch.device("cuda:0")
M, K = 256 * 1568, 384


def ev_time(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


A = G.bf16_randn(M, K, seed=1)
for epi, N in (("BF16", 1152), ("GELU", 1536)):
    W = G.bf16_randn(N, K, scale=0.05, seed=2)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == "GELU" else None
    d = G.gemm_desc(A, W, M, N, K, G.EPI[epi], C, bias=torch.zeros(N, device=dev), C2=C2)
    for tile in (15, 16, 10):
        out = []
        for dbg in ("0", "1"):
            os.environ["BVC_GEMM_DEBUG"] = dbg
            f = lambda: G.run_gemm([d], G.NT, tile_cfg=tile)
            ev_time(f, 2)
            out.append(min(ev_time(f) for _ in range(3)))
        os.environ["BVC_GEMM_DEBUG"] = "0"
        print(f"{epi:5s} N={N} tile {tile}: {out[0]:8.1f} us with stores, {out[1]:8.1f} us with the stores dropped", flush=True)
