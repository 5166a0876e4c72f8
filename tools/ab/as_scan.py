"""gemm_as.hip: time per 128-row unit as a function of the number of N tiles (slope = one N tile = 6 K steps + its epilogue, intercept = the
per-unit cost: A-block burst, fragment reads).  Usage: python tools/ab/as_scan.py [clips]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_util as G   # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, K = B * 1568, 384
units_per_cu = (M / 128) / 256


def ev_time(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


A = G.bf16_randn(M, K, seed=1)
print(f"M = {M} ({units_per_cu:.2f} units per CU); us per launch -> us per unit")
for epi in ("BF16", "GELU"):
    for tile in (15, 16):
        rows = []
        for N in (384, 768, 1152, 1536):
            W = G.bf16_randn(N, K, scale=0.05, seed=2)
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == "GELU" else None
            d = G.gemm_desc(A, W, M, N, K, G.EPI[epi], C, bias=torch.zeros(N, device=dev), C2=C2)
            f = lambda: G.run_gemm([d], G.NT, tile_cfg=tile)
            ev_time(f, 2)
            t = min(ev_time(f) for _ in range(3))
            rows.append((N // 128, t, t / np.ceil(units_per_cu)))
            del W, C, C2
        nt = np.array([r[0] for r in rows], dtype=float)
        pu = np.array([r[2] for r in rows])
        slope, icpt = np.polyfit(nt, pu, 1)
        print(f"{epi:5s} tile {tile}: " + "  ".join(f"N={128 * int(a)}: {b:7.1f} us ({c:5.1f}/unit)" for a, b, c in rows) +
              f"   -> {slope:5.2f} us per N tile ({slope / 6:5.3f} per K step incl. epilogue share), {icpt:5.2f} us per unit fixed", flush=True)
