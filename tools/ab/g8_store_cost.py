"""Experiments build only (tools/gpu_check.sh exp): the 256 x 256 persistent GEMM with and without its global stores
(BVC_GEMM_DEBUG=1 -> zero-record store descriptors: dropped by the range check, same instruction stream and waits)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"


def t(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


Bc = 64
for name, M, N, K, epi in [("dec qkv", Bc * 1568, 1152, 384, "BF16"), ("dec fc1", Bc * 1568, 1536, 384, "GELU"), ("enc fc1", Bc * 160, 3072, 768, "GELU"),
                           ("enc qkv", Bc * 160, 2304, 768, "BF16"), ("square 8192", 8192, 8192, 8192, "BF16")]:
    A, B = G.bf16_randn(M, K), G.bf16_randn(N, K, seed=1)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    kw = {"bias": torch.randn(N, device=dev)}
    if epi == "GELU":
        kw["C2"] = torch.zeros_like(C)
    d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **kw)
    res = {}
    for rnd in range(5):
        for dbg in ("0", "1"):
            os.environ["BVC_GEMM_DEBUG"] = dbg
            G.run_gemm([d], G.NT, 10)
            torch.cuda.synchronize()
            res.setdefault(dbg, []).append(t(lambda: G.run_gemm([d], G.NT, 10)))
    os.environ["BVC_GEMM_DEBUG"] = "0"
    a, b = statistics.median(res["0"]), statistics.median(res["1"])
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{name:12s} {epi:5s} {tiles:5d} tiles ({tiles / 256:.2f} per CU): with stores {a:7.1f} us, stores dropped {b:7.1f} us", flush=True)
