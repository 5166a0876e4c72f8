"""Race screen for the persistent GEMM: many back-to-back launches at several shapes / layouts / epilogues while other launches
keep the memory system busy, each compared bit-for-bit with the per-tile kernel's result.  A synchronisation slip in the
chained K-step stream (a slot read before its LDS-DMA landed, or refilled before every wave left it) shows up as a mismatch
that comes and goes; every iteration must match."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"
iters = int(os.environ.get("BVC_SCREEN_ITERS", "150"))
cases = [(G.NT, "BF16", 100352, 1152, 384, 3), (G.NT, "GELU", 50176, 1536, 384, 3), (G.NN, "DGELU", 50176, 1536, 384, 3), (G.NT, "BF16", 9000, 2056, 128, 3),
         (G.NT, "BF16", 100352, 1152, 384, 0), (G.NT, "GELU", 25088, 1536, 384, 0), (G.NN, "DGELU", 25088, 1536, 384, 0),
         (G.NT, "RESID", 20000, 776, 768, 0), (G.NN, "BF16", 50176, 384, 1152, 0), (G.NT, "BF16", 30000, 392, 384, 1),
         (G.NT, "F32", 9000, 2056, 128, 0), (G.NN, "RESID", 16640, 1024, 3072, 0)]
bad = 0
for layout, epi, M, N, K, tile in cases:
    A = G.bf16_randn(M, K, seed=1)
    B = G.bf16_randn(N, K, seed=2) if layout == G.NT else G.bf16_randn(K, N, seed=2)
    f32 = epi in ("F32", "RESID")
    kw = {}
    if epi == "GELU":
        kw = dict(bias=torch.randn(N, device=dev))
    if epi == "RESID":
        kw = dict(bias=torch.randn(N, device=dev), resid=torch.randn(M, N, device=dev))
    if epi == "DGELU":
        kw = dict(aux=G.bf16_randn(M, N, seed=3))

    def run(t, C, C2):
        k2 = dict(kw)
        if epi == "GELU":
            k2["C2"] = C2
        G.run_gemm([G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **k2)], layout, t, -1)

    dt = torch.float32 if f32 else torch.bfloat16
    ref, ref2 = torch.zeros(M, N, device=dev, dtype=dt), torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    run(0 if tile == 3 else tile, ref, ref2)
    torch.cuda.synchronize()
    noise = torch.randn(64 << 20, device=dev)
    mism = 0
    for it in range(iters):
        C, C2 = torch.empty_like(ref), torch.empty_like(ref2)
        noise.mul_(1.0001)                      # a streaming kernel in front, so launches overlap with memory traffic
        run(9 if tile == 3 else tile + 6, C, C2)      # "tile 3" in the case list = deferred-store variant (config 9) against config 0
        noise.add_(0.5)
        if it % 10 == 9 or it == iters - 1:
            torch.cuda.synchronize()
        if not torch.equal(C, ref) or (epi == "GELU" and not torch.equal(C2, ref2)):
            mism += 1
    torch.cuda.synchronize()
    print(f"layout {['NT','NN'][layout]} {epi:5s} M={M} N={N} K={K} {'deferred' if tile == 3 else 'tile' + str(tile)}: {iters} launches, {mism} mismatches", flush=True)
    bad += mism
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} mismatching launches)")
sys.exit(1 if bad else 0)
