"""(tile, split-K) sweep of the weight-gradient (TN) products at the VideoMAE-base shapes, singles and the per-layer
grouped launch, all in ONE process so the comparison is same-box.  Steers stack.hip:plan_dw.  -> gpurun_out/dw_sweep.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = "cuda"


def main():
    Bc = int(os.environ.get("BVC_BATCH", "16"))
    Me, Md, Mm = Bc * 160, Bc * 1568, Bc * 1408
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    singles = [("head dW", 1536, 384, Mm), ("e2d dW", 384, 768, Md), ("patch dW", 768, 1536, Me), ("dec fc1 dW", 1536, 384, Md),
               ("jepa proj dW", 768, 384, Bc * 4 * 40), ("jepa embed dW", 384, 768, Bc * 4 * 100)]
    for name, M, N, K in singles:
        A, B = G.bf16_randn(K, M), G.bf16_randn(K, N)
        C = torch.zeros(M, N, device=dev)
        res = []
        for tile in (0, 1, 2):
            for split in (1, 2, 3, 4, 6, 8, 12, 16, 24):
                if (K + 63) // 64 < split * 4:
                    continue
                d = G.gemm_desc(A, B, M, N, K, G.EPI["F32"], C, split_k=split)
                ms = timeit(lambda: G.run_gemm([d], G.TN, tile, 2))
                res.append((ms, tile, split))
        res.sort()
        emit(f"{name:14s} M{M} N{N} K{K}: " + "  ".join(f"t{t}s{s}={ms*1e3:.1f}us" for ms, t, s in res[:8]))
    for tag, M, D, I in (("enc", Me, 768, 3072), ("dec", Md, 384, 1536), ("jepa pred", Bc * 4 * 140, 384, 1536), ("jepa enc", Bc * 100, 768, 3072)):
        dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
        dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
        dqkv = G.bf16_randn(M, 3 * D, seed=11)
        outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
        bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
        res = []
        for tile in (0, 1, 2):
            for split in (1, 2, 3, 4, 5, 6, 8):
                if (M + 63) // 64 < split * 4:
                    continue
                descs = [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),
                         G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),
                         G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),
                         G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]
                ms = timeit(lambda: G.run_gemm(descs, G.TN, tile, 2))
                res.append((ms, tile, split))
        res.sort()
        emit(f"{tag:10s} group K{M} D{D}: " + "  ".join(f"t{t}s{s}={ms*1e3:.1f}us" for ms, t, s in res[:10]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dw_sweep.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
