"""Same-process A/B of the 256-row persistent GEMM (csrc/gemm8.hip, tile configs 10 / 11) against the 128 x 128 per-tile /
persistent kernels ("auto" with bvc_set_option("gemm8", -1)) on every product of the VideoMAE-base step at BVC_BATCH clips.  Interleaved rounds,
median of the per-round times (HIP events on the launch stream)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"


def time_once(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    G.L.set_option("gemm8", -1)       # "auto" below = the selection WITHOUT the 256-row kernel; tiles 10 / 11 name it explicitly
    Bc = int(os.environ.get("BVC_BATCH", "64"))
    Me, Md, Mm = Bc * 160, Bc * 1568, Bc * 1408
    cases = [
        ("enc qkv", G.NT, Me, 2304, 768, "BF16"), ("enc proj", G.NT, Me, 768, 768, "RESID"), ("enc fc1", G.NT, Me, 3072, 768, "GELU"),
        ("enc fc2", G.NT, Me, 768, 3072, "RESID"),
        ("dec qkv", G.NT, Md, 1152, 384, "BF16"), ("dec proj", G.NT, Md, 384, 384, "RESID"), ("dec fc1", G.NT, Md, 1536, 384, "GELU"),
        ("dec fc2", G.NT, Md, 384, 1536, "RESID"), ("head", G.NT, Mm, 1536, 384, "BF16"), ("patch", G.NT, Me, 768, 1536, "F32"),
        ("enc dX fc2", G.NN, Me, 3072, 768, "DGELU"), ("enc dX fc1", G.NN, Me, 768, 3072, "BF16"), ("enc dX qkv", G.NN, Me, 768, 2304, "BF16"),
        ("enc dX proj", G.NN, Me, 768, 768, "BF16"),
        ("dec dX fc2", G.NN, Md, 1536, 384, "DGELU"), ("dec dX fc1", G.NN, Md, 384, 1536, "BF16"), ("dec dX qkv", G.NN, Md, 384, 1152, "BF16"),
        ("dec dX proj", G.NN, Md, 384, 384, "BF16"),
        ("square 4096", G.NT, 4096, 4096, 4096, "BF16"), ("square 8192", G.NT, 8192, 8192, 8192, "BF16"),
    ]
    rounds = int(os.environ.get("BVC_ROUNDS", "7"))
    for name, lay, M, N, K, epi in cases:
        if lay == G.NT:
            A, B = G.bf16_randn(M, K), G.bf16_randn(N, K, seed=1)
        else:
            A, B = G.bf16_randn(M, K), G.bf16_randn(K, N, seed=1)
        f32 = epi in ("RESID", "F32")
        C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        kw = {}
        if epi == "GELU":
            kw["C2"] = torch.zeros_like(C)
        if epi == "RESID":
            kw["resid"] = torch.randn(M, N, device=dev)
        if epi == "DGELU":
            kw["aux"] = G.bf16_randn(M, N, seed=5)
        if epi != "DGELU":
            kw["bias"] = torch.randn(N, device=dev)
        d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **kw)
        tiles = [-1, 10, 11]
        times = {t: [] for t in tiles}
        iters = 5
        for t in tiles:
            for _ in range(2):
                G.run_gemm([d], lay, t)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for t in tiles:
                times[t].append(time_once(lambda: G.run_gemm([d], lay, t), iters))
        fl = 2.0 * M * N * K
        parts = []
        for t in tiles:
            us = statistics.median(times[t])
            parts.append(f"{'auto' if t < 0 else 'tile%d' % t} {us:7.1f}us {fl / us / 1e6:6.1f}TF")
        print(f"{name:12s} {epi:6s} " + " | ".join(parts), flush=True)
    # the grouped weight-gradient launch of one layer
    for tag, M, D, I in (("enc", Me, 768, 3072), ("dec", Md, 384, 1536)):
        dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
        dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
        dqkv = G.bf16_randn(M, 3 * D, seed=11)
        outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
        bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
        flops = 2.0 * M * (D * I * 2 + D * D * 4)

        def mk(split):
            return [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),
                    G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),
                    G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),
                    G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]
        combos = [("auto128", 0, 4 if not (tag == "enc" and Bc <= 16) else 1)]
        for tile in (10, 11):
            for split in ((1, 2, 3, 4) if tag == "enc" else (2, 4, 6, 8, 12)):
                combos.append((f"tile{tile}", tile, split))
        res = {}
        for _ in range(rounds):
            for nm, tile, split in combos:
                ds = mk(split)
                if (nm, split) not in res:
                    res[(nm, split)] = []
                    G.run_gemm(ds, G.TN, tile)
                    torch.cuda.synchronize()
                res[(nm, split)].append(time_once(lambda: G.run_gemm(ds, G.TN, tile), 3))
        for (nm, split), v in res.items():
            us = statistics.median(v)
            print(f"{tag} dW group {nm} split {split}: {us:7.1f}us {flops / us / 1e6:6.1f}TF", flush=True)


if __name__ == "__main__":
    main()
