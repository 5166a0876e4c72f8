"""Per-kernel microbenchmarks of libbvc_hip.so at the VideoMAE-base shapes (batch 16): TFLOP/s or GB/s per launch,
timed with HIP events on the launch stream.  Used to steer optimisation; numbers go to gpurun_out/micro.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

L = G.L
dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_case(name, layout, M, N, K, tile=-1, split=1, epi="BF16", stages=-1):
    if layout == G.NT:
        A, B = G.bf16_randn(M, K), G.bf16_randn(N, K)
    elif layout == G.NN:
        A, B = G.bf16_randn(M, K), G.bf16_randn(K, N)
    else:
        A, B = G.bf16_randn(K, M), G.bf16_randn(K, N)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == "F32" else torch.bfloat16)
    d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, split_k=split)
    ms = timeit(lambda: G.run_gemm([d], layout, tile, stages))
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    return {"name": name, "layout": ["NT", "NN", "TN"][layout], "M": M, "N": N, "K": K, "tile": tile, "stages": stages, "split": split,
            "ms": round(ms, 4), "tflops": round(tf, 1)}


def attn_case(B, N, H, HD=64):
    D = HD * H
    qkv = G.bf16_randn(B * N, 3 * D)
    ctx = torch.zeros(B * N, D, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B * H, N, device=dev)
    dctx = G.bf16_randn(B * N, D, seed=2)
    dqkv = torch.zeros_like(qkv)
    delta = torch.zeros(B * H, N, device=dev)
    f = lambda: L.check(L.lib().bvc_op_attention_fwd(G.ptr(qkv), G.ptr(ctx), G.ptr(lse), B, N, H, HD, G.stream()))
    b = lambda: L.check(L.lib().bvc_op_attention_bwd(G.ptr(qkv), G.ptr(ctx), G.ptr(dctx), G.ptr(lse), G.ptr(delta), G.ptr(dqkv), B, N, H, HD, G.stream()))
    mf, mb = timeit(f), timeit(b)
    flops = 4.0 * B * H * N * N * HD
    return {"name": f"attn B{B} N{N} H{H}", "fwd_ms": round(mf, 4), "fwd_tflops": round(flops / (mf * 1e-3) / 1e12, 1),
            "bwd_ms": round(mb, 4), "bwd_tflops_5prod": round(2.5 * flops / (mb * 1e-3) / 1e12, 1)}


def main():
    Bc = int(os.environ.get("BVC_BATCH", "16"))
    Me, Md, Mm = Bc * 160, Bc * 1568, Bc * 1408
    out = []
    cases = [
        ("enc qkv", G.NT, Me, 2304, 768), ("enc proj", G.NT, Me, 768, 768), ("enc fc1", G.NT, Me, 3072, 768), ("enc fc2", G.NT, Me, 768, 3072),
        ("dec qkv", G.NT, Md, 1152, 384), ("dec proj", G.NT, Md, 384, 384), ("dec fc1", G.NT, Md, 1536, 384), ("dec fc2", G.NT, Md, 384, 1536),
        ("head", G.NT, Mm, 1536, 384), ("patch", G.NT, Me, 768, 1536),
        ("enc dX fc2", G.NN, Me, 3072, 768), ("enc dX fc1", G.NN, Me, 768, 3072), ("enc dX qkv", G.NN, Me, 768, 2304), ("enc dX proj", G.NN, Me, 768, 768),
        ("dec dX fc2", G.NN, Md, 1536, 384), ("dec dX fc1", G.NN, Md, 384, 1536), ("dec dX qkv", G.NN, Md, 384, 1152), ("dec dX proj", G.NN, Md, 384, 384),
        ("enc dW fc1", G.TN, 3072, 768, Me), ("enc dW fc2", G.TN, 768, 3072, Me), ("enc dW qkv", G.TN, 2304, 768, Me), ("dec dW fc1", G.TN, 1536, 384, Md),
        ("square 4096", G.NT, 4096, 4096, 4096),
    ]
    sweep = os.environ.get("BVC_SWEEP", "1") == "1"
    for name, lay, M, N, K in cases:
        epi = "F32" if lay == G.TN else "BF16"
        split = 5 if (lay == G.TN and "dec" in name) else 1
        combos = [(-1, -1)]
        if sweep:
            combos += [(t, s) for t in (0, 1, 2) for s in (2, 4)]    # stages 4 = the alternative K-loop (same-run A/B)
        best = None
        for tile, st in combos:
            r = gemm_case(name, lay, M, N, K, tile, split, epi, st)
            out.append(r)
            if tile >= 0 and (best is None or r["ms"] < best["ms"]):
                best = r
            if tile < 0:
                auto = r
        os.environ["BVC_GEMM_NO_PERSIST"] = "1"      # same-run A/B: per-tile kernel instead of the persistent one
        legacy = gemm_case(name, lay, M, N, K, -1, split, epi, -1)
        os.environ.pop("BVC_GEMM_NO_PERSIST", None)
        legacy["walk"] = "per-tile"
        out.append(legacy)
        line = f"{name:12s} auto {auto['ms']*1e3:7.1f}us {auto['tflops']:6.1f}TF | per-tile kernel {legacy['ms']*1e3:7.1f}us {legacy['tflops']:6.1f}TF"
        if best:
            line += f" | best tile{best['tile']} st{best['stages']} {best['ms']*1e3:7.1f}us {best['tflops']:6.1f}TF"
        print(line, flush=True)
    # the grouped weight-gradient launch of one layer (4 problems, fused bias gradients)
    for tag, M, D, I in (("enc", Me, 768, 3072), ("dec", Md, 384, 1536)):
        dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
        dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
        dqkv = G.bf16_randn(M, 3 * D, seed=11)
        outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
        bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
        flops = 2.0 * M * (D * I * 2 + D * D * 4)
        for walk in ("panel", "legacy"):
            if walk == "legacy":
                os.environ["BVC_GEMM_LEGACY_WALK"] = "1"
            sp, tl = (1, 0) if tag == "enc" and Bc <= 16 else (4, 0)
            descs = [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=sp),
                     G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=sp),
                     G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=sp),
                     G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=sp)]
            ms = timeit(lambda: G.run_gemm(descs, G.TN, tl, 2))
            os.environ.pop("BVC_GEMM_LEGACY_WALK", None)
            r = {"name": f"{tag} dW group", "walk": walk, "tile": tl, "split": sp, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}
            out.append(r); print(r, flush=True)
        for split in (() if not sweep else ((1, 2) if tag == "enc" else (2, 4, 6))):
            for tile in (0, 1, 2):
                descs = [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),
                         G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),
                         G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),
                         G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]
                for stg in (2, 3):
                    ms = timeit(lambda: G.run_gemm(descs, G.TN, tile, stg))
                    r = {"name": f"{tag} dW group", "tile": tile, "split": split, "lb": stg, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}
                    out.append(r); print(r, flush=True)
    for (B, N, H) in [(Bc, 160, 12), (Bc, 1568, 6), (Bc, 1568, 12)]:
        for plain in (0, 1):    # same-run A/B of the XCD-aware block map (attention.hip:attn_block)
            if plain:
                os.environ["BVC_ATTN_PLAIN_GRID"] = "1"
            else:
                os.environ.pop("BVC_ATTN_PLAIN_GRID", None)
            r = attn_case(B, N, H)
            r["grid"] = "plain" if plain else "xcd"
            out.append(r); print(r, flush=True)
    os.environ.pop("BVC_ATTN_PLAIN_GRID", None)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "micro.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
