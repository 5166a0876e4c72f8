"""Race screen for the persistent one-workgroup-per-CU GEMM (csrc/gemm8.hip, tile configs 10 / 11 / 12): many back-to-back launches of every
epilogue class (0: BF16 / GELU / RELU in registers, un-drained stores; 3: GELU'; 1: f32 outputs with side inputs, LOSS; 2: weight
gradients, unsplit and as the split-K group of a layer) while streaming kernels keep the memory system busy.

What is compared: for NT / NN / unsplit TN products every launch must equal, BIT FOR BIT, the per-tile kernel's result (tile
config 0: same K order per output element, same epilogue arithmetic - csrc/gemm8.hip header).  The split-K group accumulates with
f32 atomics, whose order is free: there every launch must agree with the unsplit per-tile result to 2e-5 of the output's norm (a
staged K tile read before its LDS-DMA landed, or re-staged under a reader, is O(1) wrong in the tiles it hits).

A synchronisation slip in the stream of K tiles (staggered wave rows, counted vmcnt across unit boundaries, stores queued behind
the next unit's prefetch) shows as a mismatch that comes and goes with memory load; every launch has to match."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"
iters = int(os.environ.get("BVC_SCREEN_ITERS", "100"))
EPI = dict(G.EPI, RELU=9, DRELU=10)
# (layout, epilogue, M, N, K, tile): shapes of the step at 64 - 256 clips plus ragged ones (M, N not multiples of the tile)
cases = [
    (G.NT, "BF16", 100352, 1152, 384, 10), (G.NT, "GELU", 50176, 1536, 384, 10), (G.NT, "RELU", 20000, 776, 768, 10),
    (G.NT, "BF16", 40960, 2304, 768, 11), (G.NN, "BF16", 100352, 384, 1536, 11), (G.NN, "BF16", 40960, 768, 3072, 10),
    (G.NN, "DGELU", 50176, 1536, 384, 10), (G.NN, "DGELU", 20480, 3072, 768, 10),
    (G.NT, "RESID", 40960, 768, 3072, 10), (G.NT, "RESID", 100352, 384, 1536, 11), (G.NT, "F32", 20000, 776, 1536, 10),
    (G.NT, "LOSS", 45056, 1536, 384, 10),
    (G.TN, "F32", 768, 3072, 20480, 10), (G.TN, "F32", 1152, 384, 50176, 11), (G.TN, "F32", 1152, 1536, 30080, 12),
    (G.TN, "F32", 392, 776, 9000, 12),
]
bad = 0
noise = torch.randn(64 << 20, device=dev)
for layout, epi, M, N, K, tile in cases:
    if layout == G.TN:
        A, B = G.bf16_randn(K, M, seed=1), G.bf16_randn(K, N, seed=2)
    else:
        A = G.bf16_randn(M, K, seed=1)
        B = G.bf16_randn(N, K, seed=2) if layout == G.NT else G.bf16_randn(K, N, seed=2)
    f32 = epi in ("F32", "RESID")
    kw = {}
    if epi in ("BF16", "GELU", "RELU", "RESID", "F32", "LOSS") and layout != G.TN:
        kw["bias"] = torch.randn(N, device=dev)
    if epi == "RESID":
        kw["resid"] = torch.randn(M, N, device=dev)
    if epi == "DGELU":
        kw["aux"] = G.bf16_randn(M, N, seed=3)
    if epi == "LOSS":
        kw["labels"] = torch.randn(M, N, device=dev)

    def run(t, C, C2, part, rs):
        k2 = dict(kw)
        if epi == "GELU":
            k2["C2"] = C2
        if epi == "LOSS":
            k2["partial"] = part
        if layout == G.TN:
            k2["rowsum"] = rs
            k2["lda"], k2["ldb"] = M, N
        G.run_gemm([G.gemm_desc(A, B, M, N, K, EPI[epi], C, **k2)], layout, t, -1)

    dt = torch.float32 if f32 or layout == G.TN else torch.bfloat16

    def outs():
        return (torch.zeros(M, N, device=dev, dtype=dt), torch.zeros(M, N, device=dev, dtype=torch.bfloat16),
                torch.zeros(1 << 16, device=dev), torch.zeros(M, device=dev))

    ref = outs()
    run(0, *ref)
    torch.cuda.synchronize()
    mism = 0
    for it in range(iters):
        got = outs()
        noise.mul_(1.0001)                      # a streaming kernel in front, so launches overlap with memory traffic
        run(tile, *got)
        noise.add_(0.5)
        ok = torch.equal(got[0], ref[0]) and (epi != "GELU" or torch.equal(got[1], ref[1]))
        if epi == "LOSS":      # per-tile partials are indexed by tile: compare their sum (tile sizes differ), loosely
            ok = ok and abs(float(got[2].sum()) - float(ref[2].sum())) <= 1e-4 * abs(float(ref[2].sum()))
        if layout == G.TN:     # the fused bias gradient is atomically accumulated over column-0 workgroups only: order-free sums
            ok = ok and float((got[3] - ref[3]).norm()) <= 2e-5 * float(ref[3].norm())
        if not ok and mism == 0:      # say what differs, once per case
            d = (got[0].float() - ref[0].float()).abs()
            print(f"   first mismatch at launch {it}: C max abs diff {float(d.max()):.3e} in {int((d > 0).sum())} elements"
                  f" (|C| max {float(ref[0].float().abs().max()):.3e}); partial sums {float(got[2].sum()):.9e} vs {float(ref[2].sum()):.9e};"
                  f" rowsum diff {float((got[3] - ref[3]).norm()):.3e}", flush=True)
        mism += 0 if ok else 1
    torch.cuda.synchronize()
    print(f"{['NT', 'NN', 'TN'][layout]} {epi:5s} M={M} N={N} K={K} tile{tile}: {iters} launches, {mism} mismatches", flush=True)
    bad += mism

# the split-K weight-gradient group of one layer (class 2 with atomics), encoder and decoder widths, as plan_dw launches it
for tag, M, D, I, tile, split in (("enc", 20480, 768, 3072, 10, 2), ("dec", 100352, 384, 1536, 10, 6), ("dec", 50176, 384, 1536, 11, 4),
                                 ("dec", 100352, 384, 1536, 12, 7), ("dec", 25088, 384, 1536, 12, 7),
                                 # round 4: the balanced walk (enc: 216 units + 108 tails) at 64 clips; tile config 13 (accumulated, unsplit, ViT-L widths)
                                 ("enc", 10240, 768, 3072, 10, 2), ("vit-l", 25600, 1024, 4096, 13, 1)):
    dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
    dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
    dqkv = G.bf16_randn(M, 3 * D, seed=11)
    shapes = [(D, I), (I, D), (D, D), (3 * D, D)]

    def group(outs, bs, sp):
        return [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=sp),
                G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=sp),
                G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=sp),
                G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=sp)]

    def fresh():
        return [torch.zeros(s, device=dev) for s in shapes], [torch.zeros(s[0], device=dev) for s in shapes]

    ro, rb = fresh()
    G.run_gemm(group(ro, rb, 1), G.TN, 0)
    torch.cuda.synchronize()
    mism, worst = 0, 0.0
    n = max(20, iters // 3)
    for it in range(n):
        o, b = fresh()
        noise.mul_(1.0001)
        G.run_gemm(group(o, b, split), G.TN, tile)
        noise.add_(0.5)
        e = max(float((x - y).norm() / y.norm()) for x, y in zip(o + b, ro + rb))
        worst = max(worst, e)
        mism += 0 if e <= 2e-5 else 1
    print(f"TN group {tag} M={M} tile{tile} split {split}: {n} launches, {mism} beyond 2e-5 (worst rel {worst:.1e})", flush=True)
    bad += mism
# round 5: the row epilogues of the 128 x 384 NT / NN tiles (EC 4: residual + LayerNorm forward, EC 5: LayerNorm backward; staggered start,
# cross-wave statistics through LDS behind ONE barrier, wave-private column accumulators) and the A-stationary kernel (gemm_as.hip: A block
# in registers, three-slot B ring, counted waits that include the epilogue's stores, late wave group one barrier behind).  The f32 output of
# EC 4 must equal the per-tile residual epilogue bit for bit; everything else (bf16 LayerNorm output, mean, rstd, the LayerNorm backward's
# rows) is a fixed-order computation and must equal the FIRST launch bit for bit; dgamma / dbeta go through atomics (1e-6).
for M, K in ((100352, 384), (100352, 1536), (50300, 1152)):
    N = 384
    A, W = G.bf16_randn(M, K, seed=21), G.bf16_randn(N, K, scale=0.05, seed=22)
    bias, gamma, beta = torch.randn(N, device=dev) * 0.1, 1 + 0.1 * torch.randn(N, device=dev), 0.1 * torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev)
    Cp = torch.zeros(M, N, device=dev)
    G.run_gemm([G.gemm_desc(A, W, M, N, K, G.EPI["RESID"], Cp, bias=bias, resid=resid)], G.NT, 0)
    first, mism = None, 0
    for it in range(iters):
        C, C2 = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
        noise.mul_(1.0001)
        G.run_gemm([G.gemm_desc(A, W, M, N, K, G.EPI["RESID_LN"], C, bias=bias, resid=resid, C2=C2, ln_gamma=gamma, ln_beta=beta, ln_mean=mean,
                                ln_rstd=rstd, ln_eps=1e-6)], G.NT)
        noise.add_(0.5)
        if first is None:
            first = (C2, mean, rstd)
        ok = torch.equal(C, Cp) and torch.equal(C2, first[0]) and torch.equal(mean, first[1]) and torch.equal(rstd, first[2])
        mism += 0 if ok else 1
    print(f"NT RESID_LN M={M} K={K} tile12: {iters} launches, {mism} mismatches", flush=True)
    bad += mism
    dY, Wt = G.bf16_randn(M, K, seed=23), G.bf16_randn(K, N, scale=0.05, seed=24)
    x = torch.randn(M, N, device=dev) * 2 + 0.5
    mu, rs = x.mean(1).contiguous(), (1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)).contiguous()
    dres0, part = torch.randn(M, N, device=dev), torch.zeros(512 * 2 * N, device=dev)
    first, mism = None, 0
    for it in range(iters):
        dres, dbf = dres0.clone(), torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        noise.mul_(1.0001)
        G.run_gemm([G.gemm_desc(dY, Wt, M, N, K, G.EPI["DLN"], dres, C2=dbf, ln_gamma=gamma, ln_mean=mu, ln_rstd=rs, ln_x=x, ln_part=part,
                                ln_dgamma=dg, ln_dbeta=db)], G.NN)
        noise.add_(0.5)
        if first is None:
            first = (dres, dbf, dg, db)
        ok = torch.equal(dres, first[0]) and torch.equal(dbf, first[1]) and float((dg - first[2]).norm()) <= 1e-6 * float(first[2].norm()) and \
            float((db - first[3]).norm()) <= 1e-6 * float(first[3].norm())
        mism += 0 if ok else 1
    print(f"NN DLN      M={M} K={K} tile12: {iters} launches, {mism} mismatches", flush=True)
    bad += mism
for M, N, epi, tile in ((100352, 1152, "BF16", 15), (100352, 1536, "GELU", 15), (50300, 1536, "GELU", 16), (401408, 1152, "BF16", 15), (100352, 1152, "BF16", 17), (66000, 128, "GELU", 18)):
    K = 384
    A, W, bias = G.bf16_randn(M, K, seed=31), G.bf16_randn(N, K, scale=0.05, seed=32), torch.randn(N, device=dev) * 0.1
    R1, R2 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16), torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    G.run_gemm([G.gemm_desc(A, W, M, N, K, G.EPI[epi], R1, bias=bias, C2=R2 if epi == "GELU" else None)], G.NT, 0)
    mism = 0
    for it in range(iters):
        C1, C2 = torch.zeros_like(R1), torch.zeros_like(R2)
        noise.mul_(1.0001)
        G.run_gemm([G.gemm_desc(A, W, M, N, K, G.EPI[epi], C1, bias=bias, C2=C2 if epi == "GELU" else None)], G.NT, tile)
        noise.add_(0.5)
        ok = torch.equal(C1, R1) and (epi != "GELU" or torch.equal(C2, R2))
        mism += 0 if ok else 1
    print(f"NT {epi:5s} M={M} N={N} K={K} tile{tile} (gemm_as): {iters} launches, {mism} mismatches", flush=True)
    bad += mism
print("GEMM8 RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} mismatching launches)")
sys.exit(1 if bad else 0)
