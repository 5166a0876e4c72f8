"""Per-product same-process A/B of the LayerNorm epilogues (gemm8.hip EC 4 / 5) against the product + separate LayerNorm pass they
replace, on the VideoMAE decoder's shapes.  Usage: python tools/ab/rowln_products.py [clips] [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_util as G   # noqa: E402

L = G.L
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
M, N = B * 1568, 384


def ev_time(fn, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def fwd_case(K):
    A = G.bf16_randn(M, K, seed=1)
    W = G.bf16_randn(N, K, scale=0.05, seed=2)
    bias, gamma, beta = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
    resid = torch.randn(M, N, device=dev)
    C, C2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    df = G.gemm_desc(A, W, M, N, K, G.EPI["RESID_LN"], C, bias=bias, resid=resid, C2=C2, ln_gamma=gamma, ln_beta=beta, ln_mean=mean, ln_rstd=rstd, ln_eps=1e-6)
    dp = G.gemm_desc(A, W, M, N, K, G.EPI["RESID"], C, bias=bias, resid=resid)

    def fused():
        G.run_gemm([df], G.NT)

    def gemm_only():
        G.run_gemm([dp], G.NT)

    def ln_only():
        L.check(L.lib().bvc_op_layernorm_fwd(G.ptr(C), 0, 0, 0, G.ptr(gamma), G.ptr(beta), G.ptr(C2), G.ptr(mean), G.ptr(rstd), M, N, 1e-6, G.stream()), "ln")
    return fused, gemm_only, ln_only, G.bvc._ops.gemm_kernel_name(dp, G.NT)


def bwd_case(K):
    dY = G.bf16_randn(M, K, seed=3)
    W = G.bf16_randn(K, N, scale=0.05, seed=4)
    x = torch.randn(M, N, device=dev)
    gamma = torch.ones(N, device=dev)
    mean, rstd = x.mean(1).contiguous(), (1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)).contiguous()
    dres = torch.zeros(M, N, device=dev)
    dbf, dln = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    part = torch.zeros(512 * 2 * N, device=dev)
    ws = torch.zeros(int(L.lib().bvc_op_layernorm_bwd_workspace(M, N)), device=dev)
    df = G.gemm_desc(dY, W, M, N, K, G.EPI["DLN"], dres, C2=dbf, ln_gamma=gamma, ln_mean=mean, ln_rstd=rstd, ln_x=x, ln_part=part, ln_dgamma=dg, ln_dbeta=db)
    dp = G.gemm_desc(dY, W, M, N, K, G.EPI["BF16"], dln)

    def fused():
        G.run_gemm([df], G.NN)

    def gemm_only():
        G.run_gemm([dp], G.NN)

    def ln_only():
        L.check(L.lib().bvc_op_layernorm_bwd(G.ptr(dln), G.ptr(x), 0, 0, 0, G.ptr(mean), G.ptr(rstd), G.ptr(gamma), G.ptr(dres), 1, G.ptr(dbf), G.ptr(dg), G.ptr(db),
                                             G.ptr(ws), M, N, G.stream()), "ln_bwd")
    return fused, gemm_only, ln_only, G.bvc._ops.gemm_kernel_name(dp, G.NN)


print(f"{B} clips: M = {M}, N = {N}; us per launch, median [min .. max] over {ROUNDS} interleaved rounds")
print(f"{'product':28s} {'fused':>22s} {'fused, same start':>22s} {'product alone':>22s} {'LayerNorm alone':>22s} {'fused / (sum)':>14s}   kernel of the separate product")
for name, mk, K in (("dec proj + LN fwd", fwd_case, 384), ("dec fc2 + LN fwd", fwd_case, 1536), ("dec dX fc1 + LN bwd", bwd_case, 1536),
                    ("dec dX qkv + LN bwd", bwd_case, 1152)):
    fused, gemm_only, ln_only, kname = mk(K)
    for f in (fused, gemm_only, ln_only):
        ev_time(f, 2)
    t = {0: [], 1: [], 2: [], 3: []}
    for r in range(ROUNDS):
        for i, f in ((0, fused), (3, fused), (1, gemm_only), (2, ln_only), (2, ln_only), (1, gemm_only), (3, fused), (0, fused)):
            G.L.set_option("row_stagger", 0 if i == 3 else 1)
            t[i].append(ev_time(f))
    G.L.set_option("row_stagger", 1)

    def fmt(a):
        a = np.array(a)
        return f"{np.median(a):8.1f} [{a.min():6.1f} ..{a.max():7.1f}]"
    ratio = np.median(t[0]) / (np.median(t[1]) + np.median(t[2]))
    print(f"{name:28s} {fmt(t[0]):>22s} {fmt(t[3]):>22s} {fmt(t[1]):>22s} {fmt(t[2]):>22s} {ratio:14.3f}   {kname}", flush=True)
    del fused, gemm_only, ln_only
    torch.cuda.empty_cache()
