"""In-run per-kernel roofline of the VideoMAE-base training step (what bench.py reports as `roofline.kernel` / `roofline.kernels`).

Every GEMM product, attention call and LayerNorm of one step (the schedule of csrc/videomae.hip + csrc/stack.hip, i.e.
VideoMAEForPreTraining.forward / backward, HF:531-671) is launched ALONE on random operands of the step's shapes and timed with
HIP events on the launch stream; `bvc_op_gemm_kernel` (include/bvc.h) names the kernel instantiation each product runs on, exactly
as rocprofv3 prints it, so the rows below can be laid next to a `rocprofv3 --kernel-trace --stats` table of the same command
(profiles/).  Per row: launches per step, time per step, algorithmic FLOPs (2MNK per product; attention: 4 N^2 d per head forward,
10 N^2 d backward = the five products the mathematics needs) and algorithmic bytes (operands read once + outputs and side inputs),
the roof that is further away, and the fraction of it.

Nothing here is a correctness check and nothing here runs inside the timed region of the benchmark.
"""
import ctypes

import torch

from . import _lib, _ops

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E peak, MI355X_MICROARCH.md


def _time(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters       # us per launch


def _rb(dev, *shape):
    return torch.randn(*shape, device=dev, dtype=torch.bfloat16)


def _gemm_rows(dev, B, add):
    """add(kernel, label, launches, us, flops, bytes) for every GEMM of the step."""
    E = _ops.EPI
    Mv, Md, Mm = B * 160, B * 1568, B * 1408
    D, I, Dd, Id, P = 768, 3072, 384, 1536, 1536

    def one(label, layout, M, N, K, epi, count, tile=-1):
        A = _rb(dev, M, K)
        Bm = _rb(dev, N, K) if layout == _ops.NT else _rb(dev, K, N)
        Bm.mul_(0.02)
        kw = {}
        out_f32 = epi in ("F32", "RESID", "POS", "E2D", "F32_BF16", "RESID_LN", "DLN")
        nbytes = 2.0 * (M * K + N * K)
        Mo = M
        if epi == "E2D":
            Mo = (M // 160) * 1568
        C = torch.empty(Mo, N, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
        nbytes += (4.0 if out_f32 else 2.0) * M * N
        if epi not in ("DGELU", "E2D", "DLN"):
            kw["bias"] = torch.zeros(N, device=dev)
        if epi == "GELU":
            kw["C2"] = torch.empty_like(C)
            nbytes += 2.0 * M * N
        if epi == "F32_BF16":
            kw["C2"] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            nbytes += 2.0 * M * N
        if epi in ("RESID", "RESID_LN"):
            kw["resid"] = torch.randn(M, N, device=dev)
            nbytes += 4.0 * M * N
        if epi == "RESID_LN":      # + the LayerNorm of the complete rows: bf16 output, mean / rstd (gemm8.hip EC 4)
            kw.update(C2=torch.empty(M, N, device=dev, dtype=torch.bfloat16), ln_gamma=torch.ones(N, device=dev), ln_beta=torch.zeros(N, device=dev),
                      ln_mean=torch.empty(M, device=dev), ln_rstd=torch.empty(M, device=dev), ln_eps=1e-12)
            nbytes += 2.0 * M * N
        if epi == "DLN":           # LayerNorm backward in the epilogue (gemm8.hip EC 5): LayerNorm input + residual gradient in, gradient + bf16 copy out
            kw.update(C2=torch.empty(M, N, device=dev, dtype=torch.bfloat16), ln_gamma=torch.ones(N, device=dev), ln_x=torch.randn(M, N, device=dev),
                      ln_mean=torch.zeros(M, device=dev), ln_rstd=torch.ones(M, device=dev), ln_part=torch.empty(512 * 2 * N, device=dev),
                      ln_dgamma=torch.zeros(N, device=dev), ln_dbeta=torch.zeros(N, device=dev))
            nbytes += (4.0 + 4.0 + 2.0) * M * N
        if epi == "DGELU":
            kw["aux"] = _rb(dev, M, N)
            nbytes += 2.0 * M * N
        if epi in ("POS", "E2D"):
            kw["rowtok"] = (torch.arange(M, device=dev, dtype=torch.int32) % 160) * 9      # 160 ascending tokens of 1568 per clip
            kw["pos"] = torch.randn(1568, N, device=dev)
            nbytes += 4.0 * M * N
            if epi == "E2D":
                kw["rin"], kw["rout"] = 160, 1568
        if epi == "LOSS":
            kw["labels"] = torch.randn(M, N, device=dev)
            kw["partial"] = torch.empty(1 << 16, device=dev)
            nbytes += 4.0 * M * N
        d = _ops.gemm_desc(A, Bm, M, N, K, E[epi], C, **kw)
        name = _ops.gemm_kernel_name(d, layout, tile)
        us = _time(lambda: _ops.gemm(d, layout, tile))
        add(name, label, count, us, 2.0 * M * N * K, nbytes)

    def dw_single(label, Mo, No, K, count):
        A, Bm = _rb(dev, K, Mo), _rb(dev, K, No)
        C = torch.zeros(Mo, No, device=dev)
        rs = torch.zeros(Mo, device=dev)
        d = _ops.gemm_desc(A, Bm, Mo, No, K, E["F32"], C, rowsum=rs)
        tile, _ = _ops.plan_dw([d])
        name = _ops.gemm_kernel_name(d, _ops.TN, tile)
        us = _time(lambda: _ops.gemm(d, _ops.TN, tile))
        add(name, label, count, us, 2.0 * Mo * No * K, 2.0 * K * (Mo + No) + 4.0 * Mo * No)

    def dw_group(label, M, Dm, Im, count):
        dy, act, dh, ln2, dqkv = _rb(dev, M, Dm), _rb(dev, M, Im), _rb(dev, M, Im), _rb(dev, M, Dm), _rb(dev, M, 3 * Dm)
        shapes = [(Dm, Im), (Im, Dm), (Dm, Dm), (3 * Dm, Dm)]
        outs = [torch.zeros(s, device=dev) for s in shapes]
        bs = [torch.zeros(s[0], device=dev) for s in shapes]
        ds = [_ops.gemm_desc(dy, act, Dm, Im, M, E["F32"], outs[0], rowsum=bs[0]),
              _ops.gemm_desc(dh, ln2, Im, Dm, M, E["F32"], outs[1], rowsum=bs[1]),
              _ops.gemm_desc(dy, ln2, Dm, Dm, M, E["F32"], outs[2], rowsum=bs[2]),
              _ops.gemm_desc(dqkv, ln2, 3 * Dm, Dm, M, E["F32"], outs[3], rowsum=bs[3])]
        tile, split = _ops.plan_dw(ds)
        name = _ops.gemm_kernel_name(ds, _ops.TN, tile)
        us = _time(lambda: _ops.gemm(ds, _ops.TN, tile), iters=3)
        fl = 2.0 * M * (2 * Dm * Im + 4 * Dm * Dm)
        nb = 2.0 * M * (2 * Dm + 2 * Im + 2 * Dm + Dm + 3 * Dm) + 4.0 * (2 * Dm * Im + 4 * Dm * Dm)    # eight operand slices, four outputs
        add(name, f"{label} (tile {tile}, split {split})", count, us, fl, nb)

    NT, NN = _ops.NT, _ops.NN
    # the decoder's LayerNorms inside the neighbouring products' epilogues (csrc/stack.hip: fuse_row_ln) at this batch?
    fused = bool(_lib.lib().bvc_op_row_ln_selected(Md, Dd, Id, 6))
    one("patch embed", NT, Mv, D, 1536, "POS", 1)
    one("enc qkv", NT, Mv, 3 * D, D, "BF16", 12)
    one("enc proj", NT, Mv, D, D, "RESID", 12)
    one("enc fc1+GELU", NT, Mv, I, D, "GELU", 12)
    one("enc fc2", NT, Mv, D, I, "RESID", 12)
    one("enc->dec", NT, Mv, Dd, D, "E2D", 1)
    one("dec qkv", NT, Md, 3 * Dd, Dd, "BF16", 4)
    if fused:
        one("dec proj + LayerNorm", NT, Md, Dd, Dd, "RESID_LN", 4)
    else:
        one("dec proj", NT, Md, Dd, Dd, "RESID", 4)
    one("dec fc1+GELU", NT, Md, Id, Dd, "GELU", 4)
    if fused:      # the last layer's fc2 has no next LayerNorm of the stack to produce
        one("dec fc2 + next LayerNorm", NT, Md, Dd, Id, "RESID_LN", 3)
        one("dec fc2", NT, Md, Dd, Id, "RESID", 1)
    else:
        one("dec fc2", NT, Md, Dd, Id, "RESID", 4)
    one("head+MSE", NT, Mm, P, Dd, "LOSS", 1)
    one("head dX", NN, Mm, Dd, P, "BF16", 1)
    one("dec dX fc2", NN, Md, Id, Dd, "DGELU", 4)
    one("dec dX fc1 + LayerNorm bwd" if fused else "dec dX fc1", NN, Md, Dd, Id, "DLN" if fused else "BF16", 4)
    one("dec dX proj", NN, Md, Dd, Dd, "BF16", 4)
    one("dec dX qkv + LayerNorm bwd" if fused else "dec dX qkv", NN, Md, Dd, 3 * Dd, "DLN" if fused else "BF16", 4)
    one("enc->dec dX", NN, Mv, D, Dd, "F32_BF16", 1)
    one("enc dX fc2", NN, Mv, I, D, "DGELU", 12)
    one("enc dX fc1", NN, Mv, D, I, "BF16", 12)
    one("enc dX proj", NN, Mv, D, D, "BF16", 12)
    one("enc dX qkv", NN, Mv, D, 3 * D, "BF16", 12)
    dw_single("head dW", P, Dd, Mm, 1)
    dw_single("enc->dec dW", Dd, D, Mv, 1)
    dw_single("patch dW", D, 1536, Mv, 1)
    dw_group("dec layer dW group", Md, Dd, Id, 4)
    dw_group("enc layer dW group", Mv, D, I, 12)


def _attention_rows(dev, B, add):
    lib = _lib.lib()
    st = _lib.current_stream_ptr
    for label, N, H, count in (("enc", 160, 12, 12), ("dec", 1568, 6, 4)):
        Dm = 64 * H
        qkv, ctx, dctx, dqkv = _rb(dev, B * N, 3 * Dm), _rb(dev, B * N, Dm), _rb(dev, B * N, Dm), _rb(dev, B * N, 3 * Dm)
        lse = torch.empty(B * H * N, device=dev)
        delta = torch.empty(B * H * N, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        fwd = lambda: _lib.check(lib.bvc_op_attention_fwd(p(qkv), p(ctx), p(lse), B, N, H, 64, st()), "attention_fwd")
        dq = lambda: _lib.check(lib.bvc_op_attention_bwd_part(p(qkv), p(ctx), p(dctx), p(lse), p(delta), p(dqkv), B, N, H, 64, 1, st()), "attention_bwd dq")
        dkv = lambda: _lib.check(lib.bvc_op_attention_bwd_part(p(qkv), p(ctx), p(dctx), p(lse), p(delta), p(dqkv), B, N, H, 64, 2, st()), "attention_bwd dkdv")
        unit = 2.0 * N * N * 64 * H * B                 # one N x N x 64 product per head
        tok = 2.0 * B * N * Dm                          # bytes of one bf16 [B N][D] tensor
        add("bvc::attn_fwd_kernel<64>", f"{label} attention fwd", count, _time(fwd), 2 * unit, 4 * tok)
        if N <= 160:
            # sequences of up to 160 tokens: the whole backward of a (clip, head) in ONE workgroup (csrc/attention.hip: attn_bwd_head_kernel),
            # five products, q / k / v / dO / O read and dq / dk / dv written once
            bwd = lambda: _lib.check(lib.bvc_op_attention_bwd(p(qkv), p(ctx), p(dctx), p(lse), p(delta), p(dqkv), B, N, H, 64, st()), "attention_bwd")
            add(f"bvc::attn_bwd_head_kernel<64, {5 if N > 128 else 0}>", f"{label} attention bwd (whole head: S, dP, dV, dK, dQ)", count, _time(bwd),
                5 * unit, 8 * tok)
            continue
        add("bvc::attn_bwd_dq_kernel<64>", f"{label} attention bwd dQ (S, dP, dQ)", count, _time(dq), 3 * unit, 6 * tok)
        add("bvc::attn_bwd_dkdv_kernel<64>", f"{label} attention bwd dK dV (recomputes S, dP)", count, _time(dkv), 2 * unit, 6 * tok)


def _layernorm_rows(dev, B, add):
    lib = _lib.lib()
    st = _lib.current_stream_ptr
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    fused = bool(lib.bvc_op_row_ln_selected(B * 1568, 384, 1536, 6))     # then only the first layer's first LayerNorm is a pass of its own
    for label, M, Dm, count, count_bwd in (("enc", B * 160, 768, 24, 24), ("dec", B * 1568, 384, 1 if fused else 8, 0 if fused else 8),
                                           ("final", B * 1408, 384, 1, 1)):
        x, dres = torch.randn(M, Dm, device=dev), torch.randn(M, Dm, device=dev)
        gmm, bta = torch.ones(Dm, device=dev), torch.zeros(Dm, device=dev)
        y, dy, dresb = _rb(dev, M, Dm), _rb(dev, M, Dm), _rb(dev, M, Dm)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        dg, db = torch.zeros(Dm, device=dev), torch.zeros(Dm, device=dev)
        ws = torch.empty(int(lib.bvc_op_layernorm_bwd_workspace(M, Dm)), device=dev)
        fwd = lambda: _lib.check(lib.bvc_op_layernorm_fwd(p(x), 0, 0, 0, p(gmm), p(bta), p(y), p(mean), p(rstd), M, Dm, 1e-12, st()), "layernorm_fwd")
        bwd = lambda: _lib.check(lib.bvc_op_layernorm_bwd(p(dy), p(x), 0, 0, 0, p(mean), p(rstd), p(gmm), p(dres), 1, p(dresb), p(dg), p(db),
                                                          p(ws), M, Dm, st()), "layernorm_bwd")
        e = float(M) * Dm
        add("bvc::ln_fwd_kernel", f"{label} LayerNorm fwd", count, _time(fwd), 0.0, 6 * e)
        if count_bwd:
            add("bvc::ln_bwd_kernel (+ ln_param_reduce)", f"{label} LayerNorm bwd", count_bwd, _time(bwd), 0.0, 16 * e)


def step_kernels(batch, device):
    """[{kernel, launches_per_step, us_per_step, share_of_probed, tflops, gb_per_s, bound, frac, products}] sorted by time."""
    rows = {}

    def add(kernel, label, count, us, flops, nbytes):
        r = rows.setdefault(kernel, {"kernel": kernel, "launches_per_step": 0, "us_per_step": 0.0, "flops": 0.0, "bytes": 0.0, "products": []})
        r["launches_per_step"] += count
        r["us_per_step"] += count * us
        r["flops"] += count * flops
        r["bytes"] += count * nbytes
        r["products"].append({"name": label, "launches": count, "launch_us": round(us, 1),
                              "tflops": round(flops / us / 1e6, 1), "gb_per_s": round(nbytes / us / 1e3, 1)})

    with torch.no_grad():
        _gemm_rows(device, batch, add)
        _attention_rows(device, batch, add)
        _layernorm_rows(device, batch, add)
    torch.cuda.empty_cache()
    out = sorted(rows.values(), key=lambda r: -r["us_per_step"])
    total = sum(r["us_per_step"] for r in out)
    for r in out:
        us = r["us_per_step"]
        t_mfma, t_hbm = r["flops"] / (PEAK_BF16_TFLOPS * 1e6), r["bytes"] / (PEAK_HBM_GBS * 1e3)
        r["share_of_probed"] = round(us / total, 4)
        r["tflops"] = round(r["flops"] / us / 1e6, 1)
        r["gb_per_s"] = round(r["bytes"] / us / 1e3, 1)
        r["bound"] = "hbm" if t_hbm > t_mfma else "mfma"
        r["frac"] = round(max(t_mfma, t_hbm) / us, 4)
        r["us_per_step"] = round(us, 1)
        r["algorithmic_flops"] = r.pop("flops")
        r["algorithmic_bytes"] = r.pop("bytes")
    return out, total
