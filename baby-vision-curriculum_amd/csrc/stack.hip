// Shared pre-LN transformer stack: allocation, one-layer forward / backward schedules (host code).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "stack.h"

namespace bvc {

int64_t ParamTable::add(const std::string& name, std::initializer_list<int64_t> shp) {
    ParamEntry e;
    e.name = name;
    e.offset = total;
    e.numel = 1;
    e.ndim = (int)shp.size();
    int i = 0;
    for (auto s : shp) { e.shape[i++] = s; e.numel *= s; }
    for (; i < 5; ++i) e.shape[i] = 1;
    entries.push_back(e);
    total += e.numel;
    return e.offset;
}

// hf_names: transformers' VideoMAELayer keys (separate query/key/value tensors, adjacent in memory);
// otherwise the JEPA Block keys (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2)
LayerOff add_layer_params(ParamTable& t, const std::string& p, int64_t d, int64_t inter, bool hf_names) {
    LayerOff o;
    if (hf_names) {
        o.ln1w = t.add(p + "layernorm_before.weight", {d});
        o.ln1b = t.add(p + "layernorm_before.bias", {d});
        o.wqkv = t.add(p + "attention.attention.query.weight", {d, d});
        t.add(p + "attention.attention.key.weight", {d, d});
        t.add(p + "attention.attention.value.weight", {d, d});
        o.bqkv = t.add(p + "attention.attention.query.bias", {d});
        t.add(p + "attention.attention.key.bias", {d});
        t.add(p + "attention.attention.value.bias", {d});
        o.wo = t.add(p + "attention.output.dense.weight", {d, d});
        o.bo = t.add(p + "attention.output.dense.bias", {d});
        o.ln2w = t.add(p + "layernorm_after.weight", {d});
        o.ln2b = t.add(p + "layernorm_after.bias", {d});
        o.w1 = t.add(p + "intermediate.dense.weight", {inter, d});
        o.b1 = t.add(p + "intermediate.dense.bias", {inter});
        o.w2 = t.add(p + "output.dense.weight", {d, inter});
        o.b2 = t.add(p + "output.dense.bias", {d});
    } else {
        o.ln1w = t.add(p + "norm1.weight", {d});
        o.ln1b = t.add(p + "norm1.bias", {d});
        o.wqkv = t.add(p + "attn.qkv.weight", {3 * d, d});
        o.bqkv = t.add(p + "attn.qkv.bias", {3 * d});
        o.wo = t.add(p + "attn.proj.weight", {d, d});
        o.bo = t.add(p + "attn.proj.bias", {d});
        o.ln2w = t.add(p + "norm2.weight", {d});
        o.ln2b = t.add(p + "norm2.bias", {d});
        o.w1 = t.add(p + "mlp.fc1.weight", {inter, d});
        o.b1 = t.add(p + "mlp.fc1.bias", {inter});
        o.w2 = t.add(p + "mlp.fc2.weight", {d, inter});
        o.b2 = t.add(p + "mlp.fc2.bias", {d});
    }
    o.end = t.total;
    return o;
}

int alloc_stack(Arena& a, Stack& s, int D, int I, int H, int nlayers, float eps, size_t M, size_t BHN) {
    s.D = D; s.I = I; s.H = H; s.nlayers = nlayers; s.eps = eps;
    s.hd = D / H;
    s.hdp = s.hd <= 32 ? 32 : 64;
    BVC_REQUIRE(D % H == 0 && s.hd <= 64 && s.hd % 8 == 0, "stack: head_dim %d unsupported (multiples of 8 up to 64)", s.hd);
    s.Da = H * s.hdp;
    const size_t Da = s.Da;
    // operand extents travel as 32-bit byte counts (bvc_gemm_desc.a_bytes / b_bytes, buffer descriptors): the widest bf16
    // activation of a layer has to stay below 4 GiB (VideoMAE-base decoder: ~890 clips per GPU; from 2 GiB on - 445 clips - the
    // 256-row persistent kernel, which marks dropped loads with offset bit 31, hands the product to the 128 x 128 kernels)
    BVC_REQUIRE(M * std::max<size_t>(3 * Da, (size_t)I) * 2 < 0xFFFFFFF0ull,
                "stack: %zu tokens x %zu columns of bf16 exceed the 4 GiB operand extent; use a smaller per-GPU batch", M, std::max<size_t>(3 * Da, (size_t)I));
    if (s.hdp != s.hd) {
        TRY(a.alloc(&s.wqkv_pad, 3 * Da * D));
        TRY(a.alloc(&s.wo_pad, (size_t)D * Da));
        TRY(a.alloc(&s.bqkv_pad, 3 * Da));
        TRY(a.alloc(&s.gwqkv_pad, 3 * Da * D));
        TRY(a.alloc(&s.gbqkv_pad, 3 * Da));
        TRY(a.alloc(&s.gwo_pad, (size_t)D * Da));
    }
    s.act.resize(nlayers);
    for (auto& l : s.act) {
        TRY(a.alloc(&l.x_in, M * D));
        TRY(a.alloc(&l.h, M * D));
        TRY(a.alloc(&l.ln1o, M * D));
        TRY(a.alloc(&l.qkv, M * 3 * Da));
        TRY(a.alloc(&l.ctx, M * Da));
        TRY(a.alloc(&l.lse, BHN));
        TRY(a.alloc(&l.ln2o, M * D));
        TRY(a.alloc(&l.pre, M * I));
        TRY(a.alloc(&l.act, M * I));
        TRY(a.alloc(&l.mean1, M));
        TRY(a.alloc(&l.rstd1, M));
        TRY(a.alloc(&l.mean2, M));
        TRY(a.alloc(&l.rstd2, M));
    }
    TRY(a.alloc(&s.x_out, M * D));
    return BVC_OK;
}

int alloc_work(Arena& a, Work& w, size_t MD, size_t MI, size_t delta_elems, size_t lnpart_elems) {
    for (int i = 0; i < 3; ++i) TRY(a.alloc(&w.dyb[i], MD));
    for (int i = 0; i < 2; ++i) {
        TRY(a.alloc(&w.dhb[i], MD));
        TRY(a.alloc(&w.dqkv[i], 3 * MD));
        TRY(a.alloc(&w.dh[i], MI));
    }
    TRY(a.alloc(&w.dln, MD));
    TRY(a.alloc(&w.dctx, MD));
    TRY(a.alloc(&w.delta, delta_elems));
    TRY(a.alloc(&w.ln_part, std::max(lnpart_elems, kRowLnPartFloats)));      // (also the per-workgroup partial rows of the fused LayerNorm backward)
    // Measured on MI355X (B=16): running the grouped dW launch on a side stream next to the dX chain gains nothing
    // (1398 vs 1419 clips/s) - each GEMM already holds all of a CU's LDS - so it is opt-in for experiments.
    w.overlap = false;      // bvc_set_option("dw_overlap", 1), read at every begin_backward
    BVC_CHECK_HIP(hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking));
    BVC_CHECK_HIP(hipEventCreateWithFlags(&w.ev_fork, hipEventDisableTiming));
    BVC_CHECK_HIP(hipEventCreateWithFlags(&w.ev_join[0], hipEventDisableTiming));
    BVC_CHECK_HIP(hipEventCreateWithFlags(&w.ev_join[1], hipEventDisableTiming));
    return BVC_OK;
}

void free_work(Work& w) {
    if (w.side) { (void)hipStreamSynchronize(w.side); (void)hipStreamDestroy(w.side); w.side = nullptr; }
    if (w.ev_fork) { (void)hipEventDestroy(w.ev_fork); w.ev_fork = nullptr; }
    for (int i = 0; i < 2; ++i) if (w.ev_join[i]) { (void)hipEventDestroy(w.ev_join[i]); w.ev_join[i] = nullptr; }
}

void begin_backward(Work& w) {
    w.overlap = options().dw_overlap != 0;
    w.seq = 0;
    w.join_pending[0] = w.join_pending[1] = false;
}

GemmProblem gemm(const bf16_t* A, size_t a_elems, int lda, const bf16_t* B, size_t b_elems, int ldb, int M, int N, int K,
                 int epi, void* C, int ldc) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.a_bytes = (uint32_t)(a_elems * 2); p.b_bytes = (uint32_t)(b_elems * 2);
    p.alpha = 1.f; p.epi = epi; p.split_k = 1; p.C = C; p.ldc = ldc;
    return p;
}

// Tile and split-K choice for a group of weight-gradient products (contraction over all tokens).
// Measured (profiles/r01_b_microbench.json): when the 128x128 tiles alone cover the chip (encoder layer: 432)
// use them unsplit; otherwise 64x64 tiles with just enough K-splits for ~850 workgroups (decoder layer: 432 x 2).
int plan_dw(GemmProblem* g, int n) {
    // Largest tile that still gives about one full wave of blocks (256 CUs x 2 resident) once K is split; the split is capped
    // at 12 (each split adds one f32 atomic pass over the output) and at 8 K-steps per block.  Fitted to the same-box sweep
    // in profiles/r01_d_dw_sweep.txt (tools/ab/dw_sweep.py): dec layer group 128x128 split 4 = 125 us vs 64x64 split 2 = 155 us.
    static const int bm[3] = {128, 128, 64}, bn[3] = {128, 64, 64};
    int ksteps = 1 << 30;
    for (int i = 0; i < n; ++i) ksteps = std::min(ksteps, (g[i].K + 63) / 64);
    // The 256-row persistent kernel (gemm8.hip; tile config 10 = 256 x 256, 11 = 256 x 128): one workgroup per CU walks its units
    // (tile x K split) as one stream of K tiles, so the split is chosen for ~230 units on the 256 CUs, at least 16 K tiles each.
    // Same-process A/B at 64 clips (profiles/r02_e_gemm8_ab_b64.txt): encoder layer 256 x 256 split 2 (216 units) 181 us vs
    // 256 x 128 unsplit 218 us vs the 128 x 128 kernel 220 us; decoder layer 256 x 256 split 6 (228 units) 549 us vs 626 us.
    const int g8 = options().gemm8;
    if (g8 >= 0) {
        bool ok = true, by384 = true;
        double flops = 0.0, outs = 0.0;
        int units256 = 0, units384 = 0;
        for (int i = 0; i < n; ++i) {
            ok = ok && g[i].epi == EPI_F32 && g[i].a_bytes < 0x80000000u && g[i].b_bytes < 0x80000000u;
            by384 = by384 && g[i].N % 384 == 0 && g[i].M % 128 == 0;
            flops += 2.0 * g[i].M * g[i].N * g[i].K;
            outs += (double)g[i].M * g[i].N;
            units256 += ((g[i].M + 255) / 256) * ((g[i].N + 255) / 256);
            units384 += ((g[i].M + 127) / 128) * ((g[i].N + 383) / 384);
        }
        // Widths that are multiples of 384 but not of 256 (decoder, JEPA predictor: 384 / 1152 / 1536) leave 256 x 256 tiles partly
        // empty - the four weight gradients of a VideoMAE decoder layer fill 71 % of 38 such tiles and 100 % of 36 tiles of 128 x 384
        // (tile config 12).  Taken when the 256-wide tiling would execute >= 1.2x the MFMA work of the outputs.  K split: the fewest
        // splits that give every CU one unit (units x split as close below a multiple of 256 as it gets), at least 16 K tiles each.
        // (same-process A/B, profiles/r03_b_dw_tile12_ab_b{16,64,256}.txt: decoder layer group 1522 vs 2312 us at 256 clips, 383 vs
        //  558 at 64, 125 vs 139 (the 128 x 128 kernel) at 16 - it needs a shorter stream than the 256 x 256 tile to pay off)
        const bool long_enough = g8 > 0 ? ksteps >= 2 : (ksteps >= 16 && flops >= 80e9);
        if (ok && by384 && long_enough && units256 * 65536.0 >= 1.2 * outs) {
            // (up to 24 splits since round 4: the head's 12 tiles x 21 = 252 units, 498 vs 564 us with 16 splits at 256 clips -
            //  profiles/r04_o_head_dw_splits.txt; an atomic pass over its 2.4 MB of outputs per split is noise)
            const int smax = std::max(1, std::min(24, ksteps / (g8 > 0 ? 2 : 16)));
            int best = 1;
            double best_cost = 1e300;
            for (int sp = 1; sp <= smax; ++sp) {
                const int rounds = (units384 * sp + 255) / 256;
                const double cost = (double)rounds * ((ksteps + sp - 1) / sp) + 4.0 * rounds;     // K tiles on the critical path + an epilogue per round
                if (cost < best_cost * 0.999) { best_cost = cost; best = sp; }
            }
            if (units384 * best >= (g8 > 0 ? 1 : 160)) {
                for (int i = 0; i < n; ++i) g[i].split_k = best;
                return 12;
            }
        }
        if (ok && g8 > 0 && ksteps >= 2) {          // forced (tests, A/B tools): 256 x 256 tiles, K split for ~230 units as below
            const int split = std::max(1, std::min((232 + units256 / 2) / units256, ksteps / 2));
            for (int i = 0; i < n; ++i) g[i].split_k = split;
            return 10;
        }
        // short launches stay on the 128 x 128 kernel (at 16 clips: encoder layer 53 vs 85 us, decoder layer 139 vs 151 us,
        // profiles/r02_e_gemm8_ab_b16.txt): one workgroup per CU needs a long stream to amortise its prologue and tail
        if (ok && ksteps >= 16 && flops >= 140e9) {
            for (int t = 0; t < 2; ++t) {            // prefer the bigger tile when it still fills the chip
                const int bnw = t == 0 ? 256 : 128;
                int units = 0;
                for (int i = 0; i < n; ++i) units += ((g[i].M + 255) / 256) * ((g[i].N + bnw - 1) / bnw);
                int split = std::max(1, (232 + units / 2) / units);
                split = std::min(split, std::max(1, ksteps / 16));
                if (units >= 16 && units * split >= (t == 0 ? 200 : 160) && units * split <= 272) {
                    for (int i = 0; i < n; ++i) g[i].split_k = split;
                    return t == 0 ? 10 : 11;
                }
            }
            // 256 x 256 tiles that fill half to 15/16 of the chip unsplit and overflow it when split (ViT-L layers: 192 tiles): tile
            // config 13 - the outputs are accumulated by f32 atomics (every caller of plan_dw zeroes the gradient buffer first, as for a
            // K split), which lets gemm8's balanced walk hand the tails of the K ranges to the idle CUs (launch_gemm, plan_balance)
            if (units256 >= 128 && units256 <= 240 && ksteps >= 64) {
                for (int i = 0; i < n; ++i) g[i].split_k = 1;
                return 13;
            }
        }
    }
    const int cap = std::max(1, std::min(12, ksteps / 8));
    int tile = 2, split = 1;
    for (int t = 0; t < 3; ++t) {
        int tiles = 0;
        for (int i = 0; i < n; ++i) tiles += ((g[i].M + bm[t] - 1) / bm[t]) * ((g[i].N + bn[t] - 1) / bn[t]);
        const int s = std::max(1, std::min(cap, (480 + tiles / 2) / tiles));
        tile = t; split = s;
        if (tiles * s >= 400) break;
    }
    for (int i = 0; i < n; ++i) g[i].split_k = split;
    return tile;
}

// A 128 x 384 tile of gemm8.hip holds complete rows of a 384-wide Linear output, so the LayerNorm that follows proj / fc2 (forward) or
// precedes fc1 / qkv (backward: their dX products) runs in that product's epilogue instead of as an HBM pass of its own.  Measured at 256
// clips (profiles/r05_e_rowln_products_stagger_b256.txt): proj + LN 416 vs 330 + 173 us, fc2 + LN 693 vs 684 + 172, dX fc1 + LN bwd 866
// vs 562 + 472, dX qkv + LN bwd 749 vs 433 + 473; whole step -0.5 % at 16 clips ... -2.2 % at 256 (r05_h_rowln_ab_batches.txt).  Below
// 128 units (JEPA predictor at small batches) the separate passes stay.  bvc_set_option("row_ln", 1 / -1) forces either way; a forced gemm8 forces it too.
bool fuse_row_ln(const Stack& s, int M) {
    const int mode = options().row_ln;
    if (mode < 0 || options().gemm8 < 0) return false;
    if (!gemm_row_ln_ok(M, s.D, s.D) || s.I % 64 != 0 || (3 * s.Da) % 64 != 0) return false;
    if ((size_t)M * std::max<size_t>(3 * (size_t)s.Da, (size_t)s.I) * 2 >= 0x80000000ull) return false;    // gemm8 addresses operands below 2 GiB
    if (mode > 0 || options().gemm8 > 0) return true;
    return M >= 128 * 128;     // (16 clips of 1568 tokens = 196 units: -0.5 %; 32: -1.3 %; 64: -1.3 %; 128: -2.0 %; 256: -2.2 % of the step)
}

int layer_forward(Work& w, Stack& s, int li, const LayerOff& o, const float* x_in, float* x_out, int B, int N, hipStream_t st,
                  const LayerOff* next) {
    LayerAct& a = s.act[li];
    const int D = s.D, I = s.I, M = B * N;
    const float* P = w.params;
    const bf16_t* W = w.wbf;
    const float eps = s.eps;
    const bool fuse = fuse_row_ln(s, M);
    if (li == 0) s.ln1_ready = false;
    // (the previous layer's fc2 epilogue may have left this layer's first LayerNorm behind)
    if (!(fuse && s.ln1_ready)) TRY(launch_ln_fwd(x_in, identity_rows(), P + o.ln1w, P + o.ln1b, a.ln1o, a.mean1, a.rstd1, M, D, eps, st));
    s.ln1_ready = false;
    const int Da = s.Da;
    const bool pad = s.hdp != s.hd;
    const bf16_t *Wqkv = W + o.wqkv, *Wo = W + o.wo;
    const float* bqkv = P + o.bqkv;
    const float sm_scale = pad ? 1.0f / sqrtf((float)s.hd) : 0.f;
    if (pad) {
        TRY(launch_pad_heads(W + o.wqkv, P + o.bqkv, W + o.wo, s.wqkv_pad, s.bqkv_pad, s.wo_pad, D, s.H, s.hd, s.hdp, st));
        Wqkv = s.wqkv_pad; Wo = s.wo_pad; bqkv = s.bqkv_pad;
    }
    {
        GemmProblem p = gemm(a.ln1o, (size_t)M * D, D, Wqkv, (size_t)3 * Da * D, D, M, 3 * Da, D, EPI_BF16, a.qkv, 3 * Da);
        p.bias = bqkv;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    TRY(launch_attn_fwd(a.qkv, a.ctx, a.lse, B, N, s.H, s.hdp, st, sm_scale));
    {
        GemmProblem p = gemm(a.ctx, (size_t)M * Da, Da, Wo, (size_t)D * Da, Da, M, D, Da, fuse ? EPI_RESID_LN : EPI_RESID, a.h, D);
        p.bias = P + o.bo; p.resid = x_in;
        if (fuse) { p.C2 = a.ln2o; p.ln_gamma = P + o.ln2w; p.ln_beta = P + o.ln2b; p.ln_mean = a.mean2; p.ln_rstd = a.rstd2; p.ln_eps = eps; }
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    if (!fuse) TRY(launch_ln_fwd(a.h, identity_rows(), P + o.ln2w, P + o.ln2b, a.ln2o, a.mean2, a.rstd2, M, D, eps, st));
    {
        GemmProblem p = gemm(a.ln2o, (size_t)M * D, D, W + o.w1, (size_t)I * D, D, M, I, D, EPI_GELU, a.pre, I);
        p.bias = P + o.b1; p.C2 = a.act;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    {
        const bool fuse_next = fuse && next != nullptr && li + 1 < (int)s.act.size();
        GemmProblem p = gemm(a.act, (size_t)M * I, I, W + o.w2, (size_t)D * I, I, M, D, I, fuse_next ? EPI_RESID_LN : EPI_RESID, x_out, D);
        p.bias = P + o.b2; p.resid = a.h;
        if (fuse_next) {      // the next layer's first LayerNorm, out of this epilogue's complete rows
            LayerAct& an = s.act[li + 1];
            p.C2 = an.ln1o; p.ln_gamma = P + next->ln1w; p.ln_beta = P + next->ln1b; p.ln_mean = an.mean1; p.ln_rstd = an.rstd1; p.ln_eps = eps;
        }
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
        s.ln1_ready = fuse_next;
    }
    return BVC_OK;
}

// Fence a side-stream weight-gradient launch into the main stream and report its gradient range.
int join_side(Work& c_, int parity, hipStream_t st, bvc_bucket_fn on_bucket, void* user) {
    if (!c_.join_pending[parity]) return BVC_OK;
    BVC_CHECK_HIP(hipStreamWaitEvent(st, c_.ev_join[parity], 0));
    c_.join_pending[parity] = false;
    if (on_bucket) on_bucket(c_.pend_lo[parity], c_.pend_hi[parity] - c_.pend_lo[parity], user);
    return BVC_OK;
}

// dres (f32 [M][D]) holds d/d(layer output) on entry and d/d(layer input) on exit; dyb[seq % 3] is its bf16 copy.
// The four weight gradients (+ bias gradients) of the layer are one grouped launch on the side stream, overlapping the
// next layer's dX chain; its gradient range [o.ln1w, o.end) is reported when that launch has been fenced (two steps later).
int layer_backward(Work& c_, Stack& s, int li, const LayerOff& o, const float* x_in, float* dres, float* G, int B, int N,
                   hipStream_t st, bvc_bucket_fn on_bucket, void* user) {
    LayerAct& a = s.act[li];
    const int D = s.D, I = s.I, M = B * N;
    const float* P = c_.params;
    const bf16_t* W = c_.wbf;
    const int q = c_.seq, par = q & 1;
    bf16_t* dyb = c_.dyb[q % 3];
    bf16_t* dyb_next = c_.dyb[(q + 1) % 3];
    bf16_t *dh = c_.dh[par], *dhb = c_.dhb[par], *dqkv = c_.dqkv[par];
    // the buffers of this parity were last read by the side launch of step q-2
    TRY(join_side(c_, par, st, on_bucket, user));
    // MLP
    {
        GemmProblem p = gemm(dyb, (size_t)M * D, D, W + o.w2, (size_t)D * I, I, M, I, D, EPI_DGELU, dh, I);
        p.aux = a.pre; p.ldaux = I;
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    const bool fuse = fuse_row_ln(s, M);
    if (fuse) {      // dX of fc1 with the second LayerNorm's backward in its epilogue: dres += ..., dhb = bf16(dres), dgamma / dbeta
        GemmProblem p = gemm(dh, (size_t)M * I, I, W + o.w1, (size_t)I * D, D, M, D, I, EPI_DLN, dres, D);
        p.C2 = dhb; p.ln_x = a.h; p.ln_mean = a.mean2; p.ln_rstd = a.rstd2; p.ln_gamma = P + o.ln2w;
        p.ln_part = c_.ln_part; p.ln_dgamma = G + o.ln2w; p.ln_dbeta = G + o.ln2b;
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    } else {
        GemmProblem p = gemm(dh, (size_t)M * I, I, W + o.w1, (size_t)I * D, D, M, D, I, EPI_BF16, c_.dln, D);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
        TRY(launch_ln_bwd(c_.dln, a.h, identity_rows(), a.mean2, a.rstd2, P + o.ln2w, dres, 1, dhb, G + o.ln2w, G + o.ln2b, c_.ln_part, M, D, st));
    }
    // attention
    const int Da = s.Da;
    const bool pad = s.hdp != s.hd;
    const bf16_t *Wqkv = W + o.wqkv, *Wo = W + o.wo;
    const float sm_scale = pad ? 1.0f / sqrtf((float)s.hd) : 0.f;
    if (pad) {   // the padded weight copies are shared by all layers: rebuild this layer's
        TRY(launch_pad_heads(W + o.wqkv, P + o.bqkv, W + o.wo, s.wqkv_pad, s.bqkv_pad, s.wo_pad, D, s.H, s.hd, s.hdp, st));
        Wqkv = s.wqkv_pad; Wo = s.wo_pad;
    }
    {
        GemmProblem p = gemm(dhb, (size_t)M * D, D, Wo, (size_t)D * Da, Da, M, Da, D, EPI_BF16, c_.dctx, Da);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    TRY(launch_attn_bwd(a.qkv, a.ctx, c_.dctx, a.lse, c_.delta, dqkv, B, N, s.H, s.hdp, st, sm_scale));
    if (fuse) {      // dX of qkv with the first LayerNorm's backward in its epilogue: dres becomes d/d(layer input), dyb_next its bf16 copy
        GemmProblem p = gemm(dqkv, (size_t)M * 3 * Da, 3 * Da, Wqkv, (size_t)3 * Da * D, D, M, D, 3 * Da, EPI_DLN, dres, D);
        p.C2 = dyb_next; p.ln_x = x_in; p.ln_mean = a.mean1; p.ln_rstd = a.rstd1; p.ln_gamma = P + o.ln1w;
        p.ln_part = c_.ln_part; p.ln_dgamma = G + o.ln1w; p.ln_dbeta = G + o.ln1b;
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    } else {
        GemmProblem p = gemm(dqkv, (size_t)M * 3 * Da, 3 * Da, Wqkv, (size_t)3 * Da * D, D, M, D, 3 * Da, EPI_BF16, c_.dln, D);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    // the four weight gradients of the layer as one grouped launch:  dW = dY^T X,  db = column sums of dY
    hipStream_t ws = st;
    if (c_.overlap) {
        BVC_CHECK_HIP(hipEventRecord(c_.ev_fork, st));
        BVC_CHECK_HIP(hipStreamWaitEvent(c_.side, c_.ev_fork, 0));
        ws = c_.side;
    }
    {
        GemmProblem g[4];
        g[0] = gemm(dyb, (size_t)M * D, D, a.act, (size_t)M * I, I, D, I, M, EPI_F32, G + o.w2, I);
        g[1] = gemm(dh, (size_t)M * I, I, a.ln2o, (size_t)M * D, D, I, D, M, EPI_F32, G + o.w1, D);
        g[2] = gemm(dhb, (size_t)M * D, D, a.ctx, (size_t)M * Da, Da, D, Da, M, EPI_F32, pad ? s.gwo_pad : G + o.wo, Da);
        g[3] = gemm(dqkv, (size_t)M * 3 * Da, 3 * Da, a.ln1o, (size_t)M * D, D, 3 * Da, D, M, EPI_F32, pad ? s.gwqkv_pad : G + o.wqkv, D);
        g[0].rowsum = G + o.b2;     // bias gradients ride along as one extra MFMA column each
        g[1].rowsum = G + o.b1;
        g[2].rowsum = G + o.bo;
        g[3].rowsum = pad ? s.gbqkv_pad : G + o.bqkv;
        const int tile = plan_dw(g, 4);
        if (pad) {   // split-K accumulates with atomics: the padded scratch is zeroed like the gradient buffer is
            BVC_CHECK_HIP(hipMemsetAsync(s.gwo_pad, 0, (size_t)D * Da * 4, ws));
            BVC_CHECK_HIP(hipMemsetAsync(s.gwqkv_pad, 0, (size_t)3 * Da * D * 4, ws));
            BVC_CHECK_HIP(hipMemsetAsync(s.gbqkv_pad, 0, (size_t)3 * Da * 4, ws));
        }
        TRY(launch_gemm(g, 4, GEMM_TN, tile, ws));
        if (pad)
            TRY(launch_unpad_head_grads(s.gwqkv_pad, s.gbqkv_pad, s.gwo_pad, G + o.wqkv, G + o.bqkv, G + o.wo, D, s.H, s.hd, s.hdp, ws));
    }
    if (!fuse) TRY(launch_ln_bwd(c_.dln, x_in, identity_rows(), a.mean1, a.rstd1, P + o.ln1w, dres, 1, dyb_next, G + o.ln1w, G + o.ln1b, c_.ln_part, M, D, st));
    if (c_.overlap) {
        BVC_CHECK_HIP(hipEventRecord(c_.ev_join[par], c_.side));
        c_.join_pending[par] = true;
        c_.pend_lo[par] = o.ln1w;
        c_.pend_hi[par] = o.end;
    } else if (on_bucket) {
        on_bucket(o.ln1w, o.end - o.ln1w, user);
    }
    c_.seq = q + 1;
    return BVC_OK;
}

}  // namespace bvc
