// HBM-bound row / element kernels of the ViT step (gfx950): LayerNorm fwd/bwd, column sums for
// bias gradients, parameter cast, mask -> index lists, tube-patch gather, pixel targets, decoder
// input fill, loss finalize.  All are one-pass streaming kernels with 8-16 B per lane accesses.
#include "rowops.h"

#include <algorithm>

namespace bvc {

constexpr int kMaxChunks = 4;   // float4 chunks per lane: D <= 1024

__device__ __forceinline__ int map_row(int m, RowMap rm) {
    return rm.rin > 0 ? (m / rm.rin) * rm.rout + rm.roff + (m % rm.rin) : m;
}

// sum over the LPR lanes that share a row (64: the wave; 32: each half-wave holds its own row - xor distances < 32 stay inside a half)
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ============================================================================ LayerNorm forward
// LPR lanes per row: 64 (one wave per row) or 32 (two rows per wave: widths up to 512 that are multiples of 128 - the decoder's and
// the predictor's 384 - would leave half of the wave idle in the second of their 1.5 float4 chunks per lane); y = (x - mean) * rstd * gamma + beta in bf16; saves mean / rstd (biased variance,
// torch.nn.LayerNorm semantics, HF:336-337)
template <int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, RowMap rm, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y, float* __restrict__ y32,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
    constexpr int RPW = 64 / LPR, RPBK = 4 * RPW;        // rows per wave, rows per workgroup and pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rsel = lane / LPR;
    const int nch = D >> 2;
    // grid-stride over rows (the launch is capped at a few workgroups per CU): 100352 four-row workgroups per decoder LayerNorm
    // at 256 clips cost more in workgroup turnover than the rows take to stream
    for (int m0 = blockIdx.x * RPBK + wave * RPW; m0 < M; m0 += gridDim.x * RPBK) {
    const int m = m0 + rsel;
    const bool live = m < M;                             // (a half-wave past the last row idles; its shuffles stay in its own half)
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)map_row(live ? m : M - 1, rm) * D);
    f32x4 v[kMaxChunks];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = sub + LPR * i;
        if (c < nch) { v[i] = xr[c]; s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    const float mu = row_sum<LPR>(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = sub + LPR * i;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(row_sum<LPR>(q) / D + eps);
    if (!live) continue;
    if (sub == 0 && mean) { mean[m] = mu; rstd[m] = rs; }
    uint2* yr = y ? reinterpret_cast<uint2*>(y + (size_t)m * D) : nullptr;
    f32x4* yf = y32 ? reinterpret_cast<f32x4*>(y32 + (size_t)m * D) : nullptr;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = sub + LPR * i;
        if (c < nch) {
            f32x4 o = (v[i] - mu) * rs;
            if (gamma) o = o * reinterpret_cast<const f32x4*>(gamma)[c] + reinterpret_cast<const f32x4*>(beta)[c];
            if (yr) yr[c] = uint2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            if (yf) yf[c] = o;
        }
    }
    }
}

// ============================================================================ LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dres (+)= dx; optional bf16 copy;
// per-workgroup partials of dgamma = sum_rows dy * xhat and dbeta = sum_rows dy go to part[block][2][D]
// (every workgroup hammering the same D addresses with atomics serialises: measured 2x on the kernel);
// ln_param_reduce_kernel folds them into the gradients
template <int RPB, int LPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x, RowMap rm,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ dres,
                                                     int accumulate, bf16_t* __restrict__ dres_bf, float* __restrict__ part,
                                                     int M, int D) {
    // per-wave (per half-wave for LPR = 32) column partials, [4 RPW][dgamma | dbeta][D]: sized by D at launch (a static 32 KiB array
    // capped the kernel at 5 workgroups per CU; at D = 384 this is 24 KiB and the register budget decides: 6-7)
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rsel = lane / LPR;
    const int nch = D >> 2;
    f32x4 gam[kMaxChunks], dg[kMaxChunks], db[kMaxChunks];
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = sub + LPR * i;
        gam[i] = c < nch ? reinterpret_cast<const f32x4*>(gamma)[c] : f32x4{0, 0, 0, 0};
        dg[i] = f32x4{0, 0, 0, 0};
        db[i] = f32x4{0, 0, 0, 0};
    }
    const float invD = 1.f / D;
    // grid-stride over row groups of RPB rows: the launch is capped at a few workgroups per CU, so the column partials below are
    // written once per RESIDENT workgroup (2048 x 2 D floats) instead of once per RPB rows (25088 x 2 D at the decoder's 256-clip
    // size: 77 MB per LayerNorm backward, and a reduction kernel to match)
    const int ngroups = (M + RPB - 1) / RPB;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x)
    for (int it = 0; it < RPB / (4 * RPW); ++it) {
        const int mw = grp * RPB + (it * 4 + wave) * RPW;     // first row of this wave
        if (mw >= M) break;
        const int m = mw + rsel;
        const bool live = m < M;
        const int mc = live ? m : M - 1;
        const size_t xrow = (size_t)map_row(mc, rm) * D;
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + xrow);
        const uint2* dyr = reinterpret_cast<const uint2*>(dy + (size_t)mc * D);
        const float mu = mean[mc], rs = rstd[mc];
        f32x4 xh[kMaxChunks], g[kMaxChunks];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxChunks; ++i) {
            const int c = sub + LPR * i;
            if (c < nch) {
                const uint2 d = dyr[c];
                f32x4 dyv = {__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u),
                             __uint_as_float(d.y << 16), __uint_as_float(d.y & 0xffff0000u)};
                if (!live) dyv = f32x4{0, 0, 0, 0};      // a half-wave past the last row adds nothing to the column sums
                xh[i] = (xr[c] - mu) * rs;
                g[i] = dyv * gam[i];
                dg[i] += dyv * xh[i];
                db[i] += dyv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += g[i][e]; s2 += g[i][e] * xh[i][e]; }
            }
        }
        s1 = row_sum<LPR>(s1) * invD;
        s2 = row_sum<LPR>(s2) * invD;
        if (!live) continue;
        f32x4* dr = reinterpret_cast<f32x4*>(dres + xrow);
        uint2* db16 = dres_bf ? reinterpret_cast<uint2*>(dres_bf + xrow) : nullptr;
#pragma unroll
        for (int i = 0; i < kMaxChunks; ++i) {
            const int c = sub + LPR * i;
            if (c < nch) {
                f32x4 dx = (g[i] - s1 - xh[i] * s2) * rs;
                if (accumulate) dx += dr[c];
                dr[c] = dx;
                if (db16) db16[c] = uint2{pack2bf(dx[0], dx[1]), pack2bf(dx[2], dx[3])};
            }
        }
    }
    // cross-wave reduction of the column partials (4 RPW of them), then one partial row per workgroup
    const int slot = wave * RPW + rsel;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = sub + LPR * i;
        if (c < nch) {
            *reinterpret_cast<f32x4*>(red + (slot * 2 + 0) * D + c * 4) = dg[i];
            *reinterpret_cast<f32x4*>(red + (slot * 2 + 1) * D + c * 4) = db[i];
        }
    }
    __syncthreads();
    float* pg = part + (size_t)blockIdx.x * 2 * D;
    for (int col = threadIdx.x; col < D; col += 256) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 4 * RPW; ++w) { a += red[(2 * w) * D + col]; b += red[(2 * w + 1) * D + col]; }
        pg[col] = a;
        pg[D + col] = b;
    }
}

// dgamma[c] += sum_b part[b][0][c], dbeta[c] += sum_b part[b][1][c]; grid (ceil(2D/256), ceil(nblk/16))
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, int nblk, int D,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * D) return;
    const int b0 = blockIdx.y * 16, b1 = min(nblk, b0 + 16);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = b0 + i < b1 ? part[(size_t)(b0 + i) * 2 * D + c] : 0.f;   // 16 loads in flight
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    atomicAdd(c < D ? dgamma + c : dbeta + (c - D), s);
}

// ============================================================================ column sums
// out[n] += alpha * sum_m X[m][n]   (bias gradients), X bf16 [M][ld]; grid (ceil(N/512), ceil(M/RB))
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, int M, int N, int ld, float alpha,
                                                          const float* __restrict__ alpha_dev, float* __restrict__ out, int RB) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int r0 = blockIdx.y * RB, r1 = min(M, r0 + RB);
    if (col < N) {
        for (int m = r0 + wave; m < r1; m += 4) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (size_t)m * ld + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f((bf16_t)v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int n = blockIdx.x * 512 + c;
        if (n < N) atomicAdd(out + n, (alpha_dev ? alpha * alpha_dev[0] : alpha) * ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])));
    }
}

// out[n] += sum over mapped rows of X f32 [..][D]   (mask-token gradient); grid (ceil(D/256), ceil(M/RB))
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ X, RowMap rm, int M, int D,
                                                         float* __restrict__ out, int RB) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 4;
    f32x4 acc = {0, 0, 0, 0};
    const int r0 = blockIdx.y * RB, r1 = min(M, r0 + RB);
    if (col < D)
        for (int m = r0 + wave; m < r1; m += 4)
            acc += *reinterpret_cast<const f32x4*>(X + (size_t)map_row(m, rm) * D + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;
    const int n = blockIdx.x * 256 + c;
    if (n < D) atomicAdd(out + n, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
}

// out[b][d] = mean over the N tokens of clip b (sequence_output.mean(1), HF VideoMAEForVideoClassification.forward);
// grid (ceil(D/64), B); thread = 4 columns x one of 16 row phases; fixed summation order (no atomics)
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, int N, int D, float* __restrict__ out) {
    __shared__ float red[16][64];
    const int cg = threadIdx.x & 15, rp = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cg * 4;
    const float* xb = x + (size_t)blockIdx.y * N * D;
    f32x4 acc = {0, 0, 0, 0};
    if (col < D)
        for (int m = rp; m < N; m += 16) acc += *reinterpret_cast<const f32x4*>(xb + (size_t)m * D + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rp][cg * 4 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
        const int n = blockIdx.x * 64 + threadIdx.x;
        if (n < D) out[(size_t)blockIdx.y * D + n] = t / (float)N;
    }
}

// dx[b][m][d] = dmean[b][d] / N   (backward of token_mean); one f32x4 per thread
__global__ void token_mean_bwd_kernel(const float* __restrict__ dmean, float* __restrict__ dx, int N, int D, size_t total4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const size_t e = i * 4;
    const int d = (int)(e % D);
    const size_t b = e / ((size_t)N * D);
    f32x4 g = *reinterpret_cast<const f32x4*>(dmean + b * D + d);
    const float inv = 1.f / (float)N;
    g *= inv;
    *reinterpret_cast<f32x4*>(dx + e) = g;
}

// ============================================================================ elementwise
__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(in + i);
        const f32x4 b = *reinterpret_cast<const f32x4*>(in + i + 4);
        *reinterpret_cast<uint4*>(out + i) = uint4{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    } else {
        for (size_t j = i; j < n; ++j) out[j] = f2bf(in[j]);
    }
}

// out bf16 [M][D] = in f32 [map_row(m)][D]
__global__ void gather_rows_bf16_kernel(const float* __restrict__ in, RowMap rm, bf16_t* __restrict__ out, int M, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 item
    const int nch = D >> 2;
    if (i >= (size_t)M * nch) return;
    const int m = (int)(i / nch), c = (int)(i % nch);
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + (size_t)map_row(m, rm) * D + c * 4);
    *reinterpret_cast<uint2*>(out + (size_t)m * D + c * 4) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}

// ============================================================================ mask -> ordered index lists
// one wave per clip: ascending token order for both lists (boolean-mask gather order, HF:121,578-579,661)
__global__ void mask_index_kernel(const uint8_t* __restrict__ mask, int L, int nvis, int nmask, int* __restrict__ vis_idx,
                                  int* __restrict__ msk_idx, int* __restrict__ status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int nv = 0, nm = 0;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int base = 0; base < L; base += 64) {
        const int t = base + lane;
        const bool in = t < L;
        const bool mk = in && mask[(size_t)b * L + t] != 0;
        const unsigned long long bm = __ballot(mk), bv = __ballot(in && !mk);
        if (mk) { const int p = nm + __popcll(bm & below); if (p < nmask) msk_idx[(size_t)b * nmask + p] = t; }
        if (in && !mk) { const int p = nv + __popcll(bv & below); if (p < nvis) vis_idx[(size_t)b * nvis + p] = t; }
        nm += __popcll(bm);
        nv += __popcll(bv);
    }
    if (lane == 0 && (nv != nvis || nm != nmask)) atomicOr(status, 1);
}

// four consecutive pixels of channel c starting at element `src` (a multiple of 4)
__device__ __forceinline__ f32x4 load_pixels4(const PixelSrc& px, size_t src, int c) {
    if (!px.is_u8) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(px.ptr) + src);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(px.ptr) + src);
    const float mu = px.mean[c & 3], sd = px.stdv[c & 3];
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((float)((w >> (8 * j)) & 0xffu) / 255.f - mu) / sd;
    return v;
}

// ============================================================================ tube-patch gather (visible tokens only)
// A[m][k], k = ((c*ts + dt)*ps + dy)*ps + dx  (Conv3d weight order, HF:157-162), clip f32 [B][T][C][H][W]
__global__ void gather_patches_kernel(const PixelSrc clip, const int* __restrict__ vis_idx, bf16_t* __restrict__ A,
                                      int B, int nvis, PatchGeom pg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = pg.C * pg.ts * pg.ps * pg.ps;
    const int k8 = K >> 3;
    if (i >= (size_t)B * nvis * k8) return;
    const int m = (int)(i / k8);
    const int k = (int)(i % k8) * 8;
    const int dx = k % pg.ps, dy = (k / pg.ps) % pg.ps, dt = (k / (pg.ps * pg.ps)) % pg.ts, c = k / (pg.ps * pg.ps * pg.ts);
    const int b = m / nvis, tok = vis_idx[m];
    const int wp = pg.W / pg.ps, hp = pg.H / pg.ps;
    const int tp = tok / (hp * wp), yp = (tok / wp) % hp, xp = tok % wp;
    const size_t src = ((((size_t)b * pg.T + tp * pg.ts + dt) * pg.C + c) * pg.H + yp * pg.ps + dy) * pg.W + xp * pg.ps + dx;
    const f32x4 a = load_pixels4(clip, src, c);
    const f32x4 d = load_pixels4(clip, src + 4, c);
    *reinterpret_cast<uint4*>(A + (size_t)m * K + k) =
        uint4{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(d[0], d[1]), pack2bf(d[2], d[3])};
}

// ============================================================================ pixel targets (HF:588-661)
// one workgroup per masked token: un-normalise, per-channel mean / unbiased variance over ts*ps*ps
// values, labels[(dt,dy,dx,c)] = (f - mean) / (sqrt(var) + 1e-6)
__global__ __launch_bounds__(256) void labels_kernel(const PixelSrc clip, const int* __restrict__ msk_idx,
                                                     float* __restrict__ labels, int nmask, PatchGeom pg, int norm_pix) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* buf = reinterpret_cast<float*>(smem);   // [C][E]
    const int E = pg.ts * pg.ps * pg.ps, C = pg.C;
    float* stat = buf + C * E;                     // [C][2]
    // (one workgroup per token on purpose: a grid-stride form with 4096 workgroups measured 1565 vs 1413 us at 256 clips - the
    //  load -> barrier -> statistics -> barrier -> store chain of a token hides its latency only behind OTHER workgroups)
    // XCD-aware order (round 4): a patch row is 64 B, half of a 128-B line whose other half belongs to the next token in x - usually the
    // next masked token of the clip.  Workgroups b and b + 1 run on different XCDs (different L2s), so with m = blockIdx.x both fetched
    // the line: 4.45 GB read for 2.2 GB of pixels at 256 clips.  Each XCD takes a contiguous run of tokens instead.
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7;
    const int m = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    const int b = m / nmask, tok = msk_idx[m];
    const int wp = pg.W / pg.ps, hp = pg.H / pg.ps;
    const int tp = tok / (hp * wp), yp = (tok / wp) % hp, xp = tok % wp;
    const float mean3[3] = {0.485f, 0.456f, 0.406f}, std3[3] = {0.229f, 0.224f, 0.225f};
    const int e4n = E >> 2;
    for (int i = threadIdx.x; i < C * e4n; i += 256) {
        const int c = i / e4n, e = (i % e4n) * 4;
        const int dx = e % pg.ps, dy = (e / pg.ps) % pg.ps, dt = e / (pg.ps * pg.ps);
        const size_t src = ((((size_t)b * pg.T + tp * pg.ts + dt) * C + c) * pg.H + yp * pg.ps + dy) * pg.W + xp * pg.ps + dx;
        f32x4 v = load_pixels4(clip, src, c);
        if (C == 3) v = v * std3[c] + mean3[c];
        *reinterpret_cast<f32x4*>(buf + c * E + e) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C; c += 4) {
        float s = 0.f;
        for (int e = lane; e < E; e += 64) s += buf[c * E + e];
        const float mu = wave_sum(s) / E;
        float q = 0.f;
        for (int e = lane; e < E; e += 64) { const float d = buf[c * E + e] - mu; q += d * d; }
        const float var = wave_sum(q) / (E - 1);
        if (lane == 0) { stat[2 * c] = norm_pix ? mu : 0.f; stat[2 * c + 1] = norm_pix ? sqrtf(var) + 1e-6f : 1.f; }
    }
    __syncthreads();
    float* out = labels + (size_t)m * C * E;
    for (int o = threadIdx.x * 4; o < C * E; o += 1024) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = (o + j) / C, c = (o + j) % C;
            r[j] = (buf[c * E + e] - stat[2 * c]) / stat[2 * c + 1];
        }
        *reinterpret_cast<f32x4*>(out + o) = r;
    }
}

// The same targets, one WAVE per masked token (round 4).  The workgroup-per-token form above keeps 8 tokens (48 KB of reads) in flight
// per CU and chains load -> barrier -> statistics -> barrier -> store inside each: 1.41 ms at 256 clips for 4.7 GB = 3.3 TB/s.  Here a
// token's 1536 values are loaded by one wave (NCH 16-byte chunks per lane, all in flight together), parked in the wave's own LDS rows
// and never meet a workgroup barrier; 24 tokens per CU are in flight.  Same arithmetic in the same order as labels_kernel (per channel:
// lane-strided sums, wave_sum, two passes), so the labels are bit-identical.  Tokens are taken XCD by XCD like above.
template <int NCH>
__global__ __launch_bounds__(256, 6) void labels_wave_kernel(const PixelSrc clip, const int* __restrict__ msk_idx,
                                                          float* __restrict__ labels, int ntok, int nmask, PatchGeom pg, int norm_pix) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = pg.ts * pg.ps * pg.ps, C = pg.C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = C * E + 2 * C + ((4 - ((C * E + 2 * C) & 3)) & 3);         // floats, kept a multiple of 16 bytes
    float* buf = reinterpret_cast<float*>(smem) + wave * per_wave;                   // [C][E]
    float* stat = buf + C * E;                                                       // [C][2]
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7;
    const int blk = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    const int m = blk * 4 + wave;
    if (m >= ntok) return;                       // whole waves leave; nothing below synchronises across waves
    const int b = m / nmask, tok = msk_idx[m];
    const int wp = pg.W / pg.ps, hp = pg.H / pg.ps;
    const int tp = tok / (hp * wp), yp = (tok / wp) % hp, xp = tok % wp;
    const float mean3[3] = {0.485f, 0.456f, 0.406f}, std3[3] = {0.229f, 0.224f, 0.225f};
    const int e4n = E >> 2, nchunk = C * e4n;
    f32x4 v[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = lane + 64 * k;
        if (i < nchunk) {
            const int c = i / e4n, e = (i % e4n) * 4;
            const int dx = e % pg.ps, dy = (e / pg.ps) % pg.ps, dt = e / (pg.ps * pg.ps);
            const size_t src = ((((size_t)b * pg.T + tp * pg.ts + dt) * C + c) * pg.H + yp * pg.ps + dy) * pg.W + xp * pg.ps + dx;
            v[k] = load_pixels4(clip, src, c);
        }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = lane + 64 * k;
        if (i < nchunk) {
            const int c = i / e4n, e = (i % e4n) * 4;
            f32x4 w = v[k];
            if (C == 3) w = w * std3[c] + mean3[c];
            *reinterpret_cast<f32x4*>(buf + c * E + e) = w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
        for (int e = lane; e < E; e += 64) s += buf[c * E + e];
        const float mu = wave_sum(s) / E;
        float q = 0.f;
        for (int e = lane; e < E; e += 64) { const float d = buf[c * E + e] - mu; q += d * d; }
        const float var = wave_sum(q) / (E - 1);
        if (lane == 0) { stat[2 * c] = norm_pix ? mu : 0.f; stat[2 * c + 1] = norm_pix ? sqrtf(var) + 1e-6f : 1.f; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* out = labels + (size_t)m * C * E;
    for (int o = lane * 4; o < C * E; o += 256) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = (o + j) / C, c = (o + j) % C;
            r[j] = (buf[c * E + e] - stat[2 * c]) / stat[2 * c + 1];
        }
        *reinterpret_cast<f32x4*>(out + o) = r;
    }
}

// Three channels, E a multiple of 256 (the reference's 2 x 16 x 16 tubes: E = 512) - the same wave-per-token form with the arithmetic
// cut down (round 4): the strip experiment (profiles/r04_y_labels_strip_kernel.patch.txt) showed that a token cost a wave ~20 us of
// vector work, not of memory time.  Here (a) the statistics come from the lane's own registers (a lane's chunks k = 2c, 2c + 1 are channel
// c: eight values per channel, two-pass mean / unbiased variance with one wave_sum each) instead of 48 LDS reads; (b) C is a constant, so
// the (element, channel) of an output position costs a multiply-shift instead of a runtime division (24 of them per lane before);
// (c) one IEEE division per channel (1 / (sqrt(var) + 1e-6)) and a multiply per value instead of a division per value: within one f32
// rounding of (f - mean) / (sqrt(var) + 1e-6).  The per-lane channel of output position 4 lane + 256 k + j is (lane + k + j) mod 3, so
// with the channel statistics rotated by lane mod 3 once, every use is a compile-time register.
template <int NE>      // NE = E / 256: chunks per lane and channel
__global__ __launch_bounds__(256, 6) void labels_wave3_kernel(const PixelSrc clip, const int* __restrict__ msk_idx, float* __restrict__ labels,
                                                              int ntok, int nmask, PatchGeom pg, int norm_pix) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = 3, E = 256 * NE, CE = C * E, NCH = C * NE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* buf = reinterpret_cast<float*>(smem) + wave * CE;                          // [C][E], this wave's token
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7;
    const int blk = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    const int m = blk * 4 + wave;
    if (m >= ntok) return;
    const int b = m / nmask, tok = msk_idx[m];
    const int wp = pg.W / pg.ps, hp = pg.H / pg.ps;
    const int tp = tok / (hp * wp), yp = (tok / wp) % hp, xp = tok % wp;
    const float mean3[3] = {0.485f, 0.456f, 0.406f}, std3[3] = {0.229f, 0.224f, 0.225f};
    f32x4 v[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        constexpr int e4n = E / 4;
        const int i = lane + 64 * k;
        const int c = k / NE, e = (i - c * e4n) * 4;                 // (64 NE chunks per channel: chunk k of a lane is channel k / NE)
        const int dx = e % pg.ps, dy = (e / pg.ps) % pg.ps, dt = e / (pg.ps * pg.ps);
        const size_t src = ((((size_t)b * pg.T + tp * pg.ts + dt) * C + c) * pg.H + yp * pg.ps + dy) * pg.W + xp * pg.ps + dx;
        v[k] = load_pixels4(clip, src, c);
    }
    float mu[3], rinv[3];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float sm = 0.f;
#pragma unroll
        for (int k = c * NE; k < (c + 1) * NE; ++k) {
            v[k] = v[k] * std3[c] + mean3[c];
            *reinterpret_cast<f32x4*>(buf + c * E + (lane + 64 * (k - c * NE)) * 4) = v[k];
            sm += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
        const float mean = wave_sum(sm) * (1.0f / E);
        float q = 0.f;
#pragma unroll
        for (int k = c * NE; k < (c + 1) * NE; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[k][j] - mean; q += d * d; }
        const float var = wave_sum(q) / (float)(E - 1);
        mu[c] = norm_pix ? mean : 0.f;
        rinv[c] = norm_pix ? 1.0f / (sqrtf(var) + 1e-6f) : 1.f;
    }
    // statistics rotated by lane mod 3: mr[i] = mu[(lane + i) mod 3]
    const int l3 = lane % 3;
    float mr[3], rr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = l3 + i;      // 0 .. 4
        mr[i] = (c == 0 || c == 3) ? mu[0] : (c == 1 || c == 4) ? mu[1] : mu[2];
        rr[i] = (c == 0 || c == 3) ? rinv[0] : (c == 1 || c == 4) ? rinv[1] : rinv[2];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* out = labels + (size_t)m * CE;
#pragma unroll
    for (int k = 0; k < CE / 256; ++k) {
        const int o = lane * 4 + 256 * k;
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = o + j, e = t / 3, c = t - 3 * e;           // output position t = e * C + c
            r[j] = (buf[c * E + e] - mr[(k + j) % 3]) * rr[(k + j) % 3];
        }
        *reinterpret_cast<f32x4*>(out + o) = r;
    }
}

// x_full[b][nvis + j][:] = mask_token + pos[msk_idx[b][j]]   (HF:580-582)
__global__ void fill_masked_kernel(float* __restrict__ xfull, const float* __restrict__ mask_token, const float* __restrict__ pos,
                                   const int* __restrict__ msk_idx, int B, int L, int nvis, int nmask, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = D >> 2;
    if (i >= (size_t)B * nmask * nch) return;
    const int m = (int)(i / nch), c = (int)(i % nch);
    const int b = m / nmask, j = m % nmask;
    const f32x4 mt = reinterpret_cast<const f32x4*>(mask_token)[c];
    const f32x4 pe = reinterpret_cast<const f32x4*>(pos + (size_t)msk_idx[m] * D)[c];
    reinterpret_cast<f32x4*>(xfull + ((size_t)b * L + nvis + j) * D)[c] = mt + pe;
}

// loss = sum(partials) / count, fixed summation order; NaN if the mask index kernel flagged a bad mask
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partial, int n, double count,
                                                            const int* __restrict__ status, float* __restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (status && *status) ? __int_as_float(0x7fc00000) : (float)(red[0] / count);
}

// ============================================================================ JEPA pieces
// token gather with an index list: logical row m of a [nsets*B*n] output reads x[(b*L + idx[m])] where b = (m / n) % B;
// used as RowGather in the target selection below
// targets (pretrain_jepa.py:384-392): h = layer_norm(target_encoder(imgs)) without affine, rows picked by the 4 prediction
// masks, set-major then sample:  out[(i*B + b)*Np + j] = LN(h[b*L + idx[(i*B + b)*Np + j]]).  One wave per row.
__global__ __launch_bounds__(256) void target_select_kernel(const float* __restrict__ h, const int* __restrict__ idx,
                                                            float* __restrict__ out, int rows, int B, int Np, int L, int D, float eps) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const int b = (m / Np) % B;
    const f32x4* xr = reinterpret_cast<const f32x4*>(h + ((size_t)b * L + idx[m]) * D);
    const int nch = D >> 2;
    f32x4 v[kMaxChunks];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) { v[i] = xr[c]; s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)m * D);
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) o[c] = (v[i] - mu) * rs;
    }
}

// predictor input (vision_transformer.py:507-524): sequence s = i*B + b is [ctx tokens of sample b ; mask_token + pos[pred idx]]
__global__ void pred_assemble_kernel(const float* __restrict__ xe, const float* __restrict__ mask_token, const float* __restrict__ pos,
                                     const int* __restrict__ idx_pred, float* __restrict__ X, int nseq, int B, int Nc, int Np, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = D >> 2, T = Nc + Np;
    if (i >= (size_t)nseq * T * nch) return;
    const int c = (int)(i % nch);
    const int t = (int)((i / nch) % T), sq = (int)(i / ((size_t)nch * T));
    f32x4 v;
    if (t < Nc) {
        v = reinterpret_cast<const f32x4*>(xe + ((size_t)(sq % B) * Nc + t) * D)[c];
    } else {
        v = reinterpret_cast<const f32x4*>(mask_token)[c] + reinterpret_cast<const f32x4*>(pos + (size_t)idx_pred[(size_t)sq * Np + (t - Nc)] * D)[c];
    }
    reinterpret_cast<f32x4*>(X + ((size_t)sq * T + t) * D)[c] = v;
}

// gradient of the replicated context tokens: dxe[b*Nc + t] = sum_i dX[(i*B + b)][t]  -> bf16 (GEMM operand)
__global__ void pred_ctx_grad_kernel(const float* __restrict__ dX, bf16_t* __restrict__ dxe, int nsets, int B, int Nc, int Np, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = D >> 2, T = Nc + Np;
    if (i >= (size_t)B * Nc * nch) return;
    const int c = (int)(i % nch);
    const int t = (int)((i / nch) % Nc), b = (int)(i / ((size_t)nch * Nc));
    f32x4 a = {0, 0, 0, 0};
    for (int k = 0; k < nsets; ++k) a += reinterpret_cast<const f32x4*>(dX + (((size_t)k * B + b) * T + t) * D)[c];
    reinterpret_cast<uint2*>(dxe + ((size_t)b * Nc + t) * D)[c] = uint2{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
}

// smooth-L1 (beta = 1, mean; pretrain_jepa.py:399-402): per-block partial sums, folded by loss_finalize_kernel
__global__ __launch_bounds__(256) void smooth_l1_fwd_kernel(const float* __restrict__ z, const float* __restrict__ h, size_t n,
                                                            float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(z + i) - *reinterpret_cast<const f32x4*>(h + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float a = fabsf(d[e]); s += a < 1.f ? 0.5f * a * a : a - 0.5f; }
        } else {
            for (size_t j = i; j < n; ++j) { const float a = fabsf(z[j] - h[j]); s += a < 1.f ? 0.5f * a * a : a - 0.5f; }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dz = gout / n * clamp(z - h, -1, 1)
__global__ void smooth_l1_bwd_kernel(const float* __restrict__ z, const float* __restrict__ h, const float* __restrict__ gout,
                                     size_t n, float* __restrict__ dz) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float g = gout[0] / (float)n;
    if (i + 4 <= n) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(z + i) - *reinterpret_cast<const f32x4*>(h + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g * fminf(fmaxf(d[e], -1.f), 1.f);
        *reinterpret_cast<f32x4*>(dz + i) = o;
    } else {
        for (size_t j = i; j < n; ++j) dz[j] = g * fminf(fmaxf(z[j] - h[j], -1.f), 1.f);
    }
}

// momentum update of the target encoder (pretrain_jepa.py:426-432): k = m * k + (1 - m) * q over a flat range
__global__ void ema_kernel(float* __restrict__ k, const float* __restrict__ q, size_t n, float m) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        const f32x4 kv = *reinterpret_cast<f32x4*>(k + i), qv = *reinterpret_cast<const f32x4*>(q + i);
        *reinterpret_cast<f32x4*>(k + i) = kv * m + qv * (1.f - m);
    } else {
        for (size_t j = i; j < n; ++j) k[j] = k[j] * m + q[j] * (1.f - m);
    }
}

// idx[b*L + t] = t  (identity token list: "all tokens" for the target encoder)
__global__ void iota_mod_kernel(int* __restrict__ idx, int n, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i % L;
}

// ============================================================================ SimCLR loss pieces
// fn = f / max(||f||, eps) in bf16 (operand of the similarity GEMM), inv[i] = 1 / max(||f_i||, eps);
// F.cosine_similarity semantics (pretraining/contrastive/pretrain_simclr.py:116).  One wave per row.
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ f, bf16_t* __restrict__ fn,
                                                            float* __restrict__ inv, int n, int p, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const f32x4* fr = reinterpret_cast<const f32x4*>(f + (size_t)row * p);
    float ss = 0.f;
    for (int c = lane; c < p / 4; c += 64) { const f32x4 v = fr[c]; ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
    const float r = 1.f / fmaxf(sqrtf(wave_sum(ss)), eps);
    if (lane == 0) inv[row] = r;
    uint2* o = reinterpret_cast<uint2*>(fn + (size_t)row * p);
    for (int c = lane; c < p / 4; c += 64) { const f32x4 v = fr[c] * r; o[c] = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])}; }
}

// df = inv * (dfn - fhat * (fhat . dfn)),  fhat = f * inv   (backward of the row normalisation)
__global__ __launch_bounds__(256) void row_normalize_bwd_kernel(const float* __restrict__ f, const float* __restrict__ inv,
                                                                const float* __restrict__ dfn, float* __restrict__ df, int n, int p) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const f32x4* fr = reinterpret_cast<const f32x4*>(f + (size_t)row * p);
    const f32x4* gr = reinterpret_cast<const f32x4*>(dfn + (size_t)row * p);
    const float r = inv[row];
    float dot = 0.f;
    for (int c = lane; c < p / 4; c += 64) { const f32x4 a = fr[c] * r, g = gr[c]; dot += a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3]; }
    dot = wave_sum(dot);
    f32x4* o = reinterpret_cast<f32x4*>(df + (size_t)row * p);
    for (int c = lane; c < p / 4; c += 64) { const f32x4 a = fr[c] * r; o[c] = (gr[c] - a * dot) * r; }
}

// loss = logsumexp(negatives) - mean(positives) from the per-tile partials of the EPI_NCE similarity GEMM
// (partial[2t] = sum exp(s - 1/T) over negatives, partial[2t+1] = sum s over positives); stats = {lse, -1/npos}
__global__ __launch_bounds__(256) void nce_finalize_kernel(const float* __restrict__ partial, int ntiles, float inv_t,
                                                           double npos, float* __restrict__ loss, float* __restrict__ stats) {
    __shared__ double red[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < ntiles; i += 256) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double lse = (double)inv_t + log(red[0][0]);
        *loss = (float)(lse - red[1][0] / npos);
        stats[0] = (float)lse;
        stats[1] = (float)(-1.0 / npos);
    }
}

// ============================================================================ fused SGD (momentum / Nesterov)
// torch.optim.SGD semantics (pretrain_videomae.py:187-189) over a flat range, one pass:
//   g = grad / grad_scale (+ wd * p);  buf = first ? g : m * buf + (1 - damp) * g;  d = nesterov ? g + m * buf : buf;
//   p -= lr * d;   the unscaled gradient is written back (what grad_logger reads after scaler.step).
// Skips everything when *found_inf != 0 (GradScaler's contract for optimisers with _step_supports_amp_scaling).
__global__ void sgd_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf, size_t n, float lr,
                                float momentum, float dampening, float wd, int nesterov, int first, int maximize,
                                const float* __restrict__ grad_scale, const float* __restrict__ found_inf, int write_grad,
                                bf16_t* __restrict__ shadow) {
    if (found_inf && *found_inf != 0.f) return;
    const float inv = grad_scale ? 1.f / *grad_scale : 1.f;
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
        f32x4 gv = *reinterpret_cast<f32x4*>(g + i) * inv;
        if (maximize) gv = -gv;
        const f32x4 gu = gv;
        if (wd != 0.f) gv += pv * wd;
        f32x4 d = gv;
        if (momentum != 0.f) {
            f32x4 bv = first ? gv : *reinterpret_cast<f32x4*>(buf + i) * momentum + gv * (1.f - dampening);
            *reinterpret_cast<f32x4*>(buf + i) = bv;
            d = nesterov ? gv + bv * momentum : bv;
        }
        const f32x4 pn = pv - d * lr;
        *reinterpret_cast<f32x4*>(p + i) = pn;
        // the bf16 copy the next forward's products read (round 4): written here, the per-forward cast pass over all parameters goes
        if (shadow) *reinterpret_cast<uint2*>(shadow + i) = uint2{pack2bf(pn[0], pn[1]), pack2bf(pn[2], pn[3])};
        if (write_grad) *reinterpret_cast<f32x4*>(g + i) = gu;
    } else {
        for (size_t j = i; j < n; ++j) {
            float gv = g[j] * inv;
            if (maximize) gv = -gv;
            const float gu = gv;
            if (wd != 0.f) gv += p[j] * wd;
            float d = gv;
            if (momentum != 0.f) {
                const float bv = first ? gv : buf[j] * momentum + gv * (1.f - dampening);
                buf[j] = bv;
                d = nesterov ? gv + bv * momentum : bv;
            }
            p[j] -= lr * d;
            if (shadow) shadow[j] = f2bf(p[j]);
            if (write_grad) g[j] = gu;
        }
    }
}

// torch.optim.Adam / AdamW (pretrain_videomae.py:190-193: AdamW(betas=(0.9, 0.95)) / Adam) over a flat range.
// adam_prep: state[0] = step count (f32, advanced unless *found_inf != 0), state[1] = lr / (1 - beta1^step),
// state[2] = sqrt(1 - beta2^step) - the scalars torch derives in double on the host, derived in double here
__global__ void adam_prep_kernel(float* __restrict__ state, double lr, double beta1, double beta2, const float* __restrict__ found_inf) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (found_inf && *found_inf != 0.f) return;
    const double step = (double)state[0] + 1.0;
    state[0] = (float)step;
    state[1] = (float)(lr / (1.0 - pow(beta1, step)));
    state[2] = (float)sqrt(1.0 - pow(beta2, step));
}

//   g = grad / grad_scale;  AdamW: p *= 1 - lr wd;  Adam: g += wd p;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)            (torch/optim/adam.py _single_tensor_adam, same operation order)
__global__ void adam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 size_t n, float omb1, float beta2, float omb2, float eps, float wd, float decay_mul, int decoupled,
                                 int maximize, const float* __restrict__ state, const float* __restrict__ grad_scale,
                                 const float* __restrict__ found_inf, int write_grad, bf16_t* __restrict__ shadow) {
    if (found_inf && *found_inf != 0.f) return;
    const float inv = grad_scale ? 1.f / *grad_scale : 1.f;
    const float step_size = state[1], bc2s = state[2];
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    const bool vec = i0 + 4 <= n;
    float pv[4], gv[4], mv[4], vv[4];
    const int cnt = vec ? 4 : (int)(n - i0);
    if (vec) {
        *reinterpret_cast<f32x4*>(pv) = *reinterpret_cast<const f32x4*>(p + i0);
        *reinterpret_cast<f32x4*>(gv) = *reinterpret_cast<const f32x4*>(g + i0);
        *reinterpret_cast<f32x4*>(mv) = *reinterpret_cast<const f32x4*>(m + i0);
        *reinterpret_cast<f32x4*>(vv) = *reinterpret_cast<const f32x4*>(v + i0);
    } else {
        for (int e = 0; e < cnt; ++e) { pv[e] = p[i0 + e]; gv[e] = g[i0 + e]; mv[e] = m[i0 + e]; vv[e] = v[i0 + e]; }
    }
    float gu[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float gr = gv[e] * inv;
        if (maximize) gr = -gr;
        gu[e] = gr;
        float pe = pv[e];
        if (wd != 0.f) {
            if (decoupled) pe *= decay_mul; else gr += wd * pe;
        }
        const float me = mv[e] + omb1 * (gr - mv[e]);
        const float ve = vv[e] * beta2 + omb2 * (gr * gr);
        const float denom = sqrtf(ve) / bc2s + eps;
        pv[e] = pe - step_size * (me / denom);
        mv[e] = me; vv[e] = ve;
    }
    if (vec) {
        *reinterpret_cast<f32x4*>(p + i0) = *reinterpret_cast<f32x4*>(pv);
        *reinterpret_cast<f32x4*>(m + i0) = *reinterpret_cast<f32x4*>(mv);
        *reinterpret_cast<f32x4*>(v + i0) = *reinterpret_cast<f32x4*>(vv);
        if (shadow) *reinterpret_cast<uint2*>(shadow + i0) = uint2{pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
        if (write_grad) *reinterpret_cast<f32x4*>(g + i0) = *reinterpret_cast<f32x4*>(gu);
    } else {
        for (int e = 0; e < cnt; ++e) {
            p[i0 + e] = pv[e]; m[i0 + e] = mv[e]; v[i0 + e] = vv[e];
            if (shadow) shadow[i0 + e] = f2bf(pv[e]);
            if (write_grad) g[i0 + e] = gu[e];
        }
    }
}

// ============================================================================ the same updates over a flat buffer with PARAMETER GROUPS
// The reference's JEPA optimiser has four groups (pretraining/predictive/helper.py:123-147: encoder / predictor weights with weight
// decay, their biases and 1-D tensors with weight_decay 0) and the flat layout interleaves weights and biases: a launch per run of
// memory-adjacent parameters of one group would be ~9 launches per layer.  Here ONE launch covers the whole flat buffer: a static
// table cuts it into segments (seg_start[s] .. seg_start[s + 1], ascending, element offsets) owned by group seg_group[s] (-1: no
// group - frozen parameters, padding - left untouched), blk_seg[b] is the segment that holds element 1024 b (so that a thread
// finds its segment in one or two steps), and the groups' hyper-parameters travel by value as kernel arguments (they change per
// step under a schedule; the table does not).  A quad of elements inside one segment takes the vector path - the same arithmetic,
// statement for statement, as sgd_step_kernel / adam_step_kernel -, a quad that straddles a boundary goes element by element.
struct SgdGroupsDev {
    float lr[BVC_OPT_MAX_GROUPS], wd[BVC_OPT_MAX_GROUPS], momentum[BVC_OPT_MAX_GROUPS], dampening[BVC_OPT_MAX_GROUPS];
    int flags[BVC_OPT_MAX_GROUPS];      // 1 = nesterov, 2 = maximize, 4 = first step (buf = g)
};
struct AdamGroupsDev {
    float omb1[BVC_OPT_MAX_GROUPS], beta2[BVC_OPT_MAX_GROUPS], omb2[BVC_OPT_MAX_GROUPS], eps[BVC_OPT_MAX_GROUPS], wd[BVC_OPT_MAX_GROUPS],
        decay_mul[BVC_OPT_MAX_GROUPS];
    int flags[BVC_OPT_MAX_GROUPS];      // 1 = decoupled (AdamW), 2 = maximize
};

// hyper-parameter of group `grp` out of a by-value table: a select chain (kernel arguments live in scalar registers and cannot be
// indexed by a per-lane value without a trip through scratch)
template <typename T>
__device__ __forceinline__ T pick_group(const T (&tab)[BVC_OPT_MAX_GROUPS], int grp) {
    T v = tab[0];
#pragma unroll
    for (int i = 1; i < BVC_OPT_MAX_GROUPS; ++i) v = grp == i ? tab[i] : v;
    return v;
}

__device__ __forceinline__ int find_segment(const int64_t* __restrict__ seg_start, int s, int64_t i) {
    while (seg_start[s + 1] <= i) ++s;
    return s;
}

__global__ __launch_bounds__(256) void sgd_step_seg_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf, int64_t n,
                                                           const int64_t* __restrict__ seg_start, const int* __restrict__ seg_group,
                                                           const int* __restrict__ blk_seg, const SgdGroupsDev G,
                                                           const float* __restrict__ grad_scale, const float* __restrict__ found_inf,
                                                           int write_grad, bf16_t* __restrict__ shadow) {
    if (found_inf && *found_inf != 0.f) return;
    const float inv = grad_scale ? 1.f / *grad_scale : 1.f;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    int s = find_segment(seg_start, blk_seg[blockIdx.x], i);
    if (i + 4 <= n && i + 4 <= seg_start[s + 1]) {
        const int grp = seg_group[s];
        if (grp < 0) return;
        const float lr = pick_group(G.lr, grp), wd = pick_group(G.wd, grp), momentum = pick_group(G.momentum, grp),
                    dampening = pick_group(G.dampening, grp);
        const int fl = pick_group(G.flags, grp);
        f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
        f32x4 gv = *reinterpret_cast<f32x4*>(g + i) * inv;
        if (fl & 2) gv = -gv;
        const f32x4 gu = gv;
        if (wd != 0.f) gv += pv * wd;
        f32x4 d = gv;
        if (momentum != 0.f) {
            f32x4 bv = (fl & 4) ? gv : *reinterpret_cast<f32x4*>(buf + i) * momentum + gv * (1.f - dampening);
            *reinterpret_cast<f32x4*>(buf + i) = bv;
            d = (fl & 1) ? gv + bv * momentum : bv;
        }
        const f32x4 pn = pv - d * lr;
        *reinterpret_cast<f32x4*>(p + i) = pn;
        if (shadow) *reinterpret_cast<uint2*>(shadow + i) = uint2{pack2bf(pn[0], pn[1]), pack2bf(pn[2], pn[3])};
        if (write_grad) *reinterpret_cast<f32x4*>(g + i) = gu;
    } else {
        for (int64_t j = i; j < n && j < i + 4; ++j) {
            s = find_segment(seg_start, s, j);
            const int grp = seg_group[s];
            if (grp < 0) continue;
            const float lr = pick_group(G.lr, grp), wd = pick_group(G.wd, grp), momentum = pick_group(G.momentum, grp),
                        dampening = pick_group(G.dampening, grp);
            const int fl = pick_group(G.flags, grp);
            float gv = g[j] * inv;
            if (fl & 2) gv = -gv;
            const float gu = gv;
            if (wd != 0.f) gv += p[j] * wd;
            float d = gv;
            if (momentum != 0.f) {
                const float bv = (fl & 4) ? gv : buf[j] * momentum + gv * (1.f - dampening);
                buf[j] = bv;
                d = (fl & 1) ? gv + bv * momentum : bv;
            }
            p[j] -= lr * d;
            if (shadow) shadow[j] = f2bf(p[j]);
            if (write_grad) g[j] = gu;
        }
    }
}

// state[3 grp + {0, 1, 2}] as adam_prep_kernel's state3, one thread per group
__global__ void adam_prep_groups_kernel(float* __restrict__ state, int ngroups, const double* __restrict__ hyper /* [ngroups][3]: lr, beta1, beta2 */,
                                        const float* __restrict__ found_inf) {
    const int grp = threadIdx.x;
    if (blockIdx.x != 0 || grp >= ngroups) return;
    if (found_inf && *found_inf != 0.f) return;
    const double lr = hyper[3 * grp], beta1 = hyper[3 * grp + 1], beta2 = hyper[3 * grp + 2];
    const double step = (double)state[3 * grp] + 1.0;
    state[3 * grp] = (float)step;
    state[3 * grp + 1] = (float)(lr / (1.0 - pow(beta1, step)));
    state[3 * grp + 2] = (float)sqrt(1.0 - pow(beta2, step));
}

__global__ __launch_bounds__(256) void adam_step_seg_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                            int64_t n, const int64_t* __restrict__ seg_start, const int* __restrict__ seg_group,
                                                            const int* __restrict__ blk_seg, const AdamGroupsDev G, const float* __restrict__ state,
                                                            const float* __restrict__ grad_scale, const float* __restrict__ found_inf,
                                                            int write_grad, bf16_t* __restrict__ shadow) {
    if (found_inf && *found_inf != 0.f) return;
    const float inv = grad_scale ? 1.f / *grad_scale : 1.f;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    int s = find_segment(seg_start, blk_seg[blockIdx.x], i0);
    const bool vec = i0 + 4 <= n && i0 + 4 <= seg_start[s + 1];
    const int cnt = i0 + 4 <= n ? 4 : (int)(n - i0);
    float pv[4], gv[4], mv[4], vv[4], gu[4];
    int grp4[4];
    if (vec) {
        *reinterpret_cast<f32x4*>(pv) = *reinterpret_cast<const f32x4*>(p + i0);
        *reinterpret_cast<f32x4*>(gv) = *reinterpret_cast<const f32x4*>(g + i0);
        *reinterpret_cast<f32x4*>(mv) = *reinterpret_cast<const f32x4*>(m + i0);
        *reinterpret_cast<f32x4*>(vv) = *reinterpret_cast<const f32x4*>(v + i0);
        const int grp = seg_group[s];
        if (grp < 0) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) grp4[e] = grp;
    } else {
        for (int e = 0; e < 4; ++e) {
            grp4[e] = -1;
            pv[e] = gv[e] = mv[e] = vv[e] = 0.f;
            if (e < cnt) {
                s = find_segment(seg_start, s, i0 + e);
                grp4[e] = seg_group[s];
                pv[e] = p[i0 + e]; gv[e] = g[i0 + e]; mv[e] = m[i0 + e]; vv[e] = v[i0 + e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int grp = grp4[e] < 0 ? 0 : grp4[e];
        const float omb1 = pick_group(G.omb1, grp), beta2 = pick_group(G.beta2, grp), omb2 = pick_group(G.omb2, grp),
                    eps = pick_group(G.eps, grp), wd = pick_group(G.wd, grp), decay_mul = pick_group(G.decay_mul, grp);
        const int fl = pick_group(G.flags, grp);
        const float step_size = state[3 * grp + 1], bc2s = state[3 * grp + 2];
        float gr = gv[e] * inv;
        if (fl & 2) gr = -gr;
        gu[e] = gr;
        float pe = pv[e];
        if (wd != 0.f) {
            if (fl & 1) pe *= decay_mul; else gr += wd * pe;
        }
        const float me = mv[e] + omb1 * (gr - mv[e]);
        const float ve = vv[e] * beta2 + omb2 * (gr * gr);
        const float denom = sqrtf(ve) / bc2s + eps;
        pv[e] = pe - step_size * (me / denom);
        mv[e] = me; vv[e] = ve;
    }
    if (vec) {
        *reinterpret_cast<f32x4*>(p + i0) = *reinterpret_cast<f32x4*>(pv);
        *reinterpret_cast<f32x4*>(m + i0) = *reinterpret_cast<f32x4*>(mv);
        *reinterpret_cast<f32x4*>(v + i0) = *reinterpret_cast<f32x4*>(vv);
        if (shadow) *reinterpret_cast<uint2*>(shadow + i0) = uint2{pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
        if (write_grad) *reinterpret_cast<f32x4*>(g + i0) = *reinterpret_cast<f32x4*>(gu);
    } else {
        for (int e = 0; e < cnt; ++e) {
            if (grp4[e] < 0) continue;
            p[i0 + e] = pv[e]; m[i0 + e] = mv[e]; v[i0 + e] = vv[e];
            if (shadow) shadow[i0 + e] = f2bf(pv[e]);
            if (write_grad) g[i0 + e] = gu[e];
        }
    }
}

// ============================================================================ zero-padded attention heads (hd -> hdp)
// The JEPA predictor inherits the encoder's head COUNT (vision_transformer.py:447,463: num_heads=encoder.num_heads), so ViT-L
// gives 16 heads of 24 dims - not an MFMA-friendly width.  The heads are run at 32 dims with zero padding: padded copies of
// the bf16 qkv / proj weights are made per use and the padded weight gradients are compacted back (all tiny: Dp = 384).
//   qkv rows:  r' = (which * H + h) * hdp + j  <-  r = (which * H + h) * hd + j   (j < hd; other rows zero)
//   proj cols: c' = h * hdp + j                <-  c = h * hd + j
__global__ void pad_qkv_kernel(const bf16_t* __restrict__ w, const float* __restrict__ b, bf16_t* __restrict__ wp, float* __restrict__ bp,
                               int H, int hd, int hdp, int D) {
    const int rp = blockIdx.x;                       // padded row in [0, 3 H hdp)
    const int j = rp % hdp, wh = rp / hdp;
    const bool live = j < hd;
    const int r = wh * hd + j;
    for (int c = threadIdx.x; c < D; c += blockDim.x) wp[(size_t)rp * D + c] = live ? w[(size_t)r * D + c] : (bf16_t)0;
    if (threadIdx.x == 0) bp[rp] = live ? b[r] : 0.f;
}
__global__ void pad_cols_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wp, int rows, int H, int hd, int hdp) {
    const int r = blockIdx.x;
    for (int cp = threadIdx.x; cp < H * hdp; cp += blockDim.x) {
        const int j = cp % hdp, h = cp / hdp;
        wp[(size_t)r * H * hdp + cp] = j < hd ? w[(size_t)r * H * hd + h * hd + j] : (bf16_t)0;
    }
}
// gradients back to the reference's layout (plain stores: nothing else writes these ranges)
__global__ void unpad_qkv_grad_kernel(const float* __restrict__ gwp, const float* __restrict__ gbp, float* __restrict__ gw,
                                      float* __restrict__ gb, int H, int hd, int hdp, int D) {
    const int r = blockIdx.x;                        // true row in [0, 3 H hd)
    const int j = r % hd, wh = r / hd;
    const int rp = wh * hdp + j;
    for (int c = threadIdx.x; c < D; c += blockDim.x) gw[(size_t)r * D + c] = gwp[(size_t)rp * D + c];
    if (threadIdx.x == 0) gb[r] = gbp[rp];
}
__global__ void unpad_cols_grad_kernel(const float* __restrict__ gwp, float* __restrict__ gw, int rows, int H, int hd, int hdp) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < H * hd; c += blockDim.x) {
        const int j = c % hd, h = c / hd;
        gw[(size_t)r * H * hd + c] = gwp[(size_t)r * H * hdp + h * hdp + j];
    }
}

// GradScaler's inf check as ONE read-only pass: *found_inf = 1 if any element of x is Inf or NaN (never cleared here).
// torch does it with _amp_foreach_non_finite_check_and_unscale_(grads, found_inf, inv_scale = 1): a read AND a write of every
// gradient, 6 multi-tensor launches for VideoMAE-base (0.24 ms per step at 3.1 TB/s); this reads 377 MB once.
__global__ void nonfinite_check_kernel(const float* __restrict__ x, size_t n, float* __restrict__ found_inf) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    bool bad = false;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (; i + 4 <= n; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        // (v - v) is 0 for finite values and NaN for Inf / NaN
        const float t = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);
        bad |= (t != 0.f);
    }
    for (; i < n; ++i) bad |= ((x[i] - x[i]) != 0.f);      // tail of the last vector
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.0f;
}

// ============================================================================ launchers
static inline unsigned blocks_for(size_t items, int per = 256) { return (unsigned)((items + per - 1) / per); }

// two rows per wave (32 lanes each) when a row is at most 512 wide and a whole number of float4 per lane: 384 = 3 per lane
static inline bool ln_half_wave_rows(int D) { return D <= 512 && D % 128 == 0; }
// grid of a grid-stride row kernel: at most `cap` workgroups, and every workgroup the same number of row groups (2560 groups on
// 2048 workgroups would give a quarter of them twice the work of the rest)
static inline int balanced_grid(int ngroups, int cap) {
    if (ngroups <= cap) return ngroups > 0 ? ngroups : 1;
    // a few groups per workgroup: keep every CU's resident slots full rather than even (encoder LayerNorm backward at 256 clips,
    // 2560 groups: 2048 workgroups 99-105 us, 1280 even ones 118 us - five per CU do not cover the HBM latency)
    if (ngroups < 4 * cap) return cap;
    const int rounds = (ngroups + cap - 1) / cap;
    return (ngroups + rounds - 1) / rounds;
}
constexpr int kLnFwdMaxBlocks = 4096;      // 16 per CU (the kernel holds its row in registers: 8 resident workgroups per CU and a second round)

int launch_ln_fwd(const float* x, RowMap rm, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                  int M, int D, float eps, hipStream_t s, float* y32) {
    BVC_REQUIRE(D % 4 == 0 && D <= kMaxChunks * 256, "ln_fwd: D=%d unsupported", D);
    if (ln_half_wave_rows(D))
        hipLaunchKernelGGL(ln_fwd_kernel<32>, dim3(balanced_grid((M + 7) / 8, kLnFwdMaxBlocks)), dim3(256), 0, s, x, rm, gamma, beta, y, y32, mean, rstd, M, D, eps);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<64>, dim3(balanced_grid((M + 3) / 4, kLnFwdMaxBlocks)), dim3(256), 0, s, x, rm, gamma, beta, y, y32, mean, rstd, M, D, eps);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

// 64 rows per workgroup for M >= 65536 was measured slower (B=64 decoder: 1276 vs 1157 us per step for the LN backward, which
// outweighs the 70 us saved in the partial reduction), so the third tier stays unused.
static inline int ln_bwd_rows_per_block(int M) { return M >= 16384 ? 16 : 4; }
constexpr int kLnBwdMaxBlocks = 2048;      // 8 per CU: every CU keeps its ~7 resident workgroups (register-limited) busy to the end

size_t ln_bwd_workspace_floats(int M, int D) {
    const int rpb = ln_bwd_rows_per_block(M);
    return (size_t)((M + rpb - 1) / rpb) * 2 * D;
}

// enough for every row count m <= Mmax (the rows-per-workgroup choice is not monotonic in m)
size_t ln_bwd_workspace_floats_upto(int Mmax, int D) {
    size_t best = ln_bwd_workspace_floats(Mmax, D);
    for (int edge : {16383, 65535}) {
        const size_t w = ln_bwd_workspace_floats(Mmax < edge ? Mmax : edge, D);
        best = w > best ? w : best;
    }
    return best;
}

int launch_ln_bwd(const bf16_t* dy, const float* x, RowMap rm, const float* mean, const float* rstd, const float* gamma,
                  float* dres, int accumulate, bf16_t* dres_bf, float* dgamma, float* dbeta, float* part, int M, int D, hipStream_t s) {
    BVC_REQUIRE(D % 4 == 0 && D <= kMaxChunks * 256, "ln_bwd: D=%d unsupported", D);
    BVC_REQUIRE(part != nullptr, "ln_bwd: workspace missing");
    // rows per workgroup: enough workgroups to keep >= 16 waves per CU streaming (the kernel is HBM-bound and
    // each wave walks its rows serially), few enough that the per-column atomics stay negligible
    const int rpb = ln_bwd_rows_per_block(M);
    const int rpg = rpb == 4 && ln_half_wave_rows(D) ? 8 : rpb;          // rows per row group of the instantiation launched below
    const int nblk = balanced_grid((M + rpg - 1) / rpg, kLnBwdMaxBlocks);
    const bool half = ln_half_wave_rows(D);
    const size_t lds = (size_t)(half ? 16 : 8) * D * sizeof(float);
#define BVC_LN_BWD(RPB_, LPR_) hipLaunchKernelGGL((ln_bwd_kernel<RPB_, LPR_>), dim3(nblk), dim3(256), lds, s, dy, x, rm, mean, rstd, gamma, dres, accumulate, dres_bf, part, M, D)
    if (rpb == 16) { if (half) BVC_LN_BWD(16, 32); else BVC_LN_BWD(16, 64); }
    else { if (half) BVC_LN_BWD(8, 32); else BVC_LN_BWD(4, 64); }
#undef BVC_LN_BWD
    BVC_CHECK_HIP(hipGetLastError());
    return launch_ln_param_reduce(part, nblk, D, dgamma, dbeta, s);
}

int launch_ln_param_reduce(const float* part, int nblk, int D, float* dgamma, float* dbeta, hipStream_t s) {
    BVC_REQUIRE(part && dgamma && dbeta && nblk > 0 && D > 0, "ln_param_reduce: bad argument");
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * D + 255) / 256, (nblk + 15) / 16), dim3(256), 0, s, part, nblk, D, dgamma, dbeta);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_colsum_bf16_scaled(const bf16_t* X, int M, int N, int ld, float alpha, const float* alpha_dev, float* out, hipStream_t s) {
    BVC_REQUIRE(N % 8 == 0 && ld % 8 == 0, "colsum_bf16: N and ld must be multiples of 8");
    const int RB = M >= 16384 ? 256 : 64;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N + 511) / 512, (M + RB - 1) / RB), dim3(256), 0, s, X, M, N, ld, alpha, alpha_dev, out, RB);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_colsum_bf16(const bf16_t* X, int M, int N, int ld, float alpha, float* out, hipStream_t s) {
    return launch_colsum_bf16_scaled(X, M, N, ld, alpha, nullptr, out, s);
}

int launch_colsum_f32(const float* X, RowMap rm, int M, int D, float* out, hipStream_t s) {
    BVC_REQUIRE(D % 4 == 0, "colsum_f32: D must be a multiple of 4");
    const int RB = 256;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((D + 255) / 256, (M + RB - 1) / RB), dim3(256), 0, s, X, rm, M, D, out, RB);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_token_mean(const float* x, int B, int N, int D, float* out, hipStream_t s) {
    BVC_REQUIRE(D % 4 == 0, "token_mean: D must be a multiple of 4");
    hipLaunchKernelGGL(token_mean_kernel, dim3((D + 63) / 64, B), dim3(256), 0, s, x, N, D, out);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_token_mean_bwd(const float* dmean, int B, int N, int D, float* dx, hipStream_t s) {
    BVC_REQUIRE(D % 4 == 0, "token_mean_bwd: D must be a multiple of 4");
    const size_t total4 = (size_t)B * N * D / 4;
    hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(blocks_for(total4)), dim3(256), 0, s, dmean, dx, N, D, total4);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_cast_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks_for((n + 7) / 8)), dim3(256), 0, s, in, out, n);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_gather_rows_bf16(const float* in, RowMap rm, bf16_t* out, int M, int D, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(blocks_for((size_t)M * (D / 4))), dim3(256), 0, s, in, rm, out, M, D);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_mask_index(const uint8_t* mask, int B, int L, int nvis, int nmask, int* vis_idx, int* msk_idx, int* status, hipStream_t s) {
    hipLaunchKernelGGL(mask_index_kernel, dim3(B), dim3(64), 0, s, mask, L, nvis, nmask, vis_idx, msk_idx, status);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_gather_patches(PixelSrc clip, const int* vis_idx, bf16_t* A, int B, int nvis, PatchGeom pg, hipStream_t s) {
    BVC_REQUIRE(pg.ps % 8 == 0 && pg.W % 4 == 0, "gather_patches: patch size must be a multiple of 8");
    const size_t items = (size_t)B * nvis * (pg.C * pg.ts * pg.ps * pg.ps / 8);
    hipLaunchKernelGGL(gather_patches_kernel, dim3(blocks_for(items)), dim3(256), 0, s, clip, vis_idx, A, B, nvis, pg);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_labels(PixelSrc clip, const int* msk_idx, float* labels, int B, int nmask, PatchGeom pg, int norm_pix, hipStream_t s) {
    BVC_REQUIRE(pg.ps % 4 == 0, "labels: patch size must be a multiple of 4");
    const int E = pg.ts * pg.ps * pg.ps;
    const size_t lds = (size_t)(pg.C * E + 2 * pg.C) * 4;
    // one wave per token when a token is at most 8 x 64 chunks of 16 bytes (C E <= 2048 values: every configuration of the reference)
    const int nchunk = pg.C * (E >> 2), ntok = B * nmask;
    if (pg.C == 3 && (E == 512 || E == 256)) {      // the reference's tubes: constant channel count, statistics from registers
        const int nblk = (ntok + 3) / 4;
        if (E == 512) hipLaunchKernelGGL(labels_wave3_kernel<2>, dim3(nblk), dim3(256), (size_t)4 * 3 * E * 4, s, clip, msk_idx, labels, ntok, nmask, pg, norm_pix);
        else hipLaunchKernelGGL(labels_wave3_kernel<1>, dim3(nblk), dim3(256), (size_t)4 * 3 * E * 4, s, clip, msk_idx, labels, ntok, nmask, pg, norm_pix);
        BVC_CHECK_HIP(hipGetLastError());
        return BVC_OK;
    }
    if (nchunk <= 512 && lds <= 16384) {
        const int per_wave = (pg.C * E + 2 * pg.C + 3) & ~3;
        const int nblk = (ntok + 3) / 4;
        if (nchunk <= 384)
            hipLaunchKernelGGL(labels_wave_kernel<6>, dim3(nblk), dim3(256), (size_t)per_wave * 16, s, clip, msk_idx, labels, ntok, nmask, pg, norm_pix);
        else
            hipLaunchKernelGGL(labels_wave_kernel<8>, dim3(nblk), dim3(256), (size_t)per_wave * 16, s, clip, msk_idx, labels, ntok, nmask, pg, norm_pix);
        BVC_CHECK_HIP(hipGetLastError());
        return BVC_OK;
    }
    hipLaunchKernelGGL(labels_kernel, dim3(B * nmask), dim3(256), lds, s, clip, msk_idx, labels, nmask, pg, norm_pix);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_fill_masked(float* xfull, const float* mask_token, const float* pos, const int* msk_idx, int B, int L, int nvis,
                       int nmask, int D, hipStream_t s) {
    hipLaunchKernelGGL(fill_masked_kernel, dim3(blocks_for((size_t)B * nmask * (D / 4))), dim3(256), 0, s, xfull, mask_token, pos, msk_idx, B, L, nvis, nmask, D);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_target_select(const float* h, const int* idx, float* out, int nsets, int B, int Np, int L, int D, float eps, hipStream_t s) {
    BVC_REQUIRE(D % 4 == 0 && D <= kMaxChunks * 256, "target_select: D=%d unsupported", D);
    const int rows = nsets * B * Np;
    hipLaunchKernelGGL(target_select_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, h, idx, out, rows, B, Np, L, D, eps);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_pred_assemble(const float* xe, const float* mask_token, const float* pos, const int* idx_pred, float* X, int nsets, int B,
                         int Nc, int Np, int D, hipStream_t s) {
    const size_t items = (size_t)nsets * B * (Nc + Np) * (D / 4);
    hipLaunchKernelGGL(pred_assemble_kernel, dim3(blocks_for(items)), dim3(256), 0, s, xe, mask_token, pos, idx_pred, X, nsets * B, B, Nc, Np, D);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_pred_ctx_grad(const float* dX, bf16_t* dxe, int nsets, int B, int Nc, int Np, int D, hipStream_t s) {
    hipLaunchKernelGGL(pred_ctx_grad_kernel, dim3(blocks_for((size_t)B * Nc * (D / 4))), dim3(256), 0, s, dX, dxe, nsets, B, Nc, Np, D);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int smooth_l1_blocks(size_t n) { return (int)std::min<size_t>(1024, (n + 1023) / 1024); }

int launch_smooth_l1_fwd(const float* z, const float* h, size_t n, float* partial, float* loss, hipStream_t s) {
    BVC_REQUIRE(n > 0, "smooth_l1: empty input");
    const int nb = smooth_l1_blocks(n);
    hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3(nb), dim3(256), 0, s, z, h, n, partial);
    BVC_CHECK_HIP(hipGetLastError());
    return launch_loss_finalize(partial, nb, (double)n, nullptr, loss, s);
}

int launch_smooth_l1_bwd(const float* z, const float* h, const float* gout, size_t n, float* dz, hipStream_t s) {
    hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, s, z, h, gout, n, dz);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_ema(float* k, const float* q, size_t n, float m, hipStream_t s) {
    if (n == 0) return BVC_OK;
    hipLaunchKernelGGL(ema_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, s, k, q, n, m);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_iota_mod(int* idx, int n, int L, hipStream_t s) {
    hipLaunchKernelGGL(iota_mod_kernel, dim3(blocks_for((size_t)n)), dim3(256), 0, s, idx, n, L);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_row_normalize(const float* f, bf16_t* fn, float* inv, int n, int p, float eps, hipStream_t s) {
    BVC_REQUIRE(p % 4 == 0, "row_normalize: width must be a multiple of 4");
    hipLaunchKernelGGL(row_normalize_kernel, dim3((n + 3) / 4), dim3(256), 0, s, f, fn, inv, n, p, eps);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_row_normalize_bwd(const float* f, const float* inv, const float* dfn, float* df, int n, int p, hipStream_t s) {
    BVC_REQUIRE(p % 4 == 0, "row_normalize_bwd: width must be a multiple of 4");
    hipLaunchKernelGGL(row_normalize_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, s, f, inv, dfn, df, n, p);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_nce_finalize(const float* partial, int ntiles, float inv_t, double npos, float* loss, float* stats, hipStream_t s) {
    hipLaunchKernelGGL(nce_finalize_kernel, dim3(1), dim3(256), 0, s, partial, ntiles, inv_t, npos, loss, stats);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_sgd_step(float* p, float* g, float* buf, size_t n, float lr, float momentum, float dampening, float wd, int nesterov,
                    int first, int maximize, const float* grad_scale, const float* found_inf, int write_grad, bf16_t* shadow,
                    hipStream_t s) {
    BVC_REQUIRE(momentum == 0.f || buf != nullptr, "sgd_step: momentum needs a buffer");
    BVC_REQUIRE(shadow == nullptr || (uintptr_t)shadow % 8 == 0, "sgd_step: the bf16 shadow must be 8-byte aligned");
    BVC_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && (buf == nullptr || (uintptr_t)buf % 16 == 0),
                "sgd_step: buffers must be 16-byte aligned");
    if (n == 0) return BVC_OK;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, s, p, g, buf, n, lr, momentum, dampening, wd,
                       nesterov, first, maximize, grad_scale, found_inf, write_grad, shadow);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_adam_prep(float* state, double lr, double beta1, double beta2, const float* found_inf, hipStream_t s) {
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, state, lr, beta1, beta2, found_inf);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_adam_step(float* p, float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps, double wd,
                     int decoupled, int maximize, const float* state, const float* grad_scale, const float* found_inf, int write_grad,
                     bf16_t* shadow, hipStream_t s) {
    BVC_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                "adam_step: buffers must be 16-byte aligned");
    BVC_REQUIRE(shadow == nullptr || (uintptr_t)shadow % 8 == 0, "adam_step: the bf16 shadow must be 8-byte aligned");
    if (n == 0) return BVC_OK;
    // hyper-parameters arrive as doubles (python floats) and are combined in double before the cast, as torch does
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, s, p, g, m, v, n, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)wd, (float)(1.0 - lr * wd), decoupled, maximize, state, grad_scale,
                       found_inf, write_grad, shadow);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

static int check_segments(const char* who, int64_t n, const int64_t* seg_start, const int* seg_group, const int* blk_seg, int nseg, int ngroups) {
    BVC_REQUIRE(seg_start && seg_group && blk_seg && nseg > 0, "%s: segment table missing", who);
    BVC_REQUIRE(ngroups >= 1 && ngroups <= BVC_OPT_MAX_GROUPS, "%s: %d parameter groups (1..%d)", who, ngroups, BVC_OPT_MAX_GROUPS);
    BVC_REQUIRE(n > 0, "%s: empty range", who);
    return BVC_OK;
}

int launch_sgd_step_segments(float* p, float* g, float* buf, int64_t n, const int64_t* seg_start, const int* seg_group, const int* blk_seg,
                             int nseg, const bvc_sgd_groups* gr, const float* grad_scale, const float* found_inf, int write_grad,
                             bf16_t* shadow, hipStream_t s) {
    BVC_REQUIRE(gr != nullptr, "sgd_step_segments: groups missing");
    if (int rc = check_segments("sgd_step_segments", n, seg_start, seg_group, blk_seg, nseg, gr->ngroups)) return rc;
    BVC_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && (buf == nullptr || (uintptr_t)buf % 16 == 0),
                "sgd_step_segments: buffers must be 16-byte aligned");
    BVC_REQUIRE(shadow == nullptr || (uintptr_t)shadow % 8 == 0, "sgd_step_segments: the bf16 shadow must be 8-byte aligned");
    SgdGroupsDev G;
    for (int i = 0; i < BVC_OPT_MAX_GROUPS; ++i) {
        const int j = i < gr->ngroups ? i : 0;
        BVC_REQUIRE(gr->momentum[j] == 0.f || buf != nullptr, "sgd_step_segments: momentum needs a buffer");
        G.lr[i] = gr->lr[j]; G.wd[i] = gr->weight_decay[j]; G.momentum[i] = gr->momentum[j]; G.dampening[i] = gr->dampening[j];
        G.flags[i] = (gr->nesterov[j] ? 1 : 0) | (gr->maximize[j] ? 2 : 0) | (gr->first_step[j] ? 4 : 0);
    }
    hipLaunchKernelGGL(sgd_step_seg_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, p, g, buf, n, seg_start, seg_group, blk_seg, G,
                       grad_scale, found_inf, write_grad, shadow);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_adam_step_segments(float* p, float* g, float* m, float* v, int64_t n, const int64_t* seg_start, const int* seg_group,
                              const int* blk_seg, int nseg, const bvc_adam_groups* gr, float* state, double* hyper_dev,
                              const float* grad_scale, const float* found_inf, int write_grad, bf16_t* shadow, hipStream_t s) {
    BVC_REQUIRE(gr != nullptr && state != nullptr && hyper_dev != nullptr, "adam_step_segments: groups / state missing");
    if (int rc = check_segments("adam_step_segments", n, seg_start, seg_group, blk_seg, nseg, gr->ngroups)) return rc;
    BVC_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                "adam_step_segments: buffers must be 16-byte aligned");
    BVC_REQUIRE(shadow == nullptr || (uintptr_t)shadow % 8 == 0, "adam_step_segments: the bf16 shadow must be 8-byte aligned");
    AdamGroupsDev G;
    double hyper[3 * BVC_OPT_MAX_GROUPS];
    for (int i = 0; i < BVC_OPT_MAX_GROUPS; ++i) {
        const int j = i < gr->ngroups ? i : 0;
        // hyper-parameters arrive as doubles (python floats) and are combined in double before the cast, as torch does
        G.omb1[i] = (float)(1.0 - gr->beta1[j]); G.beta2[i] = (float)gr->beta2[j]; G.omb2[i] = (float)(1.0 - gr->beta2[j]);
        G.eps[i] = (float)gr->eps[j]; G.wd[i] = (float)gr->weight_decay[j]; G.decay_mul[i] = (float)(1.0 - gr->lr[j] * gr->weight_decay[j]);
        G.flags[i] = (gr->decoupled[j] ? 1 : 0) | (gr->maximize[j] ? 2 : 0);
        hyper[3 * i] = gr->lr[j]; hyper[3 * i + 1] = gr->beta1[j]; hyper[3 * i + 2] = gr->beta2[j];
    }
    // the step-dependent scalars of every group in one tiny launch (lr, beta1, beta2 as doubles: 192 bytes through a pageable copy)
    BVC_CHECK_HIP(hipMemcpyAsync(hyper_dev, hyper, sizeof(double) * 3 * gr->ngroups, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(adam_prep_groups_kernel, dim3(1), dim3(64), 0, s, state, gr->ngroups, hyper_dev, found_inf);
    hipLaunchKernelGGL(adam_step_seg_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, p, g, m, v, n, seg_start, seg_group, blk_seg, G, state,
                       grad_scale, found_inf, write_grad, shadow);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_pad_heads(const bf16_t* wqkv, const float* bqkv, const bf16_t* wo, bf16_t* wqkv_p, float* bqkv_p, bf16_t* wo_p, int D, int H,
                     int hd, int hdp, hipStream_t s) {
    hipLaunchKernelGGL(pad_qkv_kernel, dim3(3 * H * hdp), dim3(128), 0, s, wqkv, bqkv, wqkv_p, bqkv_p, H, hd, hdp, D);
    hipLaunchKernelGGL(pad_cols_kernel, dim3(D), dim3(128), 0, s, wo, wo_p, D, H, hd, hdp);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_unpad_head_grads(const float* gwqkv_p, const float* gbqkv_p, const float* gwo_p, float* gwqkv, float* gbqkv, float* gwo, int D,
                            int H, int hd, int hdp, hipStream_t s) {
    hipLaunchKernelGGL(unpad_qkv_grad_kernel, dim3(3 * H * hd), dim3(128), 0, s, gwqkv_p, gbqkv_p, gwqkv, gbqkv, H, hd, hdp, D);
    hipLaunchKernelGGL(unpad_cols_grad_kernel, dim3(D), dim3(128), 0, s, gwo_p, gwo, D, H, hd, hdp);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_nonfinite_check(const float* x, size_t n, float* found_inf, hipStream_t s) {
    BVC_REQUIRE(((uintptr_t)x % 16) == 0, "nonfinite_check: buffer must be 16-byte aligned");
    if (n == 0) return BVC_OK;
    const size_t want = (n / 4 + 255) / 256 + 1;
    const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(nonfinite_check_kernel, dim3(blocks), dim3(256), 0, s, x, n, found_inf);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_loss_finalize(const float* partial, int n, double count, const int* status, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, n, count, status, loss);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

}  // namespace bvc
