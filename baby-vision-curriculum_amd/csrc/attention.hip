// Multi-head self-attention forward / backward for head_dim 64 on gfx950 (no mask, no dropout),
// flash-style: the N x N score matrix never leaves registers.
//
// Reference semantics: softmax(Q K^T / sqrt(d)) V per (batch, head)   (HF:181-206 / SDPA, HF:239-252).
// Layout in HBM: qkv bf16 [B*N][3*D] with q | k | v column blocks (each head a 64-wide slice),
// ctx bf16 [B*N][D], lse f32 [B*H][N] in log2 units (lse2 = log2 sum_k exp(s_k * scale)).
//
// All three kernels use v_mfma_f32_32x32x16_bf16 and keep the softmax operand in registers:
// a 32x32 f32 accumulator X has its column on the lane and its rows in the 16 registers, so a
// following MFMA that sums over X's ROW index takes bf16(X) as its B operand with no lane movement;
// the other operand's k order follows the same permutation (k-step s, lane half h, element j <->
// row 16 s + 8 (j>>2) + 4 h + (j&3)) and is produced by ds_read_b64_tr_b16 from a row-major LDS tile.
//   forward : S^T = K Q^T        (query on the lane)  ->  O^T += V^T P^T
//   dQ      : S^T, dP^T = V dO^T (query on the lane)  ->  dQ^T += K^T dS^T
//   dK/dV   : S = Q K^T, dP = dO V^T (key on the lane) -> dV^T += dO^T P,  dK^T += Q^T dS
// so every per-row softmax statistic is lane-local, there are no atomics, and no score tile
// round-trips through LDS.  dS/dP are recomputed in the dQ kernel (7 instead of 5 products) in
// exchange for a deterministic, atomic-free dQ.
//
// K/V (or Q/dO) tiles of 64 rows x 64 bf16 go HBM -> LDS by LDS-DMA into a 2-stage ring, XOR-swizzled
// so that both the ds_read_b128 row reads and the transposed reads of the same image are
// bank-conflict free.
#include "attention.h"

namespace bvc {

// swizzle of the 16-B chunk index of row r in a [rows][64] bf16 tile (128-B rows); serves both
// ds_read_b128 row fragments (16 rows, one chunk) and tr reads (4 rows x 4 chunks)
__device__ __forceinline__ int swz_dual(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// stage rows [row0, row0+64) x 64 bf16 starting at element column `col0` of a [rows][ld] bf16 array
// into an 8 KiB LDS image; 8 pieces of 1 KiB, two per wave
__device__ __forceinline__ void stage64(__amdgpu_buffer_rsrc_t rs, int row0, int ld, int col0, char* lds,
                                        int wave, int lane) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = wave + 4 * jj;
        const int r = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ swz_dual(r);
        const uint32_t off = (uint32_t)(((size_t)(row0 + r) * ld + col0 + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + j * 1024), 16, off, 0, 0, 0);
    }
}

// row fragment for the 32x32x16 A operand: lane (r = l&31, h = l>>5) gets tile[rbase + r][16 step + 8 h + 0..7]
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int rbase, int step, int lane) {
    const int r = rbase + (lane & 31);
    const int c = 2 * step + (lane >> 5);
    return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(
        (const __attribute__((address_space(3))) char*)(lds) + r * 128 + ((c ^ swz_dual(r)) << 4));
}

// transposed fragment for the A operand of a product that sums over the tile's ROW index with the
// permuted k order of an in-register accumulator operand: lane (i = l&31, h) gets
// tile[rbase + 16 s + 8 (j>>2) + 4 h + (j&3)][cbase + i] for j = 0..7
__device__ __forceinline__ bf16x8 frag_tr(const char* lds, int rbase, int s, int cbase, int lane) {
    const int h = lane >> 5, q = (lane >> 2) & 3, p = lane & 3;
    const int col = cbase + 16 * ((lane >> 4) & 1) + 4 * p;
    const int chunk = col >> 3, within = (col & 7) * 2;
    const int r0 = rbase + 16 * s + 4 * h + q, r1 = r0 + 8;
    const __attribute__((address_space(3))) char* base = (const __attribute__((address_space(3))) char*)(lds);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) bf16x4*)(base + r0 * 128 + ((chunk ^ swz_dual(r0)) << 4) + within));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) bf16x4*)(base + r1 * 128 + ((chunk ^ swz_dual(r1)) << 4) + within));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// accumulator registers 8s..8s+7 -> bf16 fragment (B operand of the next product)
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int s) {
    union { bf16x8 v; uint32_t u[4]; } r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.u[j] = pack2bf(a[8 * s + 2 * j], a[8 * s + 2 * j + 1]);
    return r.v;
}

// row index (within a 32-row block) held by accumulator register `reg` of lane half h
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// 8 consecutive bf16 of one row straight from HBM (row clamped by the caller)
__device__ __forceinline__ bf16x8 load8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ============================================================================ forward
// grid (ceil(N/128), B*H); 256 threads; wave w owns queries q0 + 32 w .. + 31
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                       float* __restrict__ lse, int N, int H, int D,
                                                       uint32_t qkv_bytes, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (K 8 KiB + V 8 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const int qi = blockIdx.x * 128 + wave * 32 + (lane & 31);   // this lane's query
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, qkv_bytes);

    // Q^T fragments (B operand of S^T = K Q^T): Q[qi][16 step + 8 h + 0..7]
    bf16x8 qf[4];
    {
        const bf16_t* qrow = qkv + (size_t)(b * N + min(qi, N - 1)) * ld + head * 64 + 8 * h;
#pragma unroll
        for (int st = 0; st < 4; ++st) qf[st] = load8(qrow + 16 * st);
    }
    f32x16 o0 = zero16(), o1 = zero16();   // O^T rows d = 0..31 / 32..63, column = query
    float m_run = -INFINITY, l_run = 0.f;  // running max (log2 units) and this half's share of the sum

    const int nkt = (N + 63) >> 6;
    const int krow0 = b * N;
    stage64(rs, krow0, ld, D + head * 64, smem, wave, lane);
    stage64(rs, krow0, ld, 2 * D + head * 64, smem + 8192, wave, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) {
            char* nx = smem + ((kt + 1) & 1) * 16384;
            stage64(rs, krow0 + (kt + 1) * 64, ld, D + head * 64, nx, wave, lane);
            stage64(rs, krow0 + (kt + 1) * 64, ld, 2 * D + head * 64, nx + 8192, wave, lane);
        }
        const char* kl = smem + (kt & 1) * 16384;
        const char* vl = kl + 8192;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = kt * 64 + sub * 32;
            if (key0 >= N) break;   // workgroup-uniform
            f32x16 s = zero16();
#pragma unroll
            for (int st = 0; st < 4; ++st) s = MFMA32(frag_rows(kl, sub * 32, st, lane), qf[st], s);
            // ---- online softmax, one query per lane, 16 of the 32 keys in this lane half
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (key0 + 32 > N && key0 + acc_row(r, h) >= N) s[r] = -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * scale_log2);
            const float alpha = exp2f(m_run - m_new);
            m_run = m_new;
            float rs_ = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = exp2f(s[r] * scale_log2 - m_new);
                rs_ += s[r];
            }
            l_run = l_run * alpha + rs_;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            // ---- O^T += V^T P^T
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pf = acc_to_frag(s, ks);
                o0 = MFMA32(frag_tr(vl, sub * 32, ks, 0, lane), pf, o0);
                o1 = MFMA32(frag_tr(vl, sub * 32, ks, 32, lane), pf, o1);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qi < N) {
        bf16_t* orow = ctx + (size_t)(b * N + qi) * D + head * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * h;
            uint2 a = {pack2bf(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack2bf(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
            uint2 c = {pack2bf(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack2bf(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
            *reinterpret_cast<uint2*>(orow + d) = a;
            *reinterpret_cast<uint2*>(orow + 32 + d) = c;
        }
        if (h == 0) lse[(size_t)bh * N + qi] = m_run + log2f(l_tot);
    }
}

// ============================================================================ delta = rowsum(dO * O)
// one thread per (row, head, 8-column chunk); 8 lanes per (row, head)
__global__ void attn_delta_kernel(const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                  float* __restrict__ delta, int B, int N, int H, int D) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * N * H * 8;
    float acc = 0.f;
    long long item = gid >> 3;
    const int piece = (int)(gid & 7);
    if (gid < total) {
        const long long row = item / H;
        const int head = (int)(item % H);
        const size_t off = (size_t)row * D + head * 64 + piece * 8;
        const bf16x8 a = load8(dctx + off), o = load8(ctx + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += bf2f((bf16_t)a[j]) * bf2f((bf16_t)o[j]);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (gid < total && piece == 0) {
        const long long row = item / H;
        const int head = (int)(item % H);
        const int b = (int)(row / N), q = (int)(row % N);
        delta[((size_t)b * H + head) * N + q] = acc;
    }
}

// ============================================================================ dQ
// grid (ceil(N/128), B*H); wave w owns 32 queries; loops over key tiles (K and V staged)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          bf16_t* __restrict__ dqkv, int N, int H, int D,
                                                          uint32_t qkv_bytes, float scale, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const int qi = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int qc = min(qi, N - 1);
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, qkv_bytes);

    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qrow = qkv + (size_t)(b * N + qc) * ld + head * 64 + 8 * h;
        const bf16_t* drow = dctx + (size_t)(b * N + qc) * D + head * 64 + 8 * h;
#pragma unroll
        for (int st = 0; st < 4; ++st) { qf[st] = load8(qrow + 16 * st); dof[st] = load8(drow + 16 * st); }
    }
    const float lse_q = lse[(size_t)bh * N + qc];
    const float del_q = delta[(size_t)bh * N + qc];
    f32x16 dq0 = zero16(), dq1 = zero16();

    const int nkt = (N + 63) >> 6;
    const int krow0 = b * N;
    stage64(rs, krow0, ld, D + head * 64, smem, wave, lane);
    stage64(rs, krow0, ld, 2 * D + head * 64, smem + 8192, wave, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) {
            char* nx = smem + ((kt + 1) & 1) * 16384;
            stage64(rs, krow0 + (kt + 1) * 64, ld, D + head * 64, nx, wave, lane);
            stage64(rs, krow0 + (kt + 1) * 64, ld, 2 * D + head * 64, nx + 8192, wave, lane);
        }
        const char* kl = smem + (kt & 1) * 16384;
        const char* vl = kl + 8192;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = kt * 64 + sub * 32;
            if (key0 >= N) break;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                s = MFMA32(frag_rows(kl, sub * 32, st, lane), qf[st], s);
                dp = MFMA32(frag_rows(vl, sub * 32, st, lane), dof[st], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = exp2f(s[r] * scale_log2 - lse_q);
                if (key0 + 32 > N && key0 + acc_row(r, h) >= N) p = 0.f;
                s[r] = p * (dp[r] - del_q);   // dS^T (without the 1/sqrt(d) factor, applied at the end)
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 df = acc_to_frag(s, ks);
                dq0 = MFMA32(frag_tr(kl, sub * 32, ks, 0, lane), df, dq0);
                dq1 = MFMA32(frag_tr(kl, sub * 32, ks, 32, lane), df, dq1);
            }
        }
    }
    if (qi < N) {
        bf16_t* orow = dqkv + (size_t)(b * N + qi) * ld + head * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * h;
            uint2 a = {pack2bf(dq0[4 * g] * scale, dq0[4 * g + 1] * scale), pack2bf(dq0[4 * g + 2] * scale, dq0[4 * g + 3] * scale)};
            uint2 c = {pack2bf(dq1[4 * g] * scale, dq1[4 * g + 1] * scale), pack2bf(dq1[4 * g + 2] * scale, dq1[4 * g + 3] * scale)};
            *reinterpret_cast<uint2*>(orow + d) = a;
            *reinterpret_cast<uint2*>(orow + 32 + d) = c;
        }
    }
}

// ============================================================================ dK, dV
// grid (ceil(N/128), B*H); wave w owns 32 keys; loops over query tiles (Q, dO, lse, delta staged)
// LDS stage = Q 8 KiB | dO 8 KiB | lse 256 B | delta 256 B
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int N, int H, int D,
                                                            uint32_t qkv_bytes, uint32_t dctx_bytes, uint32_t stat_bytes,
                                                            float scale, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STG = 8192 * 2 + 512;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const int ki = blockIdx.x * 128 + wave * 32 + (lane & 31);   // this lane's key
    const int kc = min(ki, N - 1);
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qkv, qkv_bytes);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dctx, dctx_bytes);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(lse, stat_bytes);
    const __amdgpu_buffer_rsrc_t re = make_rsrc(delta, stat_bytes);

    bf16x8 kf[4], vf[4];   // B operands of S = Q K^T and dP = dO V^T
    {
        const bf16_t* krow = qkv + (size_t)(b * N + kc) * ld + D + head * 64 + 8 * h;
#pragma unroll
        for (int st = 0; st < 4; ++st) { kf[st] = load8(krow + 16 * st); vf[st] = load8(krow + D + 16 * st); }
    }
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

    const int nqt = (N + 63) >> 6;
    const int qrow0 = b * N;
    auto stage = [&](int qt, char* dst) {
        stage64(rq, qrow0 + qt * 64, ld, head * 64, dst, wave, lane);
        stage64(rd, qrow0 + qt * 64, D, head * 64, dst + 8192, wave, lane);
        if (wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, LDS_PTR(dst + 16384), 4, (uint32_t)(((size_t)bh * N + qt * 64 + lane) * 4), 0, 0, 0);
        if (wave == 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(re, LDS_PTR(dst + 16384 + 256), 4, (uint32_t)(((size_t)bh * N + qt * 64 + lane) * 4), 0, 0, 0);
    };
    stage(0, smem);
    for (int qt = 0; qt < nqt; ++qt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 1 < nqt) stage(qt + 1, smem + ((qt + 1) & 1) * STG);
        const char* ql = smem + (qt & 1) * STG;
        const char* dl = ql + 8192;
        const __attribute__((address_space(3))) float* stl =
            (const __attribute__((address_space(3))) float*)((const __attribute__((address_space(3))) char*)(ql) + 16384);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int q0 = qt * 64 + sub * 32;
            if (q0 >= N) break;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                s = MFMA32(frag_rows(ql, sub * 32, st, lane), kf[st], s);
                dp = MFMA32(frag_rows(dl, sub * 32, st, lane), vf[st], dp);
            }
            // rows of s/dp are queries: registers 4g..4g+3 <-> queries q0 + 8 g + 4 h + 0..3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qq = sub * 32 + 8 * g + 4 * h;
                const f32x4 ls = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(stl + qq);
                const f32x4 de = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(stl + 64 + qq);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    float p = exp2f(s[r] * scale_log2 - ls[e]);
                    float ds = p * (dp[r] - de[e]);
                    if (q0 + 32 > N && q0 + 8 * g + 4 * h + e >= N) { p = 0.f; ds = 0.f; }
                    s[r] = p;
                    dp[r] = ds;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pf = acc_to_frag(s, ks);
                const bf16x8 df = acc_to_frag(dp, ks);
                dv0 = MFMA32(frag_tr(dl, sub * 32, ks, 0, lane), pf, dv0);
                dv1 = MFMA32(frag_tr(dl, sub * 32, ks, 32, lane), pf, dv1);
                dk0 = MFMA32(frag_tr(ql, sub * 32, ks, 0, lane), df, dk0);
                dk1 = MFMA32(frag_tr(ql, sub * 32, ks, 32, lane), df, dk1);
            }
        }
    }
    if (ki < N) {
        bf16_t* krow = dqkv + (size_t)(b * N + ki) * ld + D + head * 64;
        bf16_t* vrow = krow + D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * h;
            uint2 a = {pack2bf(dk0[4 * g] * scale, dk0[4 * g + 1] * scale), pack2bf(dk0[4 * g + 2] * scale, dk0[4 * g + 3] * scale)};
            uint2 c = {pack2bf(dk1[4 * g] * scale, dk1[4 * g + 1] * scale), pack2bf(dk1[4 * g + 2] * scale, dk1[4 * g + 3] * scale)};
            *reinterpret_cast<uint2*>(krow + d) = a;
            *reinterpret_cast<uint2*>(krow + 32 + d) = c;
            uint2 e = {pack2bf(dv0[4 * g], dv0[4 * g + 1]), pack2bf(dv0[4 * g + 2], dv0[4 * g + 3])};
            uint2 f = {pack2bf(dv1[4 * g], dv1[4 * g + 1]), pack2bf(dv1[4 * g + 2], dv1[4 * g + 3])};
            *reinterpret_cast<uint2*>(vrow + d) = e;
            *reinterpret_cast<uint2*>(vrow + 32 + d) = f;
        }
    }
}

// ============================================================================ host launchers
int launch_attn_fwd(const bf16_t* qkv, bf16_t* ctx, float* lse, int B, int N, int H, hipStream_t stream) {
    BVC_REQUIRE(B > 0 && N > 0 && H > 0, "attn_fwd: empty shape");
    const int D = H * 64;
    const size_t bytes = (size_t)B * N * 3 * D * 2;
    BVC_REQUIRE(bytes < 0xffffffffull, "attn_fwd: qkv larger than 4 GiB");
    const float scale_log2 = 0.125f * 1.4426950408889634f;
    dim3 grid((N + 127) / 128, B * H);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 32768, stream, qkv, ctx, lse, N, H, D, (uint32_t)bytes, scale_log2);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_attn_bwd(const bf16_t* qkv, const bf16_t* ctx, const bf16_t* dctx, const float* lse, float* delta,
                    bf16_t* dqkv, int B, int N, int H, hipStream_t stream) {
    BVC_REQUIRE(B > 0 && N > 0 && H > 0, "attn_bwd: empty shape");
    const int D = H * 64;
    const size_t bytes = (size_t)B * N * 3 * D * 2;
    BVC_REQUIRE(bytes < 0xffffffffull, "attn_bwd: qkv larger than 4 GiB");
    const float scale = 0.125f, scale_log2 = 0.125f * 1.4426950408889634f;
    {
        const long long total = (long long)B * N * H * 8;
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dctx, ctx, delta, B, N, H, D);
    }
    dim3 grid((N + 127) / 128, B * H);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), 2 * (16384 + 512), stream, qkv, dctx, lse, delta, dqkv, N, H, D,
                       (uint32_t)bytes, (uint32_t)((size_t)B * N * D * 2), (uint32_t)((size_t)B * H * N * 4), scale, scale_log2);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 32768, stream, qkv, dctx, lse, delta, dqkv, N, H, D,
                       (uint32_t)bytes, scale, scale_log2);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

}  // namespace bvc
