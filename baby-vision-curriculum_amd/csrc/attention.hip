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
// What bounds them (rocprofv3 --pmc at the decoder shape, 64 clips: profiles/r02_h_attention_single_wave.txt): the MFMA pipe is
// 0.39-0.47 busy at the ~2.1 GHz the chip holds, no LDS bank conflicts, LDS demand highest in dK/dV.  With 32 rows per wave every
// K / V (Q / dO) fragment read from LDS feeds ONE 32 x 32 score block (dQ: 12 KiB per 12 MFMAs).  A third wave per SIMD (dQ fits
// 168 registers) changes nothing (+-1 %).  The alternative - one wave = 96 rows with the whole register file, a third of the LDS reads
// per MFMA, software-pipelined by hand - was built and measured (experiment section below): 3-4 % at the decoder shape, nothing
// elsewhere; its ablated MFMA-only stream runs at 0.55-0.6 of the nominal rate, the same ceiling the 8192^3 GEMM shows, so these
// kernels are within 1.2-1.3x of what the chip sustains on random operands and the experiment is not the product path.
// The first version was VALU-issue bound (rocprofv3: 45 VALU per MFMA), so the loops are written to keep the VALU count down:
//   * __launch_bounds__(256, 2): <= 256 registers makes hipcc pick the VGPR form of the MFMA, so
//     accumulators are scaled / exponentiated in place (no v_accvgpr_read/write round trips);
//   * raw v_exp_f32 (arguments are <= 0 or bounded, no denormal fix-up), one v_cvt_pk_bf16_f32 per pair;
//   * every LDS fragment address is a per-lane constant computed once; stage / sub-tile / k-step are
//     immediate offsets (the tile loop is unrolled over the two LDS stages);
//   * the ragged last key tile is masked in its own branch, not with per-element selects;
//   * the O rescale is skipped while the running maximum grows by < 2^6 (P stays <= 64, exact in f32 sums).
//
// K/V (or Q/dO) tiles of 64 rows x 64 bf16 go HBM -> LDS by LDS-DMA into a 2-stage ring, XOR-swizzled
// so that both the ds_read_b128 row reads and the transposed reads of the same image are
// bank-conflict free (SQ_LDS_BANK_CONFLICT = 0 measured).
#include <stdlib.h>

#include "attention.h"

namespace bvc {

#define AS3 __attribute__((address_space(3)))
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// swizzle of the 16-B chunk index of row r in a [rows][64] bf16 tile (128-B rows); serves both
// ds_read_b128 row fragments (16 rows, one chunk) and tr reads (4 rows x 4 chunks).  Uses bits 1..3 of r
// only, so adding a multiple of 16 rows to r is a plain byte offset.
// HD = 64: 128-B rows, 8 chunks.  HD = 32: 64-B rows (4 per 256-B bank row), 4 chunks: XOR with bits 2..3 of r.
template <int HD>
__device__ __forceinline__ int swz_dual(int r) {
    if constexpr (HD == 64) return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
    else return (r >> 2) & 3;
}

// stage rows [row0, row0+64) x HD bf16 starting at element column `col0` of a [rows][ld] bf16 array
// into a 64 x HD LDS image (8 KiB / 4 KiB); pieces of 1 KiB, two / one per wave
template <int HD>
__device__ __forceinline__ void stage64(__amdgpu_buffer_rsrc_t rs, int row0, int ld, int col0, char* lds,
                                        int wave, int lane) {
    constexpr int CPR = HD / 8;                 // 16-B chunks per row
    constexpr int RPP = 64 / CPR;               // rows per 1 KiB piece
#pragma unroll
    for (int jj = 0; jj < HD / 32; ++jj) {
        const int j = wave + 4 * jj;
        const int r = RPP * j + lane / CPR;
        const int c = (lane % CPR) ^ swz_dual<HD>(r);
        const uint32_t off = (uint32_t)(((size_t)(row0 + r) * ld + col0 + c * 8) * 2);
        glds16(rs, off, (uint32_t)(size_t)((AS3 char*)lds) + (uint32_t)j * 1024u);     // asm: see common.h (no compiler-made drain)
    }
}

// Per-lane fragment addresses (byte offsets inside one 8 KiB tile image), computed once per kernel.
//   rows[st]    : row fragment of the 32x32x16 A operand, lane (r = l&31, h = l>>5) ->
//                 tile[r][16 st + 8 h + 0..7];   sub-tile s adds 4096 B
//   tr[t][half] : transposed fragment, lane (i = l&31, h) -> tile[4 h + q + 8 half (+16 ks)][32 t + i];
//                 k-step ks adds 2048 B, sub-tile s adds 4096 B
template <int HD>
struct FragAddr {
    uint32_t rows[HD / 16];
    uint32_t tr[HD / 32][2];
};

template <int HD>
__device__ __forceinline__ FragAddr<HD> make_frag_addr(int lane) {
    FragAddr<HD> a;
    constexpr int RB = HD * 2;   // bytes per tile row
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int st = 0; st < HD / 16; ++st) a.rows[st] = r * RB + (((2 * st + h) ^ swz_dual<HD>(r)) << 4);
    const int q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) {
        const int col = 32 * t + 16 * ((lane >> 4) & 1) + 4 * p;
        const int chunk = col >> 3, within = (col & 7) * 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r0 = 4 * h + q + 8 * half;
            a.tr[t][half] = r0 * RB + ((chunk ^ swz_dual<HD>(r0)) << 4) + within;
        }
    }
    return a;
}

template <int IMM>
__device__ __forceinline__ bf16x8 lds_rows(const AS3 char* base, uint32_t off) {
    return *reinterpret_cast<const AS3 bf16x8*>(base + off + IMM);
}

template <int IMM>
__device__ __forceinline__ bf16x8 lds_tr(const AS3 char* base, uint32_t off_lo, uint32_t off_hi) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 bf16x4*)(base + off_lo + IMM));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 bf16x4*)(base + off_hi + IMM));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
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

// hipcc keeps its own count of the vector-memory loads it knows; the LDS-DMA of these kernels is inline asm it does not see.
// Passing a plain load's result through an empty asm makes hipcc wait for that load HERE, before the first LDS-DMA is
// issued - otherwise its wait lands at the first use inside the tile loop and (one in-order counter) drains the prefetch too.
template <typename T>
__device__ __forceinline__ void settle(T& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// max / sum across the two lane halves (lanes l and l+32 hold the two halves of one softmax row)
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr float kDeferLog2 = 6.0f;   // rescale O only when the running max grows by more than 2^6

// ============================================================================ forward
template <int HD>
struct FwdState {
    f32x16 o[HD / 32];   // O^T row blocks of 32 head dims, column = query
    float m_run, l_run;  // running max (log2 units), this lane half's share of the row sum
};

// one 32-key sub-tile; KOFF / VOFF = byte offsets of the K and V images of the stage (+ sub-tile)
template <int HD, int KOFF, int VOFF>
__device__ __forceinline__ void fwd_subtile(const AS3 char* lds, const FragAddr<HD>& fa, const bf16x8 (&qf)[HD / 16], FwdState<HD>& st,
                                            int key0, int N, int h, float scale_log2) {
    constexpr int KS = 16 * HD * 2;
    f32x16 s = zero16();
#pragma unroll
    for (int stp = 0; stp < HD / 16; ++stp) s = MFMA32(lds_rows<KOFF>(lds, fa.rows[stp]), qf[stp], s);
    if (key0 + 32 > N) {   // ragged last tile only (workgroup-uniform branch)
        asm volatile("" ::: "memory");   // keep it a branch: if-converted, it costs 3 VALU per element on EVERY tile
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (key0 + acc_row(r, h) >= N) s[r] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = xhalf_max(mx) * scale_log2;
    if (!__all(mx - st.m_run <= kDeferLog2)) {   // wave-uniform: raise the running max, rescale what is accumulated
        const float m_new = fmaxf(st.m_run, mx);
        const float alpha = fast_exp2(st.m_run - m_new);
        st.m_run = m_new;
        st.l_run *= alpha;
#pragma unroll
        for (int t = 0; t < HD / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[t][r] *= alpha;
    }
    const float nm = -st.m_run;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s[r] = fast_exp2(fmaf(s[r], scale_log2, nm));
        rs += s[r];
    }
    st.l_run += rs;
    const bf16x8 p0 = acc_to_frag(s, 0), p1 = acc_to_frag(s, 1);
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) st.o[t] = MFMA32(lds_tr<VOFF>(lds, fa.tr[t][0], fa.tr[t][1]), p0, st.o[t]);
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) st.o[t] = MFMA32(lds_tr<VOFF + KS>(lds, fa.tr[t][0], fa.tr[t][1]), p1, st.o[t]);
}

// Block -> (tile, clip-head) map.  Workgroups are dealt round-robin to the 8 XCDs (block b and b+8 share an L2), so XCD x is
// given a contiguous run of logical ids: the tiles of one (clip, head), which all stream the same K/V (or Q/dO) rows, then
// meet in ONE L2 instead of eight.  Measured (profiles/r01_d_traffic_b16.json): with the plain (tile, head) grid the forward
// fetched 1.59 GB per step through the fabric against 0.37 GB of qkv.  remap = 0 keeps the plain order (A/B only).
__device__ __forceinline__ void attn_block(int tiles, int remap, int& tile, int& bh) {
    const int nb = gridDim.x, bid = blockIdx.x;
    int lid = bid;
    if (remap) {
        const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
        lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    }
    bh = lid / tiles;
    tile = lid - bh * tiles;
}

// grid ceil(N/128) * B*H (1-D, see attn_block); 256 threads; wave w owns queries q0 + 32 w .. + 31
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                          float* __restrict__ lse, int N, int H, int D,
                                                          uint32_t qkv_bytes, float scale_log2, int remap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (K image + V image)
    constexpr int IMG = 64 * HD * 2, STG = 2 * IMG, SUB = 32 * HD * 2;
    const AS3 char* lds = (const AS3 char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, bh;
    attn_block((N + 127) >> 7, remap, tile_, bh);
    const int b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const int qi = tile_ * 128 + wave * 32 + (lane & 31);   // this lane's query
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, qkv_bytes);
    const FragAddr<HD> fa = make_frag_addr<HD>(lane);

    bf16x8 qf[HD / 16];   // Q^T fragments (B operand of S^T = K Q^T): Q[qi][16 step + 8 h + 0..7]
    {
        const bf16_t* qrow = qkv + (size_t)(b * N + min(qi, N - 1)) * ld + head * HD + 8 * h;
#pragma unroll
        for (int st = 0; st < HD / 16; ++st) qf[st] = load8(qrow + 16 * st);
    }
    FwdState<HD> st;
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) st.o[t] = zero16();
    st.m_run = -INFINITY; st.l_run = 0.f;

    const int nkt = (N + 63) >> 6;
    const int krow0 = b * N;
    auto issue = [&](int kt, int stage) {
        stage64<HD>(rs, krow0 + kt * 64, ld, D + head * HD, smem + stage * STG, wave, lane);
        stage64<HD>(rs, krow0 + kt * 64, ld, 2 * D + head * HD, smem + stage * STG + IMG, wave, lane);
    };
#pragma unroll
    for (int stq = 0; stq < HD / 16; ++stq) settle(qf[stq]);
    issue(0, 0);
    for (int kt = 0; kt < nkt; kt += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) issue(kt + 1, 1);
        fwd_subtile<HD, 0, IMG>(lds, fa, qf, st, kt * 64, N, h, scale_log2);
        if (kt * 64 + 32 < N) fwd_subtile<HD, SUB, IMG + SUB>(lds, fa, qf, st, kt * 64 + 32, N, h, scale_log2);
        if (kt + 1 >= nkt) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < nkt) issue(kt + 2, 0);
        fwd_subtile<HD, STG, STG + IMG>(lds, fa, qf, st, kt * 64 + 64, N, h, scale_log2);
        if (kt * 64 + 96 < N) fwd_subtile<HD, STG + SUB, STG + IMG + SUB>(lds, fa, qf, st, kt * 64 + 96, N, h, scale_log2);
    }
    const float l_tot = xhalf_sum(st.l_run);
    const float inv = 1.f / l_tot;
    if (qi < N) {
        bf16_t* orow = ctx + (size_t)(b * N + qi) * D + head * HD;
#pragma unroll
        for (int t = 0; t < HD / 32; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * t + 8 * g + 4 * h;
                uint2 a = {pack2bf(st.o[t][4 * g] * inv, st.o[t][4 * g + 1] * inv), pack2bf(st.o[t][4 * g + 2] * inv, st.o[t][4 * g + 3] * inv)};
                *reinterpret_cast<uint2*>(orow + d) = a;
            }
        if (h == 0) lse[(size_t)bh * N + qi] = st.m_run + log2f(l_tot);
    }
}

// ============================================================================ dQ
template <int HD, int KOFF, int VOFF>
__device__ __forceinline__ void dq_subtile(const AS3 char* lds, const FragAddr<HD>& fa, const bf16x8 (&qf)[HD / 16],
                                           const bf16x8 (&dof)[HD / 16], f32x16 (&dq)[HD / 32], int key0, int N, int h,
                                           float scale_log2, float nlse, const f32x16& ndel) {
    constexpr int KS = 16 * HD * 2;
    // the dP^T chain starts from -delta (a loop-invariant register tuple: this lane's query is fixed), so dP - delta costs nothing here
    f32x16 s = zero16(), dp;
#pragma unroll
    for (int stp = 0; stp < HD / 16; ++stp) {
        s = MFMA32(lds_rows<KOFF>(lds, fa.rows[stp]), qf[stp], s);
        dp = MFMA32(lds_rows<VOFF>(lds, fa.rows[stp]), dof[stp], stp == 0 ? ndel : dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)   // dS^T (without the 1/sqrt(d) factor, applied at the end)
        s[r] = fast_exp2(fmaf(s[r], scale_log2, nlse)) * dp[r];
    if (key0 + 32 > N) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (key0 + acc_row(r, h) >= N) s[r] = 0.f;
    }
    const bf16x8 d0 = acc_to_frag(s, 0), d1 = acc_to_frag(s, 1);
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) dq[t] = MFMA32(lds_tr<KOFF>(lds, fa.tr[t][0], fa.tr[t][1]), d0, dq[t]);
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) dq[t] = MFMA32(lds_tr<KOFF + KS>(lds, fa.tr[t][0], fa.tr[t][1]), d1, dq[t]);
}

// grid (ceil(N/128), B*H); wave w owns 32 queries; loops over key tiles (K and V staged)
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                             const bf16_t* __restrict__ ctx, const float* __restrict__ lse,
                                                             float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H, int D,
                                                             uint32_t qkv_bytes, float scale, float scale_log2, int remap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IMG = 64 * HD * 2, STG = 2 * IMG, SUB = 32 * HD * 2;
    const AS3 char* lds = (const AS3 char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, bh;
    attn_block((N + 127) >> 7, remap, tile_, bh);
    const int b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const int qi = tile_ * 128 + wave * 32 + (lane & 31);
    const int qc = min(qi, N - 1);
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, qkv_bytes);
    const FragAddr<HD> fa = make_frag_addr<HD>(lane);

    bf16x8 qf[HD / 16], dof[HD / 16];
    float del_q = 0.f;
    {
        const bf16_t* qrow = qkv + (size_t)(b * N + qc) * ld + head * HD + 8 * h;
        const bf16_t* drow = dctx + (size_t)(b * N + qc) * D + head * HD + 8 * h;
        const bf16_t* orow_in = ctx + (size_t)(b * N + qc) * D + head * HD + 8 * h;
#pragma unroll
        for (int st = 0; st < HD / 16; ++st) {
            qf[st] = load8(qrow + 16 * st);
            dof[st] = load8(drow + 16 * st);
            // delta = rowsum(dO * O) of this query (the softmax-gradient correction): the two half-waves hold disjoint halves
            // of the row, so it costs one more 16-B load per step here instead of a pass of its own over dO and O
            const bf16x8 of = load8(orow_in + 16 * st);
#pragma unroll
            for (int j = 0; j < 8; ++j) del_q += bf2f((bf16_t)dof[st][j]) * bf2f((bf16_t)of[j]);
        }
    }
    del_q += __shfl_xor(del_q, 32, 64);
    if (h == 0 && qi < N) delta[(size_t)bh * N + qi] = -del_q;     // NEGATED: the dK/dV kernel, launched after this one, starts its dP chain from it
    float nlse = -lse[(size_t)bh * N + qc];
#pragma unroll
    for (int stq = 0; stq < HD / 16; ++stq) { settle(qf[stq]); settle(dof[stq]); }
    settle(nlse); settle(del_q);
    f32x16 ndel;
#pragma unroll
    for (int r = 0; r < 16; ++r) ndel[r] = -del_q;
    f32x16 dq[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) dq[t] = zero16();

    const int nkt = (N + 63) >> 6;
    const int krow0 = b * N;
    auto issue = [&](int kt, int stage) {
        stage64<HD>(rs, krow0 + kt * 64, ld, D + head * HD, smem + stage * STG, wave, lane);
        stage64<HD>(rs, krow0 + kt * 64, ld, 2 * D + head * HD, smem + stage * STG + IMG, wave, lane);
    };
    issue(0, 0);
    for (int kt = 0; kt < nkt; kt += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) issue(kt + 1, 1);
        dq_subtile<HD, 0, IMG>(lds, fa, qf, dof, dq, kt * 64, N, h, scale_log2, nlse, ndel);
        if (kt * 64 + 32 < N) dq_subtile<HD, SUB, IMG + SUB>(lds, fa, qf, dof, dq, kt * 64 + 32, N, h, scale_log2, nlse, ndel);
        if (kt + 1 >= nkt) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < nkt) issue(kt + 2, 0);
        dq_subtile<HD, STG, STG + IMG>(lds, fa, qf, dof, dq, kt * 64 + 64, N, h, scale_log2, nlse, ndel);
        if (kt * 64 + 96 < N)
            dq_subtile<HD, STG + SUB, STG + IMG + SUB>(lds, fa, qf, dof, dq, kt * 64 + 96, N, h, scale_log2, nlse, ndel);
    }
    if (qi < N) {
        bf16_t* orow = dqkv + (size_t)(b * N + qi) * ld + head * HD;
#pragma unroll
        for (int t = 0; t < HD / 32; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * t + 8 * g + 4 * h;
                uint2 a = {pack2bf(dq[t][4 * g] * scale, dq[t][4 * g + 1] * scale), pack2bf(dq[t][4 * g + 2] * scale, dq[t][4 * g + 3] * scale)};
                *reinterpret_cast<uint2*>(orow + d) = a;
            }
    }
}

// ============================================================================ one wave = NB x 32 rows (experiment, round 2)
// NOT on the product path: compiled only into an experiments build (-DBVC_EXPERIMENTS) and selected there by BVC_ATTN_DQ_W=1, for
// the A/B recorded in profiles/r02_h_attention_single_wave.txt.  Result: 3-4 % faster than the 32-row dQ kernel at the decoder
// shape, equal at N = 160, 15 % slower at head_dim 32 - not worth its inline-asm hazard contract, so the 32-row kernels stay.
#ifdef BVC_EXPERIMENTS
// The kernels above give every wave 32 rows and read each K / V (or Q / dO) fragment from LDS for ONE 32 x 32 score block:
// 16 LDS instructions (12 KiB) per 12 MFMAs in dQ (counters: profiles/r02_h_attention_single_wave.txt).
// The kernel below gives ONE wave NB = 3 blocks of 32 rows (96 queries) and the whole 512-register file
// (one wave per SIMD, four single-wave workgroups per CU): every fragment read from LDS feeds NB MFMAs, the NB blocks are
// independent dependency chains (one block's softmax arithmetic runs beside another block's MFMAs), and with one wave per
// workgroup there is no barrier at all: the K / V (Q / dO) stream is a private 4-stage ring of 32-row tiles filled by LDS-DMA
// three tiles ahead behind a counted vmcnt.
constexpr int kRing = 4;          // LDS stages of one 32-row tile pair; prefetch distance kRing - 1

template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() {
#if defined(BVC_ATTN_ABL) && (BVC_ATTN_ABL & 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// per-lane source byte offsets of the HD/16 pieces (1 KiB of LDS each) of a 32 x HD tile image, for tile row 0
template <int HD>
struct TileSrc { uint32_t v[HD / 16]; };
template <int HD>
__device__ __forceinline__ TileSrc<HD> make_tile_src(int lane, int ld, int col0) {
    constexpr int CPR = HD / 8, RPP = 64 / CPR;   // 16-B chunks per row, rows per piece
    TileSrc<HD> t;
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) {
        const int r = RPP * j + lane / CPR;
        const int c = (lane % CPR) ^ swz_dual<HD>(r);
        t.v[j] = (uint32_t)((r * ld + col0 + c * 8) * 2);
    }
    return t;
}
// rows [row, row + 32) of the source (row_bytes = row * ld * 2, wave-uniform) -> the image at lds_addr; soff = wave-uniform extra bytes
template <int HD>
__device__ __forceinline__ void stage32(__amdgpu_buffer_rsrc_t rs, const TileSrc<HD>& src, uint32_t row_bytes, uint32_t soff, uint32_t lds_addr) {
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) glds16s(rs, src.v[j] + row_bytes, soff, lds_addr + 1024u * j);
}

// acc += A B with the accumulator pinned to the AGPR half of the register file (inline asm: the compiler's own MFMA selection is
// all-VGPR or all-AGPR per function, and these kernels need both: the score accumulators are VALU operands, the 96 registers of
// dQ^T / dK^T / dV^T are touched by MFMAs only).  hipcc does not know this is an MFMA, so the CALLER keeps the hazard distances:
// a dependent MFMA on the same accumulator at least two MFMA slots later, operands that no VALU instruction wrote in the last two
// issue slots, a VALU read of the accumulator only after `mfma_drain()`.
__device__ __forceinline__ void mfma_acc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the same with two wait states in front: for an operand a VALU instruction may have written just before (the hazard hipcc covers
// with an s_nop for its own MFMAs)
__device__ __forceinline__ void mfma_acc_safe(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// ---------------------------------------------------------------------------- dQ, one wave = NB x 32 queries
// One wave per SIMD: nothing but this wave's own instruction order overlaps softmax arithmetic, LDS latency and DMA issue with
// the MFMAs (left to hipcc, each block's 8 score MFMAs issue back to back and its 64 VALU after them: no faster than the 32-row
// kernel).  The stream is therefore software-pipelined by hand and pinned with sched_barrier(0).  Per 32-key tile t and query
// block b: M1(b,t) = 8 score MFMAs (S^T and dP^T chains alternating), V(b,t) = 8 steps of two score elements each (fma, exp, sub,
// mul, one packed convert: ~40 issue cycles), M2(b,t) = 4 MFMAs of the dQ^T update.  Phase b of tile t issues
//     M1(b+1, t)  [b = NB-1: M1(0, t+1)]   beside   V(b, t),   then   M2(b-1, t)   [b = 0: none],
// one MFMA per slot with a V step in two slots of three, and a last short phase issues M2(NB-1, t).  The slots without a V step
// carry the fragment reads (K^T of tile t in phase 0, the K / V rows of tile t+1 at the end of phase 1, into the registers M1(NB-1, t)
// has just finished with) and the LDS-DMA of tile t+3.  Every MFMA operand and every V step's scores are at least four slots old.
// The pipelined loop covers the full tiles; a ragged last tile (N mod 32 keys) runs once through `dq_tile_rag` after it.
template <int HD, int NB>
struct DqState {
    static constexpr int NS = HD / 16, NT = HD / 32;
    bf16x8 kr[NS], vr[NS];             // K / V row fragments (A operands of S^T, dP^T)
    bf16x8 kt[2][NT];                  // K^T fragments (A operand of dQ^T), by k-step, head-dim block
    f32x16 s[NB], dp[NB];
    union { bf16x8 v; uint32_t u[4]; } d[NB][2];
};

#define BVC_PIN() __builtin_amdgcn_sched_barrier(0)
// S = ring stage of tile t (t mod 4)
template <int HD, int NB, int S, typename Dma>
__device__ __forceinline__ void dq_tile_r(DqState<HD, NB>& st, const AS3 char* lds, const FragAddr<HD>& fa, const bf16x8 (&qf)[NB][HD / 16],
                                          const bf16x8 (&dof)[NB][HD / 16], f32x16 (&dq)[NB][HD / 32], float scale_log2,
                                          const float (&nlse)[NB], const float (&del_q)[NB], Dma&& dma) {
    constexpr int IMG = 32 * HD * 2, STG = 2 * IMG, KS = 16 * HD * 2, NS = HD / 16, NT = HD / 32;
    constexpr int KOFF = S * STG, KNXT = ((S + 1) & (kRing - 1)) * STG, VNXT = KNXT + IMG;
    constexpr int M1N = 2 * NS, M2N = 2 * NT, PCS = 2 * NS;
    auto m1 = [&](int blk, int i) {     // score MFMA i of block blk
        const int stp = i >> 1;
        if ((i & 1) == 0) st.s[blk] = MFMA32(st.kr[stp], qf[blk][stp], stp == 0 ? zero16() : st.s[blk]);
        else st.dp[blk] = MFMA32(st.vr[stp], dof[blk][stp], stp == 0 ? zero16() : st.dp[blk]);
    };
    auto m2 = [&](int blk, int i) {
        const int ks = i / NT, t = i % NT;
        mfma_acc(dq[blk][t], st.kt[ks][t], st.d[blk][ks].v);
        if (NT == 1) mfma_drain();      // head_dim 32: the two updates of a block hit the same accumulator back to back
    };
    // elements 2i, 2i+1 of block blk: dS^T (without the 1/sqrt(d) factor, applied at the end) -> one packed word of its fragment
    auto vstep = [&](int blk, int i) {
#if defined(BVC_ATTN_ABL) && (BVC_ATTN_ABL & 2)
        const float a = st.s[blk][2 * i] + st.dp[blk][2 * i], c = st.s[blk][2 * i + 1] + st.dp[blk][2 * i + 1];
#else
        const float a = fast_exp2(fmaf(st.s[blk][2 * i], scale_log2, nlse[blk])) * (st.dp[blk][2 * i] - del_q[blk]);
        const float c = fast_exp2(fmaf(st.s[blk][2 * i + 1], scale_log2, nlse[blk])) * (st.dp[blk][2 * i + 1] - del_q[blk]);
#endif
        st.d[blk][i >> 2].u[i & 3] = pack2bf(a, c);
    };
    int piece = 0;                      // DMA instructions of tile t+3 issued so far
#pragma unroll
    for (int ph = 0; ph < NB; ++ph) {
        const int slots = M1N + (ph > 0 ? M2N : 0);
        int vs = 0;
#pragma unroll
        for (int i = 0; i < slots; ++i) {
            if (i < M1N) m1(ph + 1 < NB ? ph + 1 : 0, i);
            else m2(ph - 1, i - M1N);
            const bool free_slot = (i % 3 == 2) || i >= M1N;
            if (!free_slot || slots == M1N) { if (vs < 8) vstep(ph, vs++); }
            constexpr int KT1 = M1N >= 8 ? 4 : M1N - 1;      // slot of the second K^T k-step
#if defined(BVC_ATTN_ABL) && (BVC_ATTN_ABL & 4)
            constexpr bool kReload = false;
#else
            constexpr bool kReload = true;
#endif
            if (kReload && ph == 0 && (i == 1 || i == KT1)) {          // K^T of this tile (first needed by M2(0, t) in phase 1)
                const int ks = i == 1 ? 0 : 1;
#pragma unroll
                for (int t = 0; t < NT; ++t) st.kt[ks][t] = lds_tr<KOFF>(lds, fa.tr[t][0] + ks * KS, fa.tr[t][1] + ks * KS);
            }
            if (kReload && ph == NB - 2 && i >= M1N) {                  // rows of tile t+1: M1(NB-1, t) has issued, M1(0, t+1) opens the next phase
                const int k = i - M1N;                       // 0 .. M2N-1
#pragma unroll
                for (int stp = k * NS / M2N; stp < (k + 1) * NS / M2N; ++stp) st.kr[stp] = lds_rows<KNXT>(lds, fa.rows[stp]);
#pragma unroll
                for (int stp = k * NS / M2N; stp < (k + 1) * NS / M2N; ++stp) st.vr[stp] = lds_rows<VNXT>(lds, fa.rows[stp]);
            }
            if (ph == NB - 1 && free_slot && piece < PCS) dma(piece++);
            BVC_PIN();
        }
#pragma unroll
        for (; vs < 8; ++vs) { vstep(ph, vs); BVC_PIN(); }
    }
#pragma unroll
    for (int i = 0; i < M2N; ++i) {
        m2(NB - 1, i);
#pragma unroll
        for (int k = 0; k < 2; ++k) if (piece < PCS) dma(piece++);
        BVC_PIN();
    }
#pragma unroll
    for (; piece < PCS; ++piece) dma(piece);
}

// the ragged last tile (keys >= N masked), once per workgroup and not pipelined; `stage` = byte offset of its ring stage
template <int HD, int NB>
__device__ __forceinline__ void dq_tile_rag(const AS3 char* lds, uint32_t stage, const FragAddr<HD>& fa, const bf16x8 (&qf)[NB][HD / 16],
                                         const bf16x8 (&dof)[NB][HD / 16], f32x16 (&dq)[NB][HD / 32], int key0, int N, int h,
                                         float scale_log2, const float (&nlse)[NB], const float (&del_q)[NB]) {
    constexpr int IMG = 32 * HD * 2, KS = 16 * HD * 2, NS = HD / 16, NT = HD / 32;
    bf16x8 kr[NS], vr[NS], kt[2][NT];
#pragma unroll
    for (int stp = 0; stp < NS; ++stp) { kr[stp] = lds_rows<0>(lds, fa.rows[stp] + stage); vr[stp] = lds_rows<IMG>(lds, fa.rows[stp] + stage); }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        kt[0][t] = lds_tr<0>(lds, fa.tr[t][0] + stage, fa.tr[t][1] + stage);
        kt[1][t] = lds_tr<KS>(lds, fa.tr[t][0] + stage, fa.tr[t][1] + stage);
    }
    union { bf16x8 v; uint32_t u[4]; } d[NB][2];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int stp = 0; stp < NS; ++stp) {
            s = MFMA32(kr[stp], qf[blk][stp], s);
            dp = MFMA32(vr[stp], dof[blk][stp], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(fmaf(s[r], scale_log2, nlse[blk])) * (dp[r] - del_q[blk]);
            if (key0 + acc_row(r, h) >= N) s[r] = 0.f;
        }
        d[blk][0].v = acc_to_frag(s, 0);
        d[blk][1].v = acc_to_frag(s, 1);
    }
    // block-major inside each (k-step, head-dim block): MFMAs on one accumulator stay NB slots apart
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) mfma_acc_safe(dq[blk][i % NT], kt[i / NT][i % NT], d[blk][i / NT].v);
}

// grid ceil(N / (32 NB)) * B*H single-wave workgroups; lane (i, h) owns query column i of each of the wave's NB blocks
template <int HD, int NB>
__global__ __launch_bounds__(64) void attn_bwd_dq_w_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                           const bf16_t* __restrict__ ctx, const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N, int H, int D,
                                                           uint32_t qkv_bytes, float scale, float scale_log2, int remap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IMG = 32 * HD * 2, STG = 2 * IMG, NS = HD / 16, NT = HD / 32, PCS = 2 * NS;   // PCS: DMA instructions per tile (K + V)
    const AS3 char* lds = (const AS3 char*)smem;
    const uint32_t lds0 = (uint32_t)(size_t)((AS3 char*)smem);
    const int lane = threadIdx.x;
    int tile_, bh;
    attn_block((N + 32 * NB - 1) / (32 * NB), remap, tile_, bh);
    const int b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, qkv_bytes);
    const FragAddr<HD> fa = make_frag_addr<HD>(lane);

    bf16x8 qf[NB][NS], dof[NB][NS];
    float del_q[NB], nlse[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int qi = tile_ * 32 * NB + 32 * blk + (lane & 31);
        const int qc = min(qi, N - 1);
        const bf16_t* qrow = qkv + (size_t)(b * N + qc) * ld + head * HD + 8 * h;
        const bf16_t* drow = dctx + (size_t)(b * N + qc) * D + head * HD + 8 * h;
        const bf16_t* orow_in = ctx + (size_t)(b * N + qc) * D + head * HD + 8 * h;
        float dl = 0.f;
#pragma unroll
        for (int stp = 0; stp < NS; ++stp) {
            qf[blk][stp] = load8(qrow + 16 * stp);
            dof[blk][stp] = load8(drow + 16 * stp);
            // delta = rowsum(dO * O) of this query: the two half-waves hold disjoint halves of the row
            const bf16x8 of = load8(orow_in + 16 * stp);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f((bf16_t)dof[blk][stp][j]) * bf2f((bf16_t)of[j]);
        }
        dl += __shfl_xor(dl, 32, 64);
        if (h == 0 && qi < N) delta[(size_t)bh * N + qi] = -dl;     // negated, as attn_bwd_dq_kernel stores it
        del_q[blk] = dl;
        nlse[blk] = -lse[(size_t)bh * N + qc];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads and the delta stores above leave the counter before the stream starts
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
        for (int stq = 0; stq < NS; ++stq) { settle(qf[blk][stq]); settle(dof[blk][stq]); }
        settle(nlse[blk]); settle(del_q[blk]);
    }
    f32x16 dq[NB][NT];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int t = 0; t < NT; ++t) { dq[blk][t] = zero16(); asm volatile("" : "+a"(dq[blk][t])); }   // zeroed HERE, not lazily in front of the first
    mfma_drain();                                                                                       // update (AGPR write -> MFMA read is a hazard hipcc cannot see)

    const int nt = (N + 31) >> 5, nfull = N >> 5;
    const TileSrc<HD> ksrc = make_tile_src<HD>(lane, ld, D + head * HD);
    const uint32_t tile_bytes = (uint32_t)(32 * ld * 2), row0_bytes = (uint32_t)((size_t)b * N * ld * 2), vshift = (uint32_t)(2 * D);
    // DMA instruction j (0 .. PCS-1) of tile t: the K pieces, then the V pieces (same rows, + 2 D bytes).  Tiles past the clip are
    // issued too (their rows are the next clip's, or out of range -> zeros): the counted waits then need no tail cases.
    auto dma_piece = [&](int t, int j) {
#if defined(BVC_ATTN_ABL) && (BVC_ATTN_ABL & 1)
        if (t >= kRing - 1) return;
#endif
        const uint32_t dst = lds0 + (uint32_t)(t & (kRing - 1)) * STG + (j >= NS ? IMG : 0) + 1024u * (j % NS);
        glds16s(rs, ksrc.v[j % NS] + row0_bytes + (uint32_t)t * tile_bytes, j >= NS ? vshift : 0u, dst);
    };
    DqState<HD, NB> st;
#pragma unroll
    for (int t = 0; t < kRing - 1; ++t)
#pragma unroll
        for (int j = 0; j < PCS; ++j) dma_piece(t, j);
    if (nfull > 0) {
        attn_wait_vmcnt<2 * PCS>();        // tile 0 has landed
#pragma unroll
        for (int stp = 0; stp < NS; ++stp) { st.kr[stp] = lds_rows<0>(lds, fa.rows[stp]); st.vr[stp] = lds_rows<IMG>(lds, fa.rows[stp]); }
#pragma unroll
        for (int i = 0; i < 2 * NS; ++i) {
            const int stp = i >> 1;
            if ((i & 1) == 0) st.s[0] = MFMA32(st.kr[stp], qf[0][stp], stp == 0 ? zero16() : st.s[0]);
            else st.dp[0] = MFMA32(st.vr[stp], dof[0][stp], stp == 0 ? zero16() : st.dp[0]);
        }
#if defined(BVC_ATTN_ABL) && (BVC_ATTN_ABL & 4)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) { st.kt[0][tt] = lds_tr<0>(lds, fa.tr[tt][0], fa.tr[tt][1]); st.kt[1][tt] = lds_tr<16 * HD * 2>(lds, fa.tr[tt][0], fa.tr[tt][1]); }
#endif
        // top of tile t: tile t+1 must have landed (its fragments are read during tile t); outstanding: tile t+2 only
        int t = 0;
        attn_wait_vmcnt<PCS>();
        dq_tile_r<HD, NB, 0>(st, lds, fa, qf, dof, dq, scale_log2, nlse, del_q, [&](int j) { dma_piece(t + 3, j); });
        for (t = 1; t < nfull;) {
            attn_wait_vmcnt<PCS>(); dq_tile_r<HD, NB, 1>(st, lds, fa, qf, dof, dq, scale_log2, nlse, del_q, [&](int j) { dma_piece(t + 3, j); });
            if (++t >= nfull) break;
            attn_wait_vmcnt<PCS>(); dq_tile_r<HD, NB, 2>(st, lds, fa, qf, dof, dq, scale_log2, nlse, del_q, [&](int j) { dma_piece(t + 3, j); });
            if (++t >= nfull) break;
            attn_wait_vmcnt<PCS>(); dq_tile_r<HD, NB, 3>(st, lds, fa, qf, dof, dq, scale_log2, nlse, del_q, [&](int j) { dma_piece(t + 3, j); });
            if (++t >= nfull) break;
            attn_wait_vmcnt<PCS>(); dq_tile_r<HD, NB, 0>(st, lds, fa, qf, dof, dq, scale_log2, nlse, del_q, [&](int j) { dma_piece(t + 3, j); });
            ++t;
        }
    }
    mfma_drain();                                       // before anything hipcc may place here touches the accumulators
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the workgroup's LDS is released
    if (nt > nfull) dq_tile_rag<HD, NB>(lds, (uint32_t)(nfull & (kRing - 1)) * STG, fa, qf, dof, dq, 32 * nfull, N, h, scale_log2, nlse, del_q);
    mfma_drain();                                       // the accumulators are read by VALU next

#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int qi = tile_ * 32 * NB + 32 * blk + (lane & 31);
        if (qi < N) {
            bf16_t* orow = dqkv + (size_t)(b * N + qi) * ld + head * HD;
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * t2 + 8 * g + 4 * h;
                    uint2 a = {pack2bf(dq[blk][t2][4 * g] * scale, dq[blk][t2][4 * g + 1] * scale),
                               pack2bf(dq[blk][t2][4 * g + 2] * scale, dq[blk][t2][4 * g + 3] * scale)};
                    *reinterpret_cast<uint2*>(orow + d) = a;
                }
        }
    }
}
#undef BVC_PIN
#endif  // BVC_EXPERIMENTS

// ============================================================================ dK, dV
// LDS stage = Q image | dO image | lse 256 B | -delta 256 B
template <int HD, int QOFF, int STAT>
__device__ __forceinline__ void dkdv_subtile(const AS3 char* lds, const FragAddr<HD>& fa, const bf16x8 (&kf)[HD / 16],
                                             const bf16x8 (&vf)[HD / 16], f32x16 (&dk)[HD / 32], f32x16 (&dv)[HD / 32], int q0,
                                             int N, int h, float scale_log2) {
    constexpr int IMG = 64 * HD * 2, KS = 16 * HD * 2;
    constexpr int DOFF = QOFF + IMG;
    // rows of s/dp are queries: registers 4g..4g+3 <-> queries q0 + 8 g + 4 h + 0..3 (lse / -delta broadcast from LDS).
    // The dP chain starts from -delta (the dQ kernel stores it negated): four 16-byte LDS reads ARE the initial accumulator.
    const AS3 float* stl = reinterpret_cast<const AS3 float*>(lds + STAT) + 4 * h;
    f32x16 s = zero16(), dp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 nd = *reinterpret_cast<const AS3 f32x4*>(stl + 64 + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * g + e] = nd[e];
    }
#pragma unroll
    for (int stp = 0; stp < HD / 16; ++stp) {
        s = MFMA32(lds_rows<QOFF>(lds, fa.rows[stp]), kf[stp], s);
        dp = MFMA32(lds_rows<DOFF>(lds, fa.rows[stp]), vf[stp], dp);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ls = *reinterpret_cast<const AS3 f32x4*>(stl + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float p = fast_exp2(fmaf(s[r], scale_log2, -ls[e]));
            s[r] = p;
            dp[r] = p * dp[r];
        }
    }
    if (q0 + 32 > N) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (q0 + acc_row(r, h) >= N) { s[r] = 0.f; dp[r] = 0.f; }
    }
    const bf16x8 p0 = acc_to_frag(s, 0), p1 = acc_to_frag(s, 1);
    const bf16x8 d0 = acc_to_frag(dp, 0), d1 = acc_to_frag(dp, 1);
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) {
        dv[t] = MFMA32(lds_tr<DOFF>(lds, fa.tr[t][0], fa.tr[t][1]), p0, dv[t]);
        dk[t] = MFMA32(lds_tr<QOFF>(lds, fa.tr[t][0], fa.tr[t][1]), d0, dk[t]);
    }
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) {
        dv[t] = MFMA32(lds_tr<DOFF + KS>(lds, fa.tr[t][0], fa.tr[t][1]), p1, dv[t]);
        dk[t] = MFMA32(lds_tr<QOFF + KS>(lds, fa.tr[t][0], fa.tr[t][1]), d1, dk[t]);
    }
}

// grid (ceil(N/128), B*H); wave w owns 32 keys; loops over query tiles (Q, dO, lse, delta staged)
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, int N, int H, int D,
                                                               uint32_t qkv_bytes, uint32_t dctx_bytes, uint32_t stat_bytes,
                                                               float scale, float scale_log2, int remap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IMG = 64 * HD * 2, SUB = 32 * HD * 2, STG = 2 * IMG + 512;
    const AS3 char* lds = (const AS3 char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, bh;
    attn_block((N + 127) >> 7, remap, tile_, bh);
    const int b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const int ki = tile_ * 128 + wave * 32 + (lane & 31);   // this lane's key
    const int kc = min(ki, N - 1);
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qkv, qkv_bytes);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dctx, dctx_bytes);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(lse, stat_bytes);
    const __amdgpu_buffer_rsrc_t re = make_rsrc(delta, stat_bytes);
    const FragAddr<HD> fa = make_frag_addr<HD>(lane);

    bf16x8 kf[HD / 16], vf[HD / 16];   // B operands of S = Q K^T and dP = dO V^T
    {
        const bf16_t* krow = qkv + (size_t)(b * N + kc) * ld + D + head * HD + 8 * h;
#pragma unroll
        for (int st = 0; st < HD / 16; ++st) { kf[st] = load8(krow + 16 * st); vf[st] = load8(krow + D + 16 * st); }
    }
#pragma unroll
    for (int stq = 0; stq < HD / 16; ++stq) { settle(kf[stq]); settle(vf[stq]); }
    f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t) { dk[t] = zero16(); dv[t] = zero16(); }

    const int nqt = (N + 63) >> 6;
    const int qrow0 = b * N;
    auto issue = [&](int qt, int stage) {
        char* dst = smem + stage * STG;
        stage64<HD>(rq, qrow0 + qt * 64, ld, head * HD, dst, wave, lane);
        stage64<HD>(rd, qrow0 + qt * 64, D, head * HD, dst + IMG, wave, lane);
        if (wave == 0)
            glds4(rl, (uint32_t)(((size_t)bh * N + qt * 64 + lane) * 4), (uint32_t)(size_t)((AS3 char*)dst) + 2 * IMG);
        if (wave == 1)
            glds4(re, (uint32_t)(((size_t)bh * N + qt * 64 + lane) * 4), (uint32_t)(size_t)((AS3 char*)dst) + 2 * IMG + 256);
    };
    issue(0, 0);
    for (int qt = 0; qt < nqt; qt += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 1 < nqt) issue(qt + 1, 1);
        dkdv_subtile<HD, 0, 2 * IMG>(lds, fa, kf, vf, dk, dv, qt * 64, N, h, scale_log2);
        if (qt * 64 + 32 < N) dkdv_subtile<HD, SUB, 2 * IMG + 128>(lds, fa, kf, vf, dk, dv, qt * 64 + 32, N, h, scale_log2);
        if (qt + 1 >= nqt) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 2 < nqt) issue(qt + 2, 0);
        dkdv_subtile<HD, STG, STG + 2 * IMG>(lds, fa, kf, vf, dk, dv, qt * 64 + 64, N, h, scale_log2);
        if (qt * 64 + 96 < N) dkdv_subtile<HD, STG + SUB, STG + 2 * IMG + 128>(lds, fa, kf, vf, dk, dv, qt * 64 + 96, N, h, scale_log2);
    }
    if (ki < N) {
        bf16_t* krow = dqkv + (size_t)(b * N + ki) * ld + D + head * HD;
        bf16_t* vrow = krow + D;
#pragma unroll
        for (int t = 0; t < HD / 32; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * t + 8 * g + 4 * h;
                uint2 a = {pack2bf(dk[t][4 * g] * scale, dk[t][4 * g + 1] * scale), pack2bf(dk[t][4 * g + 2] * scale, dk[t][4 * g + 3] * scale)};
                *reinterpret_cast<uint2*>(krow + d) = a;
                uint2 e = {pack2bf(dv[t][4 * g], dv[t][4 * g + 1]), pack2bf(dv[t][4 * g + 2], dv[t][4 * g + 3])};
                *reinterpret_cast<uint2*>(vrow + d) = e;
            }
    }
}

// ============================================================================ host launchers
// BVC_ATTN_PLAIN_GRID=1 switches the XCD-aware block map off (same-run A/B in tools/microbench.py; read per launch)
static int xcd_remap() {
#ifdef BVC_EXPERIMENTS      // same-process A/B of the XCD-aware block map (tools/microbench.py)
    return getenv("BVC_ATTN_PLAIN_GRID") == nullptr;
#else
    return 1;
#endif
}

template <int HD>
static int fwd_hd(const bf16_t* qkv, bf16_t* ctx, float* lse, int B, int N, int H, hipStream_t stream, float sm_scale) {
    const int D = H * HD;
    const size_t bytes = (size_t)B * N * 3 * D * 2;
    const float scale_log2 = (sm_scale > 0.f ? sm_scale : 1.0f / sqrtf((float)HD)) * 1.4426950408889634f;
    const dim3 grid((unsigned)(((N + 127) / 128) * B * H));
    hipLaunchKernelGGL(attn_fwd_kernel<HD>, grid, dim3(256), 4 * 64 * HD * 2, stream, qkv, ctx, lse, N, H, D, (uint32_t)bytes, scale_log2,
                       xcd_remap());
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

template <int HD>
static int bwd_hd(const bf16_t* qkv, const bf16_t* ctx, const bf16_t* dctx, const float* lse, float* delta, bf16_t* dqkv, int B,
                  int N, int H, hipStream_t stream, float sm_scale, int parts) {
    const int D = H * HD;
    const size_t bytes = (size_t)B * N * 3 * D * 2;
    const float scale = sm_scale > 0.f ? sm_scale : 1.0f / sqrtf((float)HD), scale_log2 = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)(((N + 127) / 128) * B * H));
    const int remap = xcd_remap();
    // dQ first: it also produces delta = rowsum(dO * O), which the dK/dV kernel consumes
    if (parts & 1) {
#ifdef BVC_EXPERIMENTS
    if (getenv("BVC_ATTN_DQ_W") != nullptr) {      // the single-wave 96-query dQ kernel (see above): same-process A/B only
        constexpr int NB = 3;
        const dim3 gridw((unsigned)(((N + 32 * NB - 1) / (32 * NB)) * B * H));
        hipLaunchKernelGGL((attn_bwd_dq_w_kernel<HD, NB>), gridw, dim3(64), kRing * 2 * 32 * HD * 2, stream, qkv, dctx, ctx, lse, delta, dqkv, N, H, D,
                           (uint32_t)bytes, scale, scale_log2, remap);
    } else
#endif
    hipLaunchKernelGGL(attn_bwd_dq_kernel<HD>, grid, dim3(256), 4 * 64 * HD * 2, stream, qkv, dctx, ctx, lse, delta, dqkv, N, H, D,
                       (uint32_t)bytes, scale, scale_log2, remap);
    }
    if (parts & 2)
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<HD>, grid, dim3(256), 2 * (2 * 64 * HD * 2 + 512), stream, qkv, dctx, lse, delta, dqkv, N, H, D,
                       (uint32_t)bytes, (uint32_t)((size_t)B * N * D * 2), (uint32_t)((size_t)B * H * N * 4), scale, scale_log2, remap);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_attn_fwd(const bf16_t* qkv, bf16_t* ctx, float* lse, int B, int N, int H, int head_dim, hipStream_t stream, float sm_scale) {
    BVC_REQUIRE(B > 0 && N > 0 && H > 0, "attn_fwd: empty shape");
    BVC_REQUIRE(head_dim == 64 || head_dim == 32, "attn_fwd: head_dim %d unsupported (32 or 64)", head_dim);
    BVC_REQUIRE((size_t)B * N * 3 * H * head_dim * 2 < 0xffffffffull, "attn_fwd: qkv larger than 4 GiB");
    return head_dim == 64 ? fwd_hd<64>(qkv, ctx, lse, B, N, H, stream, sm_scale) : fwd_hd<32>(qkv, ctx, lse, B, N, H, stream, sm_scale);
}

int launch_attn_bwd(const bf16_t* qkv, const bf16_t* ctx, const bf16_t* dctx, const float* lse, float* delta,
                    bf16_t* dqkv, int B, int N, int H, int head_dim, hipStream_t stream, float sm_scale, int parts) {
    BVC_REQUIRE(B > 0 && N > 0 && H > 0, "attn_bwd: empty shape");
    BVC_REQUIRE(head_dim == 64 || head_dim == 32, "attn_bwd: head_dim %d unsupported (32 or 64)", head_dim);
    BVC_REQUIRE((size_t)B * N * 3 * H * head_dim * 2 < 0xffffffffull, "attn_bwd: qkv larger than 4 GiB");
    return head_dim == 64 ? bwd_hd<64>(qkv, ctx, dctx, lse, delta, dqkv, B, N, H, stream, sm_scale, parts)
                          : bwd_hd<32>(qkv, ctx, dctx, lse, delta, dqkv, B, N, H, stream, sm_scale, parts);
}

}  // namespace bvc
