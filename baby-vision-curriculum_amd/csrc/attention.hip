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
template <int HD, int NW = 4>
__device__ __forceinline__ void stage64(__amdgpu_buffer_rsrc_t rs, int row0, int ld, int col0, char* lds,
                                        int wave, int lane) {
    constexpr int CPR = HD / 8;                 // 16-B chunks per row
    constexpr int RPP = 64 / CPR;               // rows per 1 KiB piece
#pragma unroll
    for (int jj = 0; jj < (HD / 8 + NW - 1) / NW; ++jj) {
        const int j = wave + NW * jj;
        if (NW > HD / 8 && j >= HD / 8) break;      // (more waves than pieces: wave-uniform)
        const int r = RPP * j + lane / CPR;
        const int c = (lane % CPR) ^ swz_dual<HD>(r);
        const uint32_t off = (uint32_t)(((size_t)(row0 + r) * ld + col0 + c * 8) * 2);
        // (readfirstlane: the LDS address is wave-uniform by construction - it has to sit in a scalar register for M0 - but hipcc cannot
        //  always prove it once the caller's loop structure gets involved)
        glds16(rs, off, __builtin_amdgcn_readfirstlane((uint32_t)(size_t)((AS3 char*)lds) + (uint32_t)j * 1024u));     // asm: see common.h (no compiler-made drain)
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

// The ragged LAST block of a sequence (round 4).  A block is four waves of 32 rows; N = 1568 leaves 32 rows for the 13th block, N = 160
// for the second: three of its waves had no rows, ran the whole loop on clamped rows and threw the result away - the block cost as
// much as a full one (1 / 13 of the decoder's attention time for 1 / 49 of its rows).  When at most two waves of the last block own
// rows, the idle waves now SHARE the loop of the owners: with `valid` owning waves (1 or 2) and gs = 4 / valid waves per 32-row tile,
// wave w works on tile w % valid and takes the 32-row sub-tiles of the other sequence dimension whose index is congruent to
// w / valid modulo gs; the partial results of a tile (running max / sum and O for the forward, plain sums for the gradients) meet in
// the LDS of the staging ring once the loop is over, and part 0 of each tile stores.  Barriers are untouched (every wave walks the
// same loop); blocks with three or four owning waves run as before (gs = 1).
struct TailSplit { int gs, own, part, valid; };
template <int NW = 4>
__device__ __forceinline__ TailSplit tail_split(int N, int tile, int wave) {
    if constexpr (NW != 4) { TailSplit t; t.valid = NW; t.gs = 1; t.own = wave; t.part = 0; return t; }     // (experiment: 8-wave blocks, no split)
    const int rows = N - tile * 128;
    TailSplit t;
    t.valid = rows >= 128 ? 4 : (rows + 31) >> 5;
    t.gs = t.valid == 1 ? 4 : t.valid == 2 ? 2 : 1;
    t.own = t.gs == 1 ? wave : wave & (t.valid - 1);
    t.part = t.gs == 1 ? 0 : wave >> (t.valid - 1);
    return t;
}
// partial accumulators of parts 1 .. gs-1 -> part 0 through LDS (`buf`: free staging memory, NT f32x16 tiles per wave); plain sums
template <int NT>
__device__ __forceinline__ void tail_reduce(char* buf, const TailSplit& ts, int lane, f32x16 (&acc)[NT]) {
    AS3 float* cl = (AS3 float*)buf;
    constexpr int SLOT = NT * 16 * 64;
    __syncthreads();
    if (ts.part > 0) {
        AS3 float* w = cl + ((ts.part - 1) * ts.valid + ts.own) * SLOT + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) w[(t * 16 + r) * 64] = acc[t][r];
    }
    __syncthreads();
    if (ts.part == 0) {
        for (int p = 1; p < ts.gs; ++p) {
            const AS3 float* w = cl + ((p - 1) * ts.valid + ts.own) * SLOT + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += w[(t * 16 + r) * 64];
        }
    }
}

// grid ceil(N/128) * B*H (1-D, see attn_block); 256 threads; wave w owns queries q0 + 32 w .. + 31
template <int HD, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                          float* __restrict__ lse, int N, int H, int D,
                                                          uint32_t qkv_bytes, float scale_log2, int remap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (K image + V image)
    constexpr int IMG = 64 * HD * 2, STG = 2 * IMG, SUB = 32 * HD * 2;
    const AS3 char* lds = (const AS3 char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, bh;
    attn_block((N + 32 * NW - 1) / (32 * NW), remap, tile_, bh);
    const int b = bh / H, head = bh % H;
    const int ld = 3 * D;
    const TailSplit ts = tail_split<NW>(N, tile_, wave);
    const int qi = tile_ * (32 * NW) + ts.own * 32 + (lane & 31);   // this lane's query
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, qkv_bytes);
    const FragAddr<HD> fa = make_frag_addr<HD>(lane);
    const int gm = ts.gs - 1;
    auto mine = [&](int sub) { return (sub & gm) == ts.part; };      // does this wave take 32-key sub-tile `sub`?

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
        stage64<HD, NW>(rs, krow0 + kt * 64, ld, D + head * HD, smem + stage * STG, wave, lane);
        stage64<HD, NW>(rs, krow0 + kt * 64, ld, 2 * D + head * HD, smem + stage * STG + IMG, wave, lane);
    };
#pragma unroll
    for (int stq = 0; stq < HD / 16; ++stq) settle(qf[stq]);
    issue(0, 0);
    for (int kt = 0; kt < nkt; kt += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) issue(kt + 1, 1);
        if (mine(2 * kt)) fwd_subtile<HD, 0, IMG>(lds, fa, qf, st, kt * 64, N, h, scale_log2);
        if (kt * 64 + 32 < N && mine(2 * kt + 1)) fwd_subtile<HD, SUB, IMG + SUB>(lds, fa, qf, st, kt * 64 + 32, N, h, scale_log2);
        if (kt + 1 >= nkt) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < nkt) issue(kt + 2, 0);
        if (mine(2 * kt + 2)) fwd_subtile<HD, STG, STG + IMG>(lds, fa, qf, st, kt * 64 + 64, N, h, scale_log2);
        if (kt * 64 + 96 < N && mine(2 * kt + 3)) fwd_subtile<HD, STG + SUB, STG + IMG + SUB>(lds, fa, qf, st, kt * 64 + 96, N, h, scale_log2);
    }
    if (ts.gs > 1) {      // workgroup-uniform: merge the parts of a query tile (flash-decoding style: common maximum, rescaled sums)
        AS3 float* cl = (AS3 float*)smem;
        constexpr int NO = HD / 32 * 16, SLOT = (NO + 2) * 64;
        __syncthreads();
        if (ts.part > 0) {
            AS3 float* w = cl + ((ts.part - 1) * ts.valid + ts.own) * SLOT + lane;
#pragma unroll
            for (int t = 0; t < HD / 32; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) w[(t * 16 + r) * 64] = st.o[t][r];
            w[NO * 64] = st.m_run;
            w[(NO + 1) * 64] = st.l_run;
        }
        __syncthreads();
        if (ts.part == 0) {
            for (int p = 1; p < ts.gs; ++p) {
                const AS3 float* w = cl + ((p - 1) * ts.valid + ts.own) * SLOT + lane;
                const float m_p = w[NO * 64], l_p = w[(NO + 1) * 64];
                const float m_new = fmaxf(st.m_run, m_p);
                const float a = st.m_run == -INFINITY ? 0.f : fast_exp2(st.m_run - m_new);     // (a part that saw no key: -inf, weight 0)
                const float c = m_p == -INFINITY ? 0.f : fast_exp2(m_p - m_new);
                st.m_run = m_new;
                st.l_run = st.l_run * a + l_p * c;
#pragma unroll
                for (int t = 0; t < HD / 32; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st.o[t][r] = st.o[t][r] * a + w[(t * 16 + r) * 64] * c;
            }
        }
    }
    const float l_tot = xhalf_sum(st.l_run);
    const float inv = 1.f / l_tot;
    if (qi < N && ts.part == 0) {
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
    const TailSplit ts = tail_split(N, tile_, wave);      // the ragged last block: idle waves share the key loop of the owners
    const int qi = tile_ * 128 + ts.own * 32 + (lane & 31);
    const int qc = min(qi, N - 1);
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, qkv_bytes);
    const FragAddr<HD> fa = make_frag_addr<HD>(lane);
    const int gm = ts.gs - 1;
    auto mine = [&](int sub) { return (sub & gm) == ts.part; };

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
    if (h == 0 && qi < N && ts.part == 0) delta[(size_t)bh * N + qi] = -del_q;     // NEGATED: the dK/dV kernel, launched after this one, starts its dP chain from it
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
        if (mine(2 * kt)) dq_subtile<HD, 0, IMG>(lds, fa, qf, dof, dq, kt * 64, N, h, scale_log2, nlse, ndel);
        if (kt * 64 + 32 < N && mine(2 * kt + 1)) dq_subtile<HD, SUB, IMG + SUB>(lds, fa, qf, dof, dq, kt * 64 + 32, N, h, scale_log2, nlse, ndel);
        if (kt + 1 >= nkt) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < nkt) issue(kt + 2, 0);
        if (mine(2 * kt + 2)) dq_subtile<HD, STG, STG + IMG>(lds, fa, qf, dof, dq, kt * 64 + 64, N, h, scale_log2, nlse, ndel);
        if (kt * 64 + 96 < N && mine(2 * kt + 3))
            dq_subtile<HD, STG + SUB, STG + IMG + SUB>(lds, fa, qf, dof, dq, kt * 64 + 96, N, h, scale_log2, nlse, ndel);
    }
    if (ts.gs > 1) tail_reduce<HD / 32>(smem, ts, lane, dq);
    if (qi < N && ts.part == 0) {
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

// (the single-wave 96-query dQ experiment of round 2 - 3-4 % at the decoder shape, not adopted, profiles/r02_h_attention_single_wave.txt -
//  lives in experiments/attention_dq_wave96.inc and is compiled only into a -DBVC_EXPERIMENTS build)
#ifdef BVC_EXPERIMENTS
#include "experiments/attention_dq_wave96.inc"
#endif

// ============================================================================ dK, dV
// LDS stage = Q image | dO image | lse 256 B | -delta 256 B
template <int HD, int QOFF, int STAT>
__device__ __forceinline__ void dkdv_subtile(const AS3 char* lds, const FragAddr<HD>& fa, const bf16x8 (&kf)[HD / 16],
                                             const bf16x8 (&vf)[HD / 16], f32x16 (&dk)[HD / 32], f32x16 (&dv)[HD / 32], int q0,
                                             int N, int h, float scale_log2) {
    constexpr int IMG = 64 * HD * 2, KS = 16 * HD * 2;
    constexpr int DOFF = QOFF + IMG;
#ifndef BVC_ATTN_STATS_LDS
    // Rows of s / dp are queries, so the row statistics (lse, -delta) are the same for every lane: broadcast reads.  Until round 4 they
    // came as eight 16-byte LDS reads per sub-tile - a third of the LDS bytes of a kernel that is bound by LDS bandwidth.  Now they
    // enter through the MFMA pipe, which has the slack: one more k-step per chain whose A operand (query on the row) carries
    // [-lse / c (hi, lo), -delta (hi, lo), 0 ...] as bf16 hi + lo pairs (16 mantissa bits: 2^-17 relative, 5e-5 on P) and whose B operand is
    // a per-lane CONSTANT (ones in the two k rows of the chain's statistic).  Two 4-byte LDS reads, ~12 vector instructions and two MFMAs
    // replace eight 16-byte reads and sixteen multiply-adds: S' = Q K^T - lse / c, then P = exp2(c S'); dP' = dO V^T - delta.
    bf16x8 sfrag, one_s, one_d;
    {
        const AS3 float* st = reinterpret_cast<const AS3 float*>(lds + STAT);
        const int r = threadIdx.x & 31;
        const float xl = -st[r] * (1.0f / scale_log2), xd = st[64 + r];        // (the dQ kernel stores delta negated)
        const uint32_t hh = pack2bf(xl, xd);                                     // the two high parts: lse' in bits 0-15, delta' in 16-31
        const float lhi = __uint_as_float(hh << 16), dhi = __uint_as_float(hh & 0xffff0000u);
        const uint32_t ll = pack2bf(xl - lhi, xd - dhi);                         // the two low parts
        union { bf16x8 v; uint32_t u[4]; } f, a, b;
        f.u[0] = h ? 0u : ((hh & 0xffffu) | (ll << 16));                         // k rows 0, 1: lse' hi, lo
        f.u[1] = h ? 0u : ((hh >> 16) | (ll & 0xffff0000u));                     // k rows 2, 3: delta' hi, lo
        f.u[2] = 0u; f.u[3] = 0u;
        a.u[0] = h ? 0u : 0x3f803f80u; a.u[1] = 0u; a.u[2] = 0u; a.u[3] = 0u;      // bf16 1.0 twice: k rows 0, 1
        b.u[0] = 0u; b.u[1] = h ? 0u : 0x3f803f80u; b.u[2] = 0u; b.u[3] = 0u;      // k rows 2, 3
        sfrag = f.v; one_s = a.v; one_d = b.v;
    }
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int stp = 0; stp < HD / 16; ++stp) {
        s = MFMA32(lds_rows<QOFF>(lds, fa.rows[stp]), kf[stp], s);
        dp = MFMA32(lds_rows<DOFF>(lds, fa.rows[stp]), vf[stp], dp);
    }
    s = MFMA32(sfrag, one_s, s);
    dp = MFMA32(sfrag, one_d, dp);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[r] * scale_log2);
        s[r] = p;
        dp[r] = p * dp[r];
    }
#else
    // (the round-2 form, kept for the A/B: rows of s/dp are queries: registers 4g..4g+3 <-> queries q0 + 8 g + 4 h + 0..3, lse / -delta
    //  broadcast from LDS; the dP chain starts from -delta: four 16-byte LDS reads ARE the initial accumulator)
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
#endif
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
    const TailSplit ts = tail_split(N, tile_, wave);      // the ragged last key block: idle waves share the query loop of the owners
    const int ki = tile_ * 128 + ts.own * 32 + (lane & 31);   // this lane's key
    const int kc = min(ki, N - 1);
    const int h = lane >> 5;
    const int gm = ts.gs - 1;
    auto mine = [&](int sub) { return (sub & gm) == ts.part; };
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
        if (mine(2 * qt)) dkdv_subtile<HD, 0, 2 * IMG>(lds, fa, kf, vf, dk, dv, qt * 64, N, h, scale_log2);
        if (qt * 64 + 32 < N && mine(2 * qt + 1)) dkdv_subtile<HD, SUB, 2 * IMG + 128>(lds, fa, kf, vf, dk, dv, qt * 64 + 32, N, h, scale_log2);
        if (qt + 1 >= nqt) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 2 < nqt) issue(qt + 2, 0);
        if (mine(2 * qt + 2)) dkdv_subtile<HD, STG, STG + 2 * IMG>(lds, fa, kf, vf, dk, dv, qt * 64 + 64, N, h, scale_log2);
        if (qt * 64 + 96 < N && mine(2 * qt + 3)) dkdv_subtile<HD, STG + SUB, STG + 2 * IMG + 128>(lds, fa, kf, vf, dk, dv, qt * 64 + 96, N, h, scale_log2);
    }
    if (ts.gs > 1) {      // (one accumulator set at a time: three partial sets of 8 KiB fit the 33 KiB ring, six do not)
        tail_reduce<HD / 32>(smem, ts, lane, dk);
        tail_reduce<HD / 32>(smem, ts, lane, dv);
    }
    if (ki < N && ts.part == 0) {
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

// ============================================================================ whole-head backward for short sequences (round 4)
// N <= 160 (the encoder's visible tokens; JEPA's context / prediction sets): Q, K, V and dO of one (clip, head) fit in LDS together
// (4 images of up to 192 rows x 64 = 96 KiB), so ONE workgroup does the whole backward of the head with S, P, dP and dS computed
// ONCE - the five products of the algorithm instead of the seven the two-kernel form executes (its dQ kernel recomputes S and dP) -
// with no second pass over HBM, no atomics and the same deterministic sums:
//   phase 0  LDS-DMA of the four images; lse and -delta = -rowsum(dO * O) of every query into LDS;
//   phase 1  key on the lane (as attn_bwd_dkdv_kernel): wave w of FIVE owns key block w; per query block S = Q K^T, dP = dO V^T - delta,
//            P = exp2(S c - lse), dS = P dP;  dV^T += dO^T P, dK^T += Q^T dS in registers; dS^T leaves as bf16 into a [32 k][32 q]
//            LDS block per (key block, query block) - the one tile that crosses LDS (the dQ product sums over the lane index of dS);
//   phase 2  query on the lane (as attn_bwd_dq_kernel): wave w owns query block w: dQ^T += K^T dS^T with BOTH operands read
//            transposed from LDS (K image, dS^T blocks).
// 146 KiB of LDS: one workgroup per CU.  At N = 160 the two-kernel form runs 2 x 128-row blocks per head at 62 % fill.
// NBT: number of 32-row blocks when known at compile time (5: the encoder's 160 tokens - both block loops unroll and hipcc overlaps one
// block's LDS reads and MFMA chains with the previous block's softmax arithmetic), 0: taken from N at run time
template <int HD, int NBT>
__global__ __launch_bounds__(320, 1) void attn_bwd_head_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                               const bf16_t* __restrict__ ctx, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, int N, int H, int D, int nheads,
                                                               uint32_t qkv_bytes, uint32_t dctx_bytes, float scale, float scale_log2) {
    static_assert(HD == 64, "whole-head backward: head_dim 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IMG = 192 * HD * 2, SUB = 32 * HD * 2, KS = 16 * HD * 2;      // image of 3 x 64 rows; 32-row block; 16-row k-step (tr reads)
    constexpr int QI = 0, KI = IMG, VI = 2 * IMG, OI = 3 * IMG;                   // Q | K | V | dO
    constexpr int STAT = 4 * IMG;                                                 // lse[192] | -delta[192]
    constexpr int DST = STAT + 2 * 192 * 4;                                       // dS^T blocks [kb][qb] of 2 KiB: [32 k][32 q] bf16
    const AS3 char* lds = (const AS3 char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = 3 * D, h = lane >> 5;
    const int NB = NBT > 0 ? NBT : (N + 31) >> 5;                 // 32-row blocks (<= 5)
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qkv, qkv_bytes), rd = make_rsrc(dctx, dctx_bytes);
    const FragAddr<HD> fa = make_frag_addr<HD>(lane);
    const FragAddr<32> fs = make_frag_addr<32>(lane);     // the dS^T blocks are 64-byte-row images

    // PERSISTENT over (clip, head) pairs: with 146 KiB of LDS only one workgroup is resident per CU, and a fresh workgroup per head
    // would pay its dispatch, its image loads and its statistics in the open.  Inside the loop the NEXT head's Q, V, dO images and
    // statistics are fetched while phase 2 runs (which reads K and dS^T only), its K image right after phase 2.
    // FIVE waves (N <= 160 = five 32-row blocks: one key block and one query block per wave; SIMD 0 hosts two waves, which overlap
    // each other's latencies); the images are staged by the first four (stage64 deals 1-KiB pieces to four waves).
    auto stage_qvo = [&](int bh) __attribute__((always_inline)) {
        if (wave < 4) {
            const int b = bh / H, head = bh % H, row0 = b * N;
            for (int rb = 0; rb * 64 < N; ++rb) {
                stage64<HD>(rq, row0 + rb * 64, ld, head * HD, smem + QI + rb * 2 * SUB, wave, lane);
                stage64<HD>(rq, row0 + rb * 64, ld, 2 * D + head * HD, smem + VI + rb * 2 * SUB, wave, lane);
                stage64<HD>(rd, row0 + rb * 64, D, head * HD, smem + OI + rb * 2 * SUB, wave, lane);
            }
        }
    };
    auto stage_k = [&](int bh) __attribute__((always_inline)) {
        if (wave < 4) {
            const int b = bh / H, head = bh % H, row0 = b * N;
            for (int rb = 0; rb * 64 < N; ++rb) stage64<HD>(rq, row0 + rb * 64, ld, D + head * HD, smem + KI + rb * 2 * SUB, wave, lane);
        }
    };
    // statistics of head `bh` into registers: lse of query `tid`, and this thread's share of -delta = -rowsum(dO * O) for the rows
    // tid >> 3, + 40, ... (8 lanes per row, 16 bytes of dO and O each, folded by three shuffles)
    float st_lse = 0.f, st_del[5];
    auto load_stats = [&](int bh) __attribute__((always_inline)) {
        const int b = bh / H, head = bh % H;
        st_lse = (tid < N) ? lse[(size_t)bh * N + tid] : 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int q = (tid >> 3) + 40 * k, c = tid & 7;
            float acc = 0.f;
            if (q < N) {
                const bf16x8 dv = load8(dctx + (size_t)(b * N + q) * D + head * HD + 8 * c);
                const bf16x8 ov = load8(ctx + (size_t)(b * N + q) * D + head * HD + 8 * c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += bf2f((bf16_t)dv[j]) * bf2f((bf16_t)ov[j]);
            }
            st_del[k] = acc;
        }
    };
    auto store_stats = [&]() __attribute__((always_inline)) {
        AS3 float* st = (AS3 float*)((AS3 char*)smem + STAT);
        if (tid < 192) st[tid] = st_lse;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float acc = st_del[k];
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            const int q = (tid >> 3) + 40 * k;
            if ((tid & 7) == 0 && q < 192) st[192 + q] = -acc;
        }
    };

    int bh = blockIdx.x;
    if (bh >= nheads) return;
    load_stats(bh);
    stage_qvo(bh);
    stage_k(bh);
    for (; bh < nheads; bh += gridDim.x) {
    const int b = bh / H, head = bh % H;
    const int nxt = bh + gridDim.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this head's images (and the previous head's stores)
    store_stats();
    __syncthreads();

    // ---- phase 1: dK, dV of this wave's key blocks; dS^T to LDS
    for (int kb = wave; kb < NB; kb += 5) {
        const int ki = kb * 32 + (lane & 31);
        bf16x8 kf[HD / 16], vf[HD / 16];
#pragma unroll
        for (int st = 0; st < HD / 16; ++st) {
            kf[st] = *reinterpret_cast<const AS3 bf16x8*>(lds + KI + kb * SUB + fa.rows[st]);
            vf[st] = *reinterpret_cast<const AS3 bf16x8*>(lds + VI + kb * SUB + fa.rows[st]);
        }
        f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
        for (int t = 0; t < HD / 32; ++t) { dk[t] = zero16(); dv[t] = zero16(); }
        const bool key_ok = ki < N;
#pragma unroll
        for (int qb = 0; qb < (NBT > 0 ? NBT : NB); ++qb) {
            const AS3 char* qim = lds + QI + qb * SUB;
            const AS3 char* oim = lds + OI + qb * SUB;
            const AS3 float* stl = reinterpret_cast<const AS3 float*>(lds + STAT) + qb * 32 + 4 * h;
            f32x16 s = zero16(), dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 nd = *reinterpret_cast<const AS3 f32x4*>(stl + 192 + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) dp[4 * g + e] = nd[e];
            }
#pragma unroll
            for (int stp = 0; stp < HD / 16; ++stp) {
                s = MFMA32(*reinterpret_cast<const AS3 bf16x8*>(qim + fa.rows[stp]), kf[stp], s);
                dp = MFMA32(*reinterpret_cast<const AS3 bf16x8*>(oim + fa.rows[stp]), vf[stp], dp);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ls = *reinterpret_cast<const AS3 f32x4*>(stl + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pr = fast_exp2(fmaf(s[r], scale_log2, -ls[e]));
                    s[r] = pr;
                    dp[r] = pr * dp[r];
                }
            }
            if (qb * 32 + 32 > N || !__all(key_ok)) {      // ragged last query block / last key block: zero what lies outside (selects, not products)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!key_ok || qb * 32 + acc_row(r, h) >= N) { s[r] = 0.f; dp[r] = 0.f; }
            }
            const bf16x8 p0 = acc_to_frag(s, 0), p1 = acc_to_frag(s, 1);
            const bf16x8 d0 = acc_to_frag(dp, 0), d1 = acc_to_frag(dp, 1);
            // dS^T[k = this lane's key][q = 8 g + 4 h + 0 .. 3]: 8 bytes per (lane, g) into the block's swizzled 64-byte rows
            {
                AS3 char* blk = (AS3 char*)smem + DST + (kb * 5 + qb) * 2048 + (lane & 31) * 64 + 8 * h;
                const int sw = swz_dual<32>(lane & 31);
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                union { bf16x8 v; u32x2 u[2]; } w0, w1;
                w0.v = d0; w1.v = d1;
                *reinterpret_cast<AS3 u32x2*>(blk + ((0 ^ sw) << 4)) = w0.u[0];
                *reinterpret_cast<AS3 u32x2*>(blk + ((1 ^ sw) << 4)) = w0.u[1];
                *reinterpret_cast<AS3 u32x2*>(blk + ((2 ^ sw) << 4)) = w1.u[0];
                *reinterpret_cast<AS3 u32x2*>(blk + ((3 ^ sw) << 4)) = w1.u[1];
            }
#pragma unroll
            for (int t = 0; t < HD / 32; ++t) {
                dv[t] = MFMA32(lds_tr<0>(oim, fa.tr[t][0], fa.tr[t][1]), p0, dv[t]);
                dk[t] = MFMA32(lds_tr<0>(qim, fa.tr[t][0], fa.tr[t][1]), d0, dk[t]);
            }
#pragma unroll
            for (int t = 0; t < HD / 32; ++t) {
                dv[t] = MFMA32(lds_tr<KS>(oim, fa.tr[t][0], fa.tr[t][1]), p1, dv[t]);
                dk[t] = MFMA32(lds_tr<KS>(qim, fa.tr[t][0], fa.tr[t][1]), d1, dk[t]);
            }
        }
        if (key_ok) {
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
    __syncthreads();
    if (nxt < nheads) {       // Q, V, dO and the statistics are dead: the next head's arrive under phase 2
        stage_qvo(nxt);
        load_stats(nxt);
    }

    // ---- phase 2: dQ of this wave's query blocks
    for (int qb = wave; qb < NB; qb += 5) {
        const int qi = qb * 32 + (lane & 31);
        f32x16 dq[HD / 32];
#pragma unroll
        for (int t = 0; t < HD / 32; ++t) dq[t] = zero16();
#pragma unroll
        for (int kb = 0; kb < (NBT > 0 ? NBT : NB); ++kb) {
            const AS3 char* kim = lds + KI + kb * SUB;
            const AS3 char* blk = lds + DST + (kb * 5 + qb) * 2048;
            const bf16x8 b0 = lds_tr<0>(blk, fs.tr[0][0], fs.tr[0][1]);
            const bf16x8 b1 = lds_tr<1024>(blk, fs.tr[0][0], fs.tr[0][1]);
#pragma unroll
            for (int t = 0; t < HD / 32; ++t) dq[t] = MFMA32(lds_tr<0>(kim, fa.tr[t][0], fa.tr[t][1]), b0, dq[t]);
#pragma unroll
            for (int t = 0; t < HD / 32; ++t) dq[t] = MFMA32(lds_tr<KS>(kim, fa.tr[t][0], fa.tr[t][1]), b1, dq[t]);
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
    __syncthreads();
    if (nxt < nheads) stage_k(nxt);          // K was read by phase 2
    }
}

// ============================================================================ host launchers
// BVC_ATTN_PLAIN_GRID=1 switches the XCD-aware block map off (same-run A/B in tools/ab/microbench.py; read per launch)
static int xcd_remap() {
#ifdef BVC_EXPERIMENTS      // same-process A/B of the XCD-aware block map (tools/ab/microbench.py)
    return getenv("BVC_ATTN_PLAIN_GRID") == nullptr;
#else
    return 1;
#endif
}

// the whole-head backward for N <= 160 can be switched off for same-process A/Bs (experiments build: BVC_ATTN_NO_HEAD_KERNEL=1)
static bool head_kernel_enabled() {
#ifdef BVC_EXPERIMENTS
    return getenv("BVC_ATTN_NO_HEAD_KERNEL") == nullptr;
#else
    return true;
#endif
}

template <int HD>
static int fwd_hd(const bf16_t* qkv, bf16_t* ctx, float* lse, int B, int N, int H, hipStream_t stream, float sm_scale) {
    const int D = H * HD;
    const size_t bytes = (size_t)B * N * 3 * D * 2;
    const float scale_log2 = (sm_scale > 0.f ? sm_scale : 1.0f / sqrtf((float)HD)) * 1.4426950408889634f;
#ifdef BVC_EXPERIMENTS
    if (getenv("BVC_ATTN_NW8") != nullptr) {      // 256-query blocks (eight waves): half the K / V staging per query - same-process A/B only
        const dim3 grid8((unsigned)(((N + 255) / 256) * B * H));
        hipLaunchKernelGGL((attn_fwd_kernel<HD, 8>), grid8, dim3(512), 4 * 64 * HD * 2, stream, qkv, ctx, lse, N, H, D, (uint32_t)bytes, scale_log2,
                           xcd_remap());
        BVC_CHECK_HIP(hipGetLastError());
        return BVC_OK;
    }
#endif
    const dim3 grid((unsigned)(((N + 127) / 128) * B * H));
    hipLaunchKernelGGL((attn_fwd_kernel<HD, 4>), grid, dim3(256), 4 * 64 * HD * 2, stream, qkv, ctx, lse, N, H, D, (uint32_t)bytes, scale_log2,
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
    if constexpr (HD == 64) {
        // short sequences: the whole backward of a (clip, head) in one workgroup (attn_bwd_head_kernel); `delta` stays untouched
        if (parts == 3 && N <= 160 && head_kernel_enabled()) {
            constexpr size_t lds = 4 * 192 * HD * 2 + 2 * 192 * 4 + 25 * 2048;
            static bool attr_set = false;
            if (!attr_set) {
                BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_head_kernel<HD, 5>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_head_kernel<HD, 0>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_set = true;
            }
            static int ncu = 0;
            if (ncu == 0) {
                int dev = 0;
                hipDeviceProp_t prop;
                BVC_CHECK_HIP(hipGetDevice(&dev));
                BVC_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
                ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            }
            const int nheads = B * H;
            const dim3 g((unsigned)(nheads < ncu ? nheads : ncu));
            if (N > 128)
                hipLaunchKernelGGL((attn_bwd_head_kernel<HD, 5>), g, dim3(320), lds, stream, qkv, dctx, ctx, lse, dqkv, N, H, D, nheads,
                                   (uint32_t)bytes, (uint32_t)((size_t)B * N * D * 2), scale, scale_log2);
            else
                hipLaunchKernelGGL((attn_bwd_head_kernel<HD, 0>), g, dim3(320), lds, stream, qkv, dctx, ctx, lse, dqkv, N, H, D, nheads,
                                   (uint32_t)bytes, (uint32_t)((size_t)B * N * D * 2), scale, scale_log2);
            BVC_CHECK_HIP(hipGetLastError());
            return BVC_OK;
        }
    }
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
