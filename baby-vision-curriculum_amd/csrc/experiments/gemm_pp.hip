// "Ping-pong" persistent bf16 GEMM for gfx950 (round 4): ONE 512-thread workgroup per CU whose two wave rows (4 waves each, one wave
// of each row on every SIMD) take TURNS: while one row runs the K loop of its 128 x 256 output unit (fragment reads + MFMAs, nothing
// else), the other row runs the EPILOGUE of the unit it finished just before (convert, bias, GELU + GELU', 16-byte stores) and issues
// the LDS-DMA of the stream of K tiles for both - then they swap.  Why (profiles/r03_d_probe_step_b256.log, r04_b_*): on the step's
// short-K products (K = 384 ... 768, outputs as large as the inputs) gemm8's 256 x 256 tile spends as long in a unit's register
// epilogue as in its K loop, with the matrix pipes idle, because all eight waves reach the epilogue together and the 128 accumulator
// registers per wave leave no room to keep a finished unit aside (256 x 128 tiles with a second accumulator set were built and
// measured in round 4: the staggered-phase protocol serialises the two rows' non-MFMA segments, so moving the epilogue into them
// gains nothing).  Here a unit's epilogue has a whole K loop of the OTHER row to hide under, and the K-loop row's instruction stream
// is reads + MFMAs only.
//
//   * unit = 128 x 256 outputs, wave tile 128 x 64 (8 x 4 MFMA tiles of v_mfma_f32_16x16x32_bf16, 128 accumulator registers);
//     the workgroup walks its units (XCD-contiguous ids, as gemm8) as one stream; wave row (sequence index & 1) computes a unit;
//   * K tiles of 64: A image [128][64] + B image [256][64] (or [64][2 x 128] for a transposed B) = 48 KiB, THREE slots (144 KiB):
//     step g (one K tile, ONE workgroup barrier) reads slot g % 3 while the epilogue row fills slot (g + 2) % 3 - a tile has two
//     steps to land;
//   * the epilogue row issues the 12 LDS-DMA pieces per wave of a tile FIRST in its step (interleaved with the arithmetic of the
//     step's first slice), then its stores; every step carries the same number of stores (C = stores of the most slices a step can
//     hold; the rest are issued with a dropped offset), so the one counted wait per step, vmcnt(12 + 2 C), is exact: everything
//     older than the previous step's stores - i.e. the tile the K-loop row reads next - has landed.  A row that turns from epilogue
//     to K loop waits for the tile it issued last with vmcnt(C) at the end of its first K step;
//   * the 16 slices (16 rows x 32 columns per wave) of an epilogue are spread evenly over the steps of the other row's K loop.
// Results are bit-identical to gemm_kernel / gemm8_kernel for the same problem (same K order per output, same epilogue arithmetic).
//
// MEASURED (round 4, profiles/r04_d_pp_all.txt, r04_d_pp_dbg.txt) - NOT ADOPTED, kept in the experiments build with its evidence:
// every product of the step is 1.5 - 2 x SLOWER than on gemm8's 256 x 256 tile.  The ablations say why:
//   * the L2 -> LDS fill is the limit of the K loop on this chip: gemm8 moves 64 KiB per 256 x 256 x 64 K tile and is already at
//     ~10 TB/s chip-wide on the 4096^3 square (1.07 GB in 107 us); a 128 x 256 unit needs 48 KiB for half the work, i.e. 1.5 x the
//     bytes - with the LDS-DMA off the same launch takes 131 instead of 211 us;
//   * the GELU epilogue is bound by vector-ALU throughput, not by anything a second role can hide: ~300 cycles per pair of outputs,
//     64 pairs per wave and unit = 20 k cycles against 7 - 8 k for the other row's K loop, and the same 128 pairs per SIMD and
//     256 x 256 outputs as gemm8 spends with two waves sharing the ALU; the best case is max(ALU, fill) per unit, which for
//     K >= 768 is the fill;
//   * (fixable, not fixed) the K-loop row reads its A fragments only one 8-MFMA sub-phase ahead: LDS latency is exposed.
// What follows from it is cheaper epilogue arithmetic (common.h: gelu_and_grad2), not another schedule.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "../gemm_tile.h"

namespace bvc {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr uint32_t kInvalidBase = 0x80000000u;   // beyond every operand this kernel accepts (extents < 2 GiB)
constexpr uint32_t kDrop = 0xFFFFFFF0u;          // >= every descriptor's extent: the access is dropped
constexpr int UM = 128, UN = 256;
constexpr int A_BYTES = UM * 64 * 2, B_BYTES = UN * 64 * 2, SLOT = A_BYTES + B_BYTES, NSLOT = 3;
constexpr int BIAS_BYTES = 16384;
constexpr int NDMA = 12;                         // LDS-DMA instructions per epilogue-row wave and K tile (48 pieces of 1 KiB / 4 waves)

// vmcnt(base + mult * C) for the wave-uniform run-time C out of {1, 2, 3, 4, 6} (the count is an immediate)
template <int BASE, int MULT>
__device__ __forceinline__ void wait_vm(int C) {
    switch (C) {
        case 1: wait_vmcnt<BASE + MULT * 1>(); break;
        case 2: wait_vmcnt<BASE + MULT * 2>(); break;
        case 3: wait_vmcnt<BASE + MULT * 3>(); break;
        case 4: wait_vmcnt<BASE + MULT * 4>(); break;
        default: wait_vmcnt<BASE + MULT * 6>(); break;
    }
}

__device__ __forceinline__ bf16x8 tr_frag_pp(uint32_t region, uint32_t lane_base, uint32_t lane_swz, uint32_t chunk16, int ks) {
    const uint32_t a = region + lane_base + (chunk16 ^ lane_swz) + (uint32_t)ks * 8192u;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 bf16x4*)(size_t)a);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 bf16x4*)(size_t)(a + 1024u));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

}  // namespace

// BT: B stored [K][N] (NN products); OUTS: 16-byte stores per slice (1: BF16 / RELU, 2: GELU - gelu' and gelu)
template <bool BT, int OUTS>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const GemmGroup g, const int total_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = wave >> 2, wc = wave & 3;
    constexpr int TM = 8, TN = 4;

    // XCD x owns a contiguous run of unit ids; its gridDim.x / 8 workgroups take them round robin
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int xq = total_units >> 3, xr = total_units & 7;
    const int x_lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int x_hi = x_lo + xq + (xcd < xr ? 1 : 0);
    const int uid0 = x_lo + slot_id;
    if (uid0 >= x_hi) return;       // uniform per workgroup, before any barrier
    const int nu = (x_hi - uid0 + nslots - 1) / nslots;

    const GemmProblem& p = g.prob[0];
    const int nkt = p.K >> 6;
    const int tiles_n = (p.N + UN - 1) / UN, tiles_m = (p.M + UM - 1) / UM;
    const int cps = (16 + nkt - 1) / nkt;          // slices a step holds at most (nkt >= 6: 3, 2 or 1)
    const int C = cps * OUTS;                      // stores per step and wave

    auto unit_origin = [&](int uid, int& m0, int& n0) __attribute__((always_inline)) {
        int tm, tn;
        const int G = g.panel[0] > 0 ? g.panel[0] : tiles_n;
        tile_of(uid, tiles_m, tiles_n, G, tm, tn);
        m0 = __builtin_amdgcn_readfirstlane(tm * UM);
        n0 = __builtin_amdgcn_readfirstlane(tn * UN);
    };

    // the bias vector, zero padded to whole units, behind the three slots
    {
        AS3 float* lbias = (AS3 float*)((AS3 char*)smem + NSLOT * SLOT);
        const int npad = tiles_n * UN;
        for (int i = tid; i < npad; i += 512) lbias[i] = (p.bias && i < p.N) ? p.bias[i] : 0.f;
    }
    const float alpha = p.alpha_dev ? p.alpha * p.alpha_dev[0] : p.alpha;      // plain loads: before any LDS-DMA is in flight
    __syncthreads();

    // ------------------------------------------------------------------ the staging cursor (kept by every wave, used by the epilogue row)
    int s_uid = uid0, s_kt = 0, s_m0, s_n0;
    bool s_valid = true;
    unit_origin(uid0, s_m0, s_n0);
    const __amdgpu_buffer_rsrc_t s_ra = make_rsrc(p.A, p.a_bytes), s_rb = make_rsrc(p.B, p.b_bytes);
    uint32_t s_la, s_lb, s_baseA, s_baseB;
    {
        const int r = 8 * wc + (lane >> 3);           // sub-piece wc of a 64-row piece; sub-piece wc + 4 is 32 rows further (same swizzle)
        s_la = (uint32_t)((r * p.lda + (((lane & 7) ^ swz_rows(r)) << 3)) * 2);
        if constexpr (!BT) {
            s_lb = (uint32_t)((r * p.ldb + (((lane & 7) ^ swz_rows(r)) << 3)) * 2);
        } else {
            const int kr = 4 * wc + (lane >> 4);      // sub-piece wc of a 32-k-row piece; wc + 4 is 16 k rows further (same swizzle)
            s_lb = (uint32_t)((kr * p.ldb + (((lane & 15) ^ swz_tr<128>(kr)) << 3)) * 2);
        }
    }
    const uint32_t hiA = (uint32_t)(32 * p.lda * 2), pieceA = (uint32_t)(64 * p.lda * 2);
    const uint32_t hiB = BT ? (uint32_t)(16 * p.ldb * 2) : (uint32_t)(32 * p.ldb * 2);
    const uint32_t pieceB = BT ? (uint32_t)(32 * p.ldb * 2) : (uint32_t)(64 * p.ldb * 2);
    auto s_bases = [&]() __attribute__((always_inline)) {
        const int k0 = s_kt * 64;
        s_baseA = !s_valid ? kInvalidBase : (uint32_t)((s_m0 * p.lda + k0) * 2);
        s_baseB = !s_valid ? kInvalidBase : BT ? (uint32_t)((k0 * p.ldb + s_n0) * 2) : (uint32_t)((s_n0 * p.ldb + k0) * 2);
    };
    // the unit after the cursor's is decoded ahead of time (integer divisions), once per unit, outside the issue paths
    int n_m0 = 0, n_n0 = 0;
    bool n_valid = false, n_stale = true;
    auto s_prepare = [&]() __attribute__((always_inline)) {
        if (n_stale) {
            n_valid = s_uid + nslots < x_hi;
            if (n_valid) unit_origin(s_uid + nslots, n_m0, n_n0);
            n_stale = false;
        }
    };
    auto s_advance = [&]() __attribute__((always_inline)) {
        if (++s_kt == nkt) {
            s_kt = 0;
            s_uid += nslots;
            s_m0 = n_m0; s_n0 = n_n0; s_valid = n_valid;
            n_stale = true;
        }
        s_bases();
    };
    s_prepare();
    s_bases();
    const uint32_t lds_base = (uint32_t)(size_t)((AS3 char*)smem);
    // LDS-DMA d (0 .. 11) of the cursor's K tile into the slot at byte offset `so`: A pieces 0, 1 and B pieces 0 .. 3, two 1-KiB
    // sub-pieces (wc, wc + 4) of each per wave
    auto issue_dma = [&](int d, uint32_t so) __attribute__((always_inline)) {
        const int hi = d & 1;
        if (d < 4) {
            const int j = d >> 1;
            glds16(s_ra, s_la + s_baseA + (uint32_t)j * pieceA + (uint32_t)hi * hiA,
                   __builtin_amdgcn_readfirstlane(lds_base + so + (uint32_t)j * 8192u + (uint32_t)(wc + 4 * hi) * 1024u));
        } else {
            const int j = (d - 4) >> 1;
            const uint32_t off = BT ? (uint32_t)(j & 1) * pieceB + (uint32_t)(j >> 1) * 256u : (uint32_t)j * pieceB;
            glds16(s_rb, s_lb + s_baseB + off + (uint32_t)hi * hiB,
                   __builtin_amdgcn_readfirstlane(lds_base + so + (uint32_t)A_BYTES + (uint32_t)j * 8192u + (uint32_t)(wc + 4 * hi) * 1024u));
        }
    };

    // lane constants of the transposed fragment reads
    const uint32_t tr_base = (uint32_t)((8 * (lane >> 4) + ((lane >> 2) & 3)) * 256 + ((lane & 3) >> 1) * 16 + (lane & 1) * 8);
    const uint32_t tr_swz = (uint32_t)swz_tr<128>(8 * (lane >> 4) + ((lane >> 2) & 3)) << 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int my_m0 = 0, my_n0 = 0;          // the unit whose accumulators this wave holds
    bool have = false;

    // slot byte offsets of the step: read slot (g % 3), fill slot ((g + 2) % 3)
    uint32_t rd_off = 0, fill_off = 2u * SLOT;
    auto rotate = [&]() __attribute__((always_inline)) {
        rd_off = rd_off == 2u * SLOT ? 0u : rd_off + SLOT;
        fill_off = fill_off == 2u * SLOT ? 0u : fill_off + SLOT;
    };

    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, kDrop);
    const __amdgpu_buffer_rsrc_t rc2 = make_rsrc(p.C2 ? p.C2 : p.C, kDrop);
    const bool relu = p.epi == EPI_RELU;
    const int Mrows = p.M, Ncols = p.N, ldc = p.ldc;

    // ------------------------------------------------------------------ prologue: wave row 1 stages K tiles 0 and 1
    if (row == 1) {
#pragma nounroll
        for (int d = 0; d < NDMA; ++d) issue_dma(d, 0u);
        s_advance();
        s_prepare();
#pragma nounroll
        for (int d = 0; d < NDMA; ++d) issue_dma(d, (uint32_t)SLOT);
        s_advance();
        s_prepare();
#pragma nounroll
        for (int k = 0; k < C; ++k) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rc, kDrop, 0, 0);
        wait_vm<NDMA, 1>(C);          // K tile 0 has landed (tile 1 and the stand-in stores of "step -1" may be in flight)
    } else {
        s_advance();
        s_prepare();
        s_advance();
        s_prepare();
    }

    // ------------------------------------------------------------------ one epilogue period: 16 slices over the other row's nkt steps
    // solo: the workgroup's last unit - no partner, no barriers, no LDS-DMA: all slices in one "step"
    auto e_period = [&](const bool solo) __attribute__((always_inline)) {
        const AS3 float* lbias = (const AS3 float*)((AS3 char*)smem + NSLOT * SLOT);
        const int q4 = lane >> 4;
        auto slice = [&](auto c_) __attribute__((always_inline)) {
            constexpr int c = decltype(c_)::value;
            constexpr int i = c >> 1, jp = c & 1;
            uint32_t pk[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pk[0][k] = pk[1][k] = 0u;
            if (have) {
                const int nb = my_n0 + wc * 64 + 4 * q4;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const f32x4 bj = *reinterpret_cast<const AS3 f32x4*>(lbias + nb + 16 * (2 * jp + jj));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float v[2], act[2];
                        v[0] = acc[i][2 * jp + jj][2 * h] * alpha + bj[2 * h];
                        v[1] = acc[i][2 * jp + jj][2 * h + 1] * alpha + bj[2 * h + 1];
                        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
                        if constexpr (OUTS == 2) {
                            gelu_split(v, act);             // v <- gelu'(v), act <- gelu(v)
                            pk[1][2 * jj + h] = pack2bf(act[0], act[1]);
                        }
                        pk[0][2 * jj + h] = pack2bf(v[0], v[1]);
                    }
                }
            }
            // lanes l / l + 16 exchange halves: every store instruction writes 16 rows x 64 contiguous bytes
            // (the lane id goes through an empty asm: hipcc otherwise hoists the offsets and bound masks of all sixteen slices to the
            //  top of the period - sixteen registers and as many scalar pairs that are live across everything)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int m = my_m0 + 16 * i + (ln & 15);
            const int n = my_n0 + wc * 64 + 16 * ((ln >> 4) & 1) + 8 * (ln >> 5) + 32 * jp;
            const uint32_t o = (have && m < Mrows && n < Ncols) ? (uint32_t)(((size_t)m * ldc + n) * 2) : kDrop;
#pragma unroll
            for (int k = 0; k < OUTS; ++k) {
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk[k][0], pk[k][2], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk[k][1], pk[k][3], false, false);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, k == 0 ? rc : rc2, o, 0, 0);
            }
        };
        const int nsteps = solo ? 1 : nkt;
        int c = 0;
#pragma nounroll
        for (int stepi = 0; stepi < nsteps; ++stepi) {
            int nst = 0;
            if (!solo) {
                asm volatile("s_barrier" ::: "memory");
                if (!BVC_DBG(g, 256)) {
#pragma nounroll
                    for (int d = 0; d < NDMA; ++d) issue_dma(d, fill_off);   // this step's K tile first, then the stores
                }
                s_advance();
            }
            // slice c belongs to step floor(c nsteps / 16)
#pragma nounroll
            while (c < 16 && c * nsteps < 16 * (stepi + 1)) {
                switch (c) {
                    case 0: slice(std::integral_constant<int, 0>{}); break;
                    case 1: slice(std::integral_constant<int, 1>{}); break;
                    case 2: slice(std::integral_constant<int, 2>{}); break;
                    case 3: slice(std::integral_constant<int, 3>{}); break;
                    case 4: slice(std::integral_constant<int, 4>{}); break;
                    case 5: slice(std::integral_constant<int, 5>{}); break;
                    case 6: slice(std::integral_constant<int, 6>{}); break;
                    case 7: slice(std::integral_constant<int, 7>{}); break;
                    case 8: slice(std::integral_constant<int, 8>{}); break;
                    case 9: slice(std::integral_constant<int, 9>{}); break;
                    case 10: slice(std::integral_constant<int, 10>{}); break;
                    case 11: slice(std::integral_constant<int, 11>{}); break;
                    case 12: slice(std::integral_constant<int, 12>{}); break;
                    case 13: slice(std::integral_constant<int, 13>{}); break;
                    case 14: slice(std::integral_constant<int, 14>{}); break;
                    default: slice(std::integral_constant<int, 15>{}); break;
                }
                ++c;
                nst += OUTS;
            }
            if (!solo) {
#pragma nounroll
                for (int k = nst; k < C; ++k) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rc, kDrop, 0, 0);
                if (!BVC_DBG(g, 64)) wait_vm<NDMA, 2>(C);   // all but this step's pieces + stores and the previous step's stores: the next K tile has landed
                rotate();
                s_prepare();
            }
        }
    };

    // ------------------------------------------------------------------ one K-loop period
    auto k_period = [&](int uid) __attribute__((always_inline)) {
        unit_origin(uid, my_m0, my_n0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
        for (int kt = 0; kt < nkt; ++kt) {
            asm volatile("s_barrier" ::: "memory");
            const char* la = smem + rd_off;
            const char* lb = la + A_BYTES;
            // k-major sub-phases (k half, row quarter): 8 MFMAs each on 2 A fragments x the 4 B fragments of the k half; the A
            // fragments of the next sub-phase and (once) the B fragments of the second k half are read while the MFMAs of the
            // current one run (two register sets each).  Per accumulator the order is k half 0, then 1 - as in gemm_kernel / gemm8.
            bf16x8 bfr[2][TN], af[2][2];
            auto load_b = [&](auto ks_) __attribute__((always_inline)) {
                constexpr int ks = decltype(ks_)::value;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[ks][j] = BT ? tr_frag_pp(lds_base + rd_off + (uint32_t)A_BYTES + (uint32_t)(wc >> 1) * 16384u, tr_base, tr_swz,
                                                 (uint32_t)(2 * (((wc & 1) * 64) + 16 * j)), ks)
                                    : read_frag<256, false>(lb, wc * 64 + 16 * j, ks, lane);
            };
            auto load_a = [&](auto s_) __attribute__((always_inline)) {
                constexpr int sp = decltype(s_)::value, ks = sp >> 2, q = sp & 3;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) af[sp & 1][ii] = read_frag<256, false>(la, 16 * (2 * q + ii), ks, lane);
            };
            auto mm = [&](auto s_) __attribute__((always_inline)) {
                constexpr int sp = decltype(s_)::value, ks = sp >> 2, q = sp & 3;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[2 * q + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[sp & 1][ii], acc[2 * q + ii][j], 0, 0, 0);
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            load_b(I0{});
            load_a(I0{});
            __builtin_amdgcn_sched_barrier(0);
            auto sub = [&](auto s_) __attribute__((always_inline)) {
                constexpr int sp = decltype(s_)::value;
                if constexpr (sp < 7) load_a(std::integral_constant<int, sp + 1>{});
                if constexpr (sp == 1) load_b(I1{});
                __builtin_amdgcn_s_setprio(1);
                if (!BVC_DBG(g, 128)) mm(s_);
                else asm volatile("" :: "v"(af[sp & 1][0]), "v"(af[sp & 1][1]), "v"(bfr[sp >> 2][0]), "v"(bfr[sp >> 2][3]));
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            };
            sub(std::integral_constant<int, 0>{}); sub(std::integral_constant<int, 1>{});
            sub(std::integral_constant<int, 2>{}); sub(std::integral_constant<int, 3>{});
            sub(std::integral_constant<int, 4>{}); sub(std::integral_constant<int, 5>{});
            sub(std::integral_constant<int, 6>{}); sub(std::integral_constant<int, 7>{});
            if (kt == 0) wait_vm<0, 1>(C);     // the K tile this row issued in its last epilogue step (read two steps from here)
            s_advance();
            s_prepare();
            rotate();
        }
        have = true;
    };

    // periods 0 .. nu - 1: one row in its K loop, the other in the epilogue of its previous unit; period nu: the row that computed
    // the last unit finishes it alone
    for (int per = 0; per <= nu; ++per) {
        const bool last = per == nu;
        if (!last && (per & 1) == row) k_period(uid0 + per * nslots);
        else if (!last || ((nu - 1) & 1) == row) e_period(last);
    }
    wait_vmcnt<0>();
}

// ------------------------------------------------------------------ host side
template <bool BT, int OUTS>
static int launch_pp_one(const GemmGroup& g, int total, hipStream_t stream) {
    constexpr size_t lds = (size_t)NSLOT * SLOT + BIAS_BYTES;
    static_assert(lds <= 160 * 1024, "LDS per CU");
    if (dry_run().on) {
        snprintf(dry_run().name, sizeof(dry_run().name), "bvc::gemm_pp_kernel<%s, %d>", BT ? "true" : "false", OUTS);
        return BVC_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<BT, OUTS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        BVC_CHECK_HIP(hipGetDevice(&dev));
        BVC_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
        if (ncu < 8) ncu = 8;
    }
    const int grid = total < ncu ? ((total + 7) / 8) * 8 : ncu;      // one workgroup per CU, a multiple of the 8 XCDs
    hipLaunchKernelGGL((gemm_pp_kernel<BT, OUTS>), dim3(grid), dim3(512), lds, stream, g, total);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

// Launcher hook used by launch_gemm (gemm.hip, tile config 14).  Returns BVC_OK after launching, 1 when the problem is not eligible:
// one NT / NN problem, bf16 outputs without side inputs (BF16 / GELU / RELU), K a multiple of 64 with at least six K tiles per unit,
// N (rounded up to 256) * 4 bytes of bias in 16 KiB of LDS, operands below 2 GiB.
int launch_gemm_pp(const GemmGroup& g, GemmLayout layout, hipStream_t stream) {
    if (g.nprob != 1 || layout == GEMM_TN) return 1;
    const GemmProblem& p = g.prob[0];
    if (p.epi != EPI_BF16 && p.epi != EPI_GELU && p.epi != EPI_RELU) return 1;
    if (p.split_k != 1 || p.K % 64 != 0 || p.K < 384 || p.rowsum) return 1;
    if (p.a_bytes >= kInvalidBase || p.b_bytes >= kInvalidBase) return 1;
    if ((size_t)((p.N + UN - 1) / UN) * UN * 4 > (size_t)BIAS_BYTES) return 1;
    const int total = g.tile_start[1];
    if (total <= 0) return 1;
    const bool two = p.epi == EPI_GELU;
    if (layout == GEMM_NT) return two ? launch_pp_one<false, 2>(g, total, stream) : launch_pp_one<false, 1>(g, total, stream);
    return two ? launch_pp_one<true, 2>(g, total, stream) : launch_pp_one<true, 1>(g, total, stream);
}

}  // namespace bvc
