// Persistent 128x128 / 128x64 GEMM for the short-K products of the step (K = 384 ... 3072, tens of tile rounds per launch).
//
// Why (tools/ab/gemm_dbg.py, profiles/r01_e_gemm_decomposition_b64.txt): in gemm_kernel every tile is its own workgroup, and at
// K = 384 about half of a workgroup's life is spent outside the K loop - dispatch, argument fetch, the first LDS-DMA round
// trip to HBM (nothing to overlap it with inside the workgroup) and the epilogue: 72 of 149 us for the decoder's qkv product
// with the stores taken out.  Here a workgroup stays resident and walks a sequence of tiles as ONE stream of K steps: the
// last two K steps of a tile refill their LDS slots with the first two K steps of the NEXT tile, so the next tile's operands
// arrive under the current tile's last MFMAs and its epilogue; the dispatch / setup cost is paid once per workgroup.
//
// Differences from gemm_kernel that make this possible:
//   * the epilogue does not borrow the staging slots (they are receiving the next tile): the accumulators are parked 16 rows
//     at a time in a wave-private 4 KiB region (LDS per workgroup 64 + 16 = 80 KiB, two workgroups still share a CU);
//   * loads, stores and LDS-DMA retire through one in-order counter (vmcnt), so every side input of the tile (bias, residual,
//     GELU' argument) and the prefetched next-tile stages are drained by ONE vmcnt(0) BEFORE the first store of the epilogue;
//     the first K step of the next tile then needs no VMEM wait at all, only the barrier.
// Scope: one problem, NT / NN operand layouts, no split-K, the epilogues of the transformer layers (BF16, GELU, RESID, DGELU,
// F32, F32_BF16).  Everything else stays on gemm_kernel.  Same tile walk, same swizzles, same fragment order: results are
// bit-identical to gemm_kernel's.
#include <stdlib.h>

#include <stdio.h>

#include "gemm_tile.h"

namespace bvc {


typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N>
__device__ __forceinline__ void wait_vmcnt_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// counted wait with a run-time count out of the few values the deferred-store schedule produces (multiples of 4 up to 24)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: wait_vmcnt_n<0>(); break;
        case 4: wait_vmcnt_n<4>(); break;
        case 8: wait_vmcnt_n<8>(); break;
        case 12: wait_vmcnt_n<12>(); break;
        case 16: wait_vmcnt_n<16>(); break;
        case 20: wait_vmcnt_n<20>(); break;
        default: wait_vmcnt_n<24>(); break;
    }
}

// DEFER (128x128 tiles, bf16-output epilogues): the finished tile is NOT stored by its epilogue; its packed bf16 chunks stay
// in registers and are stored during the first two K steps of the NEXT tile, after that step's refill loads.  vmcnt is one
// in-order counter, so a store issued before a load delays every wait for that load; issued after it, and with the EXACT
// number of younger operations written into the counted wait, the stores drain in the background for two K steps before
// any wait reaches them.  The count is exact because every chunk is stored through a buffer descriptor (rows / columns past
// the matrix get an out-of-range offset: dropped by the hardware, still counted).
// DEFER: 0 = stores in the epilogue; 1 = one bf16 output held (BF16, DGELU); 2 = two (GELU: pre and act)
template <int BN, bool BT, int DEFER = 0>
__global__ __launch_bounds__(256, 2) void gemm_persist_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    constexpr int WM = 64, WN = BN / 2, TM = 4, TN = WN / 16;
    constexpr int DMA_PER_STAGE = BM / 32 + BN / 32;
    constexpr int UNITS = WN / 4;                 // 16-B units per parked row
    constexpr int CPR = WN / 8;                   // 8-column chunks per row
    constexpr int RPU = 64 / CPR;                 // rows covered by the 64 lanes in one pass
    constexpr int U = 16 / RPU;                   // passes per 16-row round (2 for 128-wide tiles, 1 for 64-wide)
    constexpr int NSIDE = TM * U;

    const GemmProblem& p = g.prob[0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_m * tiles_n;
    // XCD x owns a contiguous run of tile ids (as in gemm_kernel); its gridDim.x / 8 resident workgroups take them round robin
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int xq = total >> 3, xr = total & 7;
    const int x_lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int x_hi = x_lo + xq + (xcd < xr ? 1 : 0);
    int lid = x_lo + slot_id;
    if (lid >= x_hi) return;       // uniform per workgroup: no barrier has been executed yet

    const int G = g.panel[0];
    const int nt = p.K / BK;       // host guarantees K % 64 == 0 and nt >= 2
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
    AS3 char* wl = (AS3 char*)smem + 2 * STAGE + wave * (16 * WN * 4);     // wave-private epilogue parking: 16 rows x WN f32
    const int epi = p.epi;
    const float alpha = p.alpha_dev ? p.alpha * p.alpha_dev[0] : p.alpha;
    // deferred stores (DEFER): packed chunks of the previous tile and where they go
    constexpr int NST = 4;                               // chunks stored per K step and per output tensor
    constexpr bool two_out = DEFER == 2;
    constexpr int nst = DEFER * NST;
    const uint32_t c_bytes = (uint32_t)((size_t)p.M * p.ldc * 2);
    const __amdgpu_buffer_rsrc_t rc0 = make_rsrc(p.C, c_bytes);
    const __amdgpu_buffer_rsrc_t rc1 = make_rsrc(two_out ? p.C2 : p.C, c_bytes);
    u32x4 held0[DEFER ? NSIDE : 1], held1[DEFER == 2 ? NSIDE : 1];
    bool pending = false;
    int m0p = 0, n0p = 0;

    int tm, tn;
    tile_of(lid, tiles_m, tiles_n, G, tm, tn);
    int m0 = tm * BM, n0 = tn * BN;
    stage_tile<BM, false>(ra, m0, 0, p.lda, smem, wave, lane);
    stage_tile<BN, BT>(rb, n0, 0, p.ldb, smem + A_BYTES, wave, lane);
    stage_tile<BM, false>(ra, m0, BK, p.lda, smem + STAGE, wave, lane);
    stage_tile<BN, BT>(rb, n0, BK, p.ldb, smem + STAGE + A_BYTES, wave, lane);
    wait_vmcnt<DMA_PER_STAGE>();
    asm volatile("s_barrier" ::: "memory");
    int gs = 0;                    // K steps consumed so far: its parity is the LDS slot of the current step
    bool fresh = false;            // first K step after an epilogue: the stages it needs were drained before the stores

    while (true) {
        const int lid_next = lid + nslots;
        const bool has_next = lid_next < x_hi;
        int m0n = 0, n0n = 0;
        if (has_next) {
            int tmn, tnn;
            tile_of(lid_next, tiles_m, tiles_n, G, tmn, tnn);
            m0n = tmn * BM; n0n = tnn * BN;
        }
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int it = 0; it < nt; ++it, ++gs) {
            char* slot = smem + (gs & 1) * STAGE;
            bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[ks][i] = read_frag<BM, false>(slot, wm * WM + 16 * i, ks, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[ks][j] = read_frag<BN, BT>(slot + A_BYTES, wn * WN + 16 * j, ks, lane);
            }
            auto mfma_half = [&](int ks) {
                if (BVC_DBG(g, 64)) __builtin_amdgcn_s_setprio(1);      // experiment: raised priority over the MFMA burst
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
                if (BVC_DBG(g, 64)) __builtin_amdgcn_s_setprio(0);
            };
            mfma_half(0);
            // this slot is refilled with K step it + 2 of the stream: of this tile, or of the next one
            const bool in_tile = it + 2 < nt;
            const bool refill = in_tile || has_next;
            if (refill) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const int mm = in_tile ? m0 : m0n, nn = in_tile ? n0 : n0n;
                const int kk = (in_tile ? it + 2 : it + 2 - nt) * BK;
                stage_tile<BM, false>(ra, mm, kk, p.lda, slot, wave, lane);
                stage_tile<BN, BT>(rb, nn, kk, p.ldb, slot + A_BYTES, wave, lane);
            }
            if (DEFER && pending && it < 2) {
                // half of the previous tile's chunks, AFTER this step's refill loads (see the kernel comment)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < NST; ++q) {
                    const int c = it * NST + q;                  // NSIDE == 2 * NST
                    const int m = m0p + wm * WM + 16 * (c / U) + (((c % U) * 64 + lane) / CPR);
                    const int n = n0p + wn * WN + (lane % CPR) * 8;
                    const uint32_t off = (m < p.M && n < p.N) ? (uint32_t)(((size_t)m * p.ldc + n) * 2) : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(it == 0 ? held0[q] : held0[NST + q], rc0, off, 0, 0);
                    if constexpr (two_out) __builtin_amdgcn_raw_buffer_store_b128(it == 0 ? held1[q] : held1[NST + q], rc1, off, 0, 0);
                }
                asm volatile("" ::: "memory");
            }
            mfma_half(1);
            if (it + 1 < nt || has_next) {     // the stream continues: K step + 1 must have landed everywhere
                if (!fresh) {
                    if (DEFER) {
                        // operations younger than the loads of K step + 1: this step's refill, and the deferred stores issued
                        // after those loads (both halves at it == 1, the second half at it == 2)
                        const int stores = pending ? (it == 1 ? 2 * nst : it == 2 ? nst : 0) : 0;
                        wait_vmcnt_dyn((refill ? DMA_PER_STAGE : 0) + stores);
                    } else {
                        if (refill) wait_vmcnt<DMA_PER_STAGE>(); else wait_vmcnt<0>();
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            fresh = false;
        }
        pending = false;

        // ------------------------------------------------------------------ epilogue of this tile
        // lane -> (row, 8-column chunk) of each 16-row round: pass u covers rows (64 u + lane) / CPR, columns 8 (lane % CPR) ..
        const int cc = lane % CPR;
        const int n = n0 + wn * WN + cc * 8;
        const bool ncol_ok = n < p.N;
        f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && ncol_ok) {
            bias0 = *reinterpret_cast<const f32x4*>(p.bias + n);
            bias1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
        }
        f32x4 side0[NSIDE], side1[DEFER ? 1 : NSIDE];       // the f32 addend (residual) exists only where stores are not deferred
        const bool side_f32 = DEFER == 0 && epi == EPI_RESID, side_aux = DEFER != 2 && epi == EPI_DGELU;
        if (side_f32 || side_aux) {
#pragma unroll
            for (int c = 0; c < NSIDE; ++c) {
                const int m = m0 + wm * WM + 16 * (c / U) + (((c % U) * 64 + lane) / CPR);
                side0[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (DEFER == 0) side1[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m >= p.M || !ncol_ok) continue;
                if (side_aux) {
                    side0[c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const bf16_t*>(p.aux) + (size_t)m * p.ldaux + n);
                } else {
                    if constexpr (DEFER == 0) {
                        const float* src = p.resid + (size_t)m * p.ldc + n;
                        side0[c] = *reinterpret_cast<const f32x4*>(src);
                        side1[c] = *reinterpret_cast<const f32x4*>(src + 4);
                    }
                }
            }
        }
        // one drain for everything issued so far: the side inputs AND the next tile's two prefetched stages (in-order vmcnt:
        // after the stores below, any wait on a younger load would also wait for those stores' write acknowledgements)
        wait_vmcnt<0>();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            {
                const int row = lane & 15;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int unit = (4 * j + (lane >> 4)) ^ (row & (UNITS - 1));
                    *reinterpret_cast<AS3 f32x4*>(wl + row * (WN * 4) + unit * 16) = acc[i][j];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = (u * 64 + lane) / CPR;
                const int m = m0 + wm * WM + 16 * i + row;
                const f32x4 lo = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc) ^ (row & (UNITS - 1))) << 4));
                const f32x4 hi = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc + 1) ^ (row & (UNITS - 1))) << 4));
                if (m >= p.M || !ncol_ok) continue;
                const int c = U * i + u;
                float v[8] = {lo[0] * alpha + bias0[0], lo[1] * alpha + bias0[1], lo[2] * alpha + bias0[2], lo[3] * alpha + bias0[3],
                              hi[0] * alpha + bias1[0], hi[1] * alpha + bias1[1], hi[2] * alpha + bias1[2], hi[3] * alpha + bias1[3]};
                const size_t idx = (size_t)m * p.ldc + n;
                auto store_f32 = [&](float* dst) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                };
                const bool hold = DEFER && has_next;       // uniform: the chunk waits in registers for the next tile's first K steps
                auto store_bf16 = [&](void* base, const float* w, bool second) {
                    const u32x4 pk = {pack2bf(w[0], w[1]), pack2bf(w[2], w[3]), pack2bf(w[4], w[5]), pack2bf(w[6], w[7])};
                    if (hold) {
                        if constexpr (DEFER == 2) { if (second) held1[c] = pk; else held0[c] = pk; }
                        else if constexpr (DEFER == 1) held0[c] = pk;
                    } else {
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(base) + idx) = pk;
                    }
                };
                switch (epi) {
                    case EPI_F32: store_f32(reinterpret_cast<float*>(p.C) + idx); break;
                    case EPI_BF16: store_bf16(p.C, v, false); break;
                    case EPI_GELU: {
                        float a[8];
                        gelu_split(v, a);          // v <- gelu'(pre), a <- gelu(pre)
                        store_bf16(p.C, v, false);
                        store_bf16(p.C2, a, true);
                    } break;
                    case EPI_RESID: {
                        if constexpr (DEFER == 0) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += side0[c][e]; v[4 + e] += side1[c][e]; }
                            store_f32(reinterpret_cast<float*>(p.C) + idx);
                        }
                    } break;
                    case EPI_DGELU: {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t w = __float_as_uint(side0[c][e]);      // aux = gelu'(pre), saved by the forward epilogue
                            v[2 * e] *= __uint_as_float(w << 16);
                            v[2 * e + 1] *= __uint_as_float(w & 0xffff0000u);
                        }
                        store_bf16(p.C, v, false);
                    } break;
                    case EPI_F32_BF16: {
                        store_f32(reinterpret_cast<float*>(p.C) + idx);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C2) + idx) =
                            u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
                    } break;
                    default: break;
                }
            }
        }
        if (!has_next) break;
        // every wave drained its own share of the next tile's stages before its stores; the barrier makes that true of all four
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fresh = true;
        if (DEFER) { pending = true; m0p = m0; n0p = n0; }
        lid = lid_next; m0 = m0n; n0 = n0n;
    }
}

// Launcher hook used by launch_gemm (gemm.hip).  cfg 0 = 128x128 tiles, 1 = 128x64.  Returns BVC_OK after launching, or 1 when
// the problem is not eligible.
template <int BN, bool BT, int DEFER = 0>
static int launch_persist_one(const GemmGroup& g, hipStream_t stream) {
    constexpr size_t lds = 2 * (size_t)(128 + BN) * 64 * 2 + 4 * 16 * (BN / 2) * 4;     // stages + parking: 80 KiB / 56 KiB
    if (dry_run().on) {
        snprintf(dry_run().name, sizeof(dry_run().name), "bvc::gemm_persist_kernel<%d, %s, %d>", BN, BT ? "true" : "false", DEFER);
        return BVC_OK;
    }
    static bool attr_set = false;
    if (lds > 65536 && !attr_set) {
        BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_persist_kernel<BN, BT, DEFER>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_persist_kernel<BN, BT, DEFER>), dim3(512), dim3(256), lds, stream, g);    // 256 CUs x 2 resident workgroups
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

int launch_gemm_persist(const GemmGroup& g, GemmLayout layout, int cfg, hipStream_t stream, int defer) {
    const GemmProblem& p = g.prob[0];
    const int bn = cfg == 0 ? 128 : 64;
    const int tiles = ((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
    const int epi = p.epi;
    const bool epi_ok = epi == EPI_F32 || epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_DGELU || epi == EPI_F32_BF16;
    if (!(layout == GEMM_NT || layout == GEMM_NN) || cfg > 1 || p.split_k != 1 || !epi_ok || p.K % 64 != 0 || p.K < 128 || g.panel[0] <= 0)
        return 1;
    static const int min_tiles = BVC_EXP_ENV("BVC_PERSIST_MIN_TILES") ? atoi(BVC_EXP_ENV("BVC_PERSIST_MIN_TILES")) : 2 * 512;
    if (tiles < min_tiles) return 1;             // fewer than two rounds of the resident workgroups: little to chain
    // deferred stores: 128x128 tiles, epilogues whose outputs are bf16, output below 4 GiB (buffer-descriptor stores)
    const bool defer_ok = cfg == 0 && (epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_DGELU) && (size_t)p.M * p.ldc * 2 < 0xFFFFFFF0ull;
    if (defer == 2 && !defer_ok) return 1;
    if (defer && defer_ok) {
        if (epi == EPI_GELU)
            return layout == GEMM_NT ? launch_persist_one<128, false, 2>(g, stream) : launch_persist_one<128, true, 2>(g, stream);
        return layout == GEMM_NT ? launch_persist_one<128, false, 1>(g, stream) : launch_persist_one<128, true, 1>(g, stream);
    }
    if (cfg == 0) return layout == GEMM_NT ? launch_persist_one<128, false>(g, stream) : launch_persist_one<128, true>(g, stream);
    return layout == GEMM_NT ? launch_persist_one<64, false>(g, stream) : launch_persist_one<64, true>(g, stream);
}

}  // namespace bvc
