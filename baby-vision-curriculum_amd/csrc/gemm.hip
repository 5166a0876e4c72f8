// bf16 x bf16 -> f32 MFMA GEMM for gfx950 (MI355X), with the fused epilogues the ViT step needs: the per-tile kernel
// (every output tile is one workgroup) and the host-side launcher / tile planner.  gemm_persist.hip holds the persistent form
// that chains tiles for the short-K products; gemm_tile.h the staging / fragment helpers both share.
//
// Structure (per 256-thread workgroup = 4 waves as 2x2, 64-deep K steps):
//   * A and B tiles go HBM -> LDS directly with `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB per
//     wave-instruction).  The buffer descriptor's bounds check returns zeros for rows past the end of
//     the allocation, which is how ragged M / ragged contraction lengths are handled - no host padding.
//   * two LDS slots, "early refill": a wave pulls all fragments of the current K step into registers, a barrier proves
//     the slot drained, the slot is refilled with K step t+2 and only then do the MFMAs run - two K steps of DMA are in
//     flight under the MFMAs; waits are counted (`s_waitcnt vmcnt(N)`) and barriers raw, so the DMA flies across them.
//   * LDS images are XOR-swizzled on the 16-byte chunk index.  LDS-DMA writes lane-linear, so the
//     swizzle is applied to the per-lane SOURCE address and again on the read (both sides or neither).
//   * k-contiguous operands are read with ds_read_b128; operands whose contraction index is the
//     strided one (dX = dY W, dW = dY^T X) stay in their natural layout in HBM and LDS and are
//     transposed on the LDS read by ds_read_b64_tr_b16 - no transposed copies of weights or activations.
//   * v_mfma_f32_16x16x32_bf16 with the operands swapped, so each lane ends up with 4 consecutive
//     output columns of one row; the epilogue parks the f32 tile in LDS and re-reads it row-major so that every global
//     load / store is a full 128-B line per 8 lanes, and fetches all its side inputs before its first store (loads and
//     stores retire through one in-order counter).
//   * block index -> tile mapping is XCD-aware (blocks b and b+8 share an XCD/L2): each XCD gets a contiguous run of
//     tiles, walked in column panels sized for its L2 (NT / NN) or along the short side with K-splits fastest (TN).
//   * up to 4 independent problems per launch (grouped GEMM) to fill 256 CUs with the small
//     weight-gradient products of one transformer layer; split-K with f32 atomics; fused bias gradients.
#include <stdio.h>
#include <stdlib.h>

#include "gemm_tile.h"

namespace bvc {

Options& options() {
    static Options o;
    return o;
}
DryRun& dry_run() {
    static thread_local DryRun d;
    return d;
}

// ------------------------------------------------------------------ the kernel
// Two LDS slots; waits use a COUNTED vmcnt and raw s_barrier so that the refill DMA keeps flying across barriers
// (a __syncthreads() would drain it).  NS is kept as a template parameter for the launcher's LDS sizing only (= 2).
template <int BM, int BN, bool AT, bool BT, int NS, int LB = 2, bool EARLY_ = true>
__global__ __launch_bounds__(256, LB) void gemm_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int DMA_PER_STAGE = BM / 32 + BN / 32;   // LDS-DMA instructions per wave per stage

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    if (BVC_DBG(g, 2) && ((blockIdx.x >> 3) & 1)) __builtin_amdgcn_s_sleep(100);

    // XCD-aware remap (bijective for any grid size): XCD x owns a contiguous run of logical ids
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
    int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);

    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if (i < g.nprob && lid >= g.tile_start[i]) pi = i;
    const GemmProblem& p = g.prob[pi];
    lid -= g.tile_start[pi];
    // Tile walk inside one problem.  An XCD runs a contiguous run of ids, ~64 of them at a time (32 CUs x 2 workgroups), and
    // whatever those 64 workgroups share must fit its 4 MiB L2:
    //   * K-splits are the SLOWEST index: the workgroups resident together then belong to one split and differ in (m, n), so
    //     they share operand slabs (with the split fastest, 4 of every 4 neighbours shared nothing);
    //   * columns are walked in panels of `panel` tiles, rows fastest-but-one: 64 neighbours form a rows x panel block whose
    //     B panel stays in L2 while the rows stream past it.  Measured before this walk (profiles/r01_d_traffic_b64_dispatches.txt):
    //     encoder fc1 at B=64 fetched 259 MB for 20 MB of operands - its 4.7 MB weight cycled through L2 once per row group.
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int split, tm, tn;
    const int G = g.panel[pi];
    if (G > 0) {
        split = lid / ntiles;
        const int t = lid - split * ntiles;
        const int full = (tiles_n / G) * G * tiles_m;          // tiles in the full-width panels
        if (t < full) {
            const int pn = t / (G * tiles_m), w = t - pn * G * tiles_m;
            tm = w / G; tn = pn * G + (w - tm * G);
        } else {
            const int r = tiles_n % G, w = t - full;
            tm = w / r; tn = (tiles_n - r) + (w - tm * r);
        }
    } else {   // legacy walk: split fastest, then along the shorter side of the tile grid
        split = lid % p.split_k;
        const int tl = lid / p.split_k;
        const bool m_fast = tiles_n > tiles_m;
        tm = m_fast ? tl % tiles_m : tl / tiles_n;
        tn = m_fast ? tl / tiles_m : tl % tiles_n;
    }
    const int tile = tm * tiles_n + tn;     // id for the per-tile loss partials (independent of the walk)
    const int m0 = tm * BM, n0 = tn * BN;

    const int nt_all = (p.K + BK - 1) / BK;
    const int per = (nt_all + p.split_k - 1) / p.split_k;
    const int t0 = split * per;
    const int t1 = min(nt_all, t0 + per);
    const int nt = t1 - t0;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias gradient fused into the weight-gradient product: db[m] = sum_k A(m,k) is one more MFMA
    // column against an all-ones operand, computed by the workgroups of the first column tile only
    const bool do_rowsum = AT && p.rowsum != nullptr && n0 == 0 && wn == 0;
    f32x4 accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

    // Two K-loop variants, A/B'd in ONE run on one box (profiles/r01_d_kloop_ab.txt; box-to-box variance on the pool is far
    // larger than the effect): the early-refill loop below wins for NN (5-15 %) and TN (~8 %), ties for NT at K <= 768 and wins
    // at long K (enc fc2 22 vs 27 us).  The plain double buffer is kept reachable (stages = 4) for such comparisons.
    constexpr bool EARLY = EARLY_;
    if constexpr (!EARLY) {
        if (nt > 0) {
            stage_tile<BM, AT>(ra, m0, t0 * BK, p.lda, smem, wave, lane);
            stage_tile<BN, BT>(rb, n0, t0 * BK, p.ldb, smem + A_BYTES, wave, lane);
        }
        for (int it = 0; it < nt; ++it) {
            // every wave drains its own DMA, then the barrier publishes the tile and proves the other slot is drained
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (it + 1 < nt) {
                char* nxt = smem + ((it + 1) & 1) * STAGE;
                stage_tile<BM, AT>(ra, m0, (t0 + it + 1) * BK, p.lda, nxt, wave, lane);
                stage_tile<BN, BT>(rb, n0, (t0 + it + 1) * BK, p.ldb, nxt + A_BYTES, wave, lane);
            }
            const char* la = smem + (it & 1) * STAGE;
            const char* lb = la + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, AT>(la, wm * WM + 16 * i, ks, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = read_frag<BN, BT>(lb, wn * WN + 16 * j, ks, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                if (do_rowsum) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accb[i], 0, 0, 0);
                }
            }
        }
    } else {
        // K loop, "early refill": every wave first pulls ALL its fragments of the current K-step into registers, a barrier
        // proves the slot is drained, the slot is refilled by LDS-DMA at once (K-step t+2) and only then do the MFMAs run -
        // from registers.  Two K-steps of DMA are in flight during the MFMAs with just two 32 KiB slots (the plain double
        // buffer had one, and rocprofv3 showed ~50 % of wave cycles waiting on it), so 2 workgroups still fit per CU.
        if (nt > 0) {
            stage_tile<BM, AT>(ra, m0, t0 * BK, p.lda, smem, wave, lane);
            stage_tile<BN, BT>(rb, n0, t0 * BK, p.ldb, smem + A_BYTES, wave, lane);
            if (nt > 1) {
                stage_tile<BM, AT>(ra, m0, (t0 + 1) * BK, p.lda, smem + STAGE, wave, lane);
                stage_tile<BN, BT>(rb, n0, (t0 + 1) * BK, p.ldb, smem + STAGE + A_BYTES, wave, lane);
                wait_vmcnt<DMA_PER_STAGE>();
            } else {
                wait_vmcnt<0>();
            }
            asm volatile("s_barrier" ::: "memory");
        }
        for (int it = 0; it < nt; ++it) {
            char* slot = smem + (it & 1) * STAGE;
            bf16x8 af[2][TM], bfr[2][TN];
    #pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
    #pragma unroll
                for (int i = 0; i < TM; ++i) af[ks][i] = read_frag<BM, AT>(slot, wm * WM + 16 * i, ks, lane);
    #pragma unroll
                for (int j = 0; j < TN; ++j) bfr[ks][j] = read_frag<BN, BT>(slot + A_BYTES, wn * WN + 16 * j, ks, lane);
            }
            auto mfma_half = [&](int ks) {
                if (BVC_DBG(g, 16)) return;     // experiment: loads and barriers only
    #pragma unroll
                for (int i = 0; i < TM; ++i)
    #pragma unroll
                    for (int j = 0; j < TN; ++j)
                        // operands swapped: D = Bfrag^T-view x Afrag gives lane (l&15) = m, regs = 4 consecutive n
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
                if (do_rowsum) {
    #pragma unroll
                    for (int i = 0; i < TM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[ks][i], accb[i], 0, 0, 0);
                }
            };
            mfma_half(0);   // needs only the first half's fragments: the second half's LDS reads retire underneath
            if (it + 2 < nt) {
                // own fragment reads retired, then the barrier: every wave is done with this slot -> refill it
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (!BVC_DBG(g, 32)) stage_tile<BM, AT>(ra, m0, (t0 + it + 2) * BK, p.lda, slot, wave, lane);
                if (!BVC_DBG(g, 8 | 32)) stage_tile<BN, BT>(rb, n0, (t0 + it + 2) * BK, p.ldb, slot + A_BYTES, wave, lane);
            }
            mfma_half(1);
            if (it + 1 < nt) {
                // K-step it+1 must have landed everywhere before the next iteration reads it; the refill just issued may fly on
                if (it + 2 < nt && !BVC_DBG(g, 8 | 32)) wait_vmcnt<DMA_PER_STAGE>(); else wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS is reused by the epilogue
    // ------------------------------------------------------------------ epilogue
    // The MFMA fragment layout gives each lane 4 columns of one row: stored directly, a wave touches 16 rows x 32 B
    // per instruction and the write path reaches only ~2.2 TB/s (measured).  Instead every wave parks its f32 tile in
    // LDS (free after the K loop; XOR-swizzled 16-B units, no bank conflicts) and re-reads it row-major, so each lane
    // owns 8 consecutive columns and every global load / store of the epilogue is a full 128-B line per 8 lanes.
    const int epi = p.epi;
    const float alpha = p.alpha_dev ? p.alpha * p.alpha_dev[0] : p.alpha;
    float sumsq = 0.f, possum = 0.f;
    const bool atomic = p.split_k > 1;
    constexpr int UNITS = WN / 4;                       // 16-B units per tile row
    AS3 char* wl = (AS3 char*)smem + wave * (WM * WN * 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = 16 * i + (lane & 15);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int unit = (4 * j + (lane >> 4)) ^ (row & (UNITS - 1));
            *reinterpret_cast<AS3 f32x4*>(wl + row * (WN * 4) + unit * 16) = acc[i][j];
        }
    }
    if (atomic) {
        // split-K: f32 atomics straight into C.  Float atomics run at full rate only when a wave-instruction covers whole
        // contiguous rows (256 B = one row of a 64-wide wave tile, or two 128-B rows of a 32-wide one), so the lanes walk
        // the parked tile one dword each, row by row, instead of the 8-column chunks of the store path.
        if (nt > 0) {
            float* cbase = reinterpret_cast<float*>(p.C);
#pragma unroll 4
            for (int idx = lane; idx < WM * WN; idx += 64) {
                const int row = idx / WN, col = idx % WN;
                const int m = m0 + wm * WM + row, n = n0 + wn * WN + col;
                const int unit = (col >> 2) ^ (row & (UNITS - 1));
                float v = *reinterpret_cast<const AS3 float*>(wl + row * (WN * 4) + unit * 16 + (col & 3) * 4) * alpha;
                if (m < p.M && n < p.N) {
                    if (p.bias && split == 0) v += p.bias[n];
                    atomicAdd(cbase + (size_t)m * p.ldc + n, v);
                }
            }
        }
    } else if (nt > 0 || !atomic) {
        constexpr int CPR = WN / 8;                     // 8-column chunks per row
        constexpr int NCH = WM * WN / 8 / 64;           // chunks per lane
        // Chunk `it` of a lane is row it * (64 / CPR) + lane / CPR, columns 8 (lane % CPR) .. +7: the columns never change.
        const int cc = lane % CPR, rsub = lane / CPR;
        const int n = n0 + wn * WN + cc * 8;
        const bool ncol_ok = n < p.N;
        // On gfx950 loads and stores retire through ONE in-order counter (vmcnt): a load issued after a store cannot be
        // waited for without waiting for that store's write acknowledgement too.  The epilogue used to alternate
        // "load side input, compute, store" per chunk and so paid one store round trip per chunk - measured as ~5 us of fixed
        // cost per tile (tools/ab/gemm_dbg.py: decoder fc1 340 us, 201 us with the stores dropped).  All side inputs of the tile
        // (bias, residual, GELU' argument, labels, positional rows) are therefore fetched FIRST, into registers the parked
        // accumulators no longer need, and the stores follow back to back.
        f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && (!atomic || split == 0) && ncol_ok) {
            bias0 = *reinterpret_cast<const f32x4*>(p.bias + n);
            bias1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
        }
        f32x4 side0[NCH], side1[NCH];     // f32 addend (residual / positional row / labels) or the 4 dwords of the bf16 aux row
        const bool side_f32 = (epi == EPI_RESID && !atomic) || epi == EPI_POS || epi == EPI_E2D || epi == EPI_LOSS;
        const bool side_aux = epi == EPI_DGELU || epi == EPI_DRELU;
        if (side_f32 || side_aux) {
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                const int m = m0 + wm * WM + it * (64 / CPR) + rsub;
                side0[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                side1[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m >= p.M || !ncol_ok) continue;
                if (side_aux) {
                    side0[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const bf16_t*>(p.aux) + (size_t)m * p.ldaux + n);
                } else {
                    const float* src = epi == EPI_RESID ? p.resid + (size_t)m * p.ldc + n
                                     : epi == EPI_LOSS  ? p.labels + (size_t)m * p.ldc + n
                                                        : p.pos + (size_t)p.rowtok[m] * p.N + n;
                    side0[it] = *reinterpret_cast<const f32x4*>(src);
                    side1[it] = *reinterpret_cast<const f32x4*>(src + 4);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int row = it * (64 / CPR) + rsub;
            const int m = m0 + wm * WM + row;
            const f32x4 lo = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc) ^ (row & (UNITS - 1))) << 4));
            const f32x4 hi = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc + 1) ^ (row & (UNITS - 1))) << 4));
            if (m >= p.M || !ncol_ok) continue;
            float v[8] = {lo[0] * alpha + bias0[0], lo[1] * alpha + bias0[1], lo[2] * alpha + bias0[2], lo[3] * alpha + bias0[3],
                          hi[0] * alpha + bias1[0], hi[1] * alpha + bias1[1], hi[2] * alpha + bias1[2], hi[3] * alpha + bias1[3]};
            const size_t idx = (size_t)m * p.ldc + n;
            auto store_f32 = [&](float* dst) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
            };
            auto store_bf16 = [&](void* base, size_t at, const float* w) {
                if (BVC_DBG(g, 1)) return;
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(base) + at) =
                    uint4{pack2bf(w[0], w[1]), pack2bf(w[2], w[3]), pack2bf(w[4], w[5]), pack2bf(w[6], w[7])};
            };
            auto add_side = [&]() {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += side0[it][e]; v[4 + e] += side1[it][e]; }
            };
            switch (epi) {
                case EPI_F32: {
                    float* c = reinterpret_cast<float*>(p.C) + idx;
                    if (atomic) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) atomicAdd(c + e, v[e]);
                    } else {
                        store_f32(c);
                    }
                } break;
                case EPI_BF16: store_bf16(p.C, idx, v); break;
                case EPI_GELU: {
                    float a[8];
                    gelu_split(v, a);              // v <- gelu'(pre), a <- gelu(pre)
                    store_bf16(p.C, idx, v);
                    store_bf16(p.C2, idx, a);
                } break;
                case EPI_RESID: {
                    float* c = reinterpret_cast<float*>(p.C) + idx;
                    if (atomic) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) atomicAdd(c + e, v[e]);
                    } else {
                        add_side();
                        store_f32(c);
                    }
                } break;
                case EPI_POS: {
                    add_side();
                    store_f32(reinterpret_cast<float*>(p.C) + idx);
                } break;
                case EPI_E2D: {
                    add_side();
                    const size_t orow = (size_t)(m / p.rin) * p.rout + (m % p.rin);
                    store_f32(reinterpret_cast<float*>(p.C) + orow * p.ldc + n);
                } break;
                case EPI_LOSS: {
                    if (p.C2) store_f32(reinterpret_cast<float*>(p.C2) + idx);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] -= side0[it][e]; v[4 + e] -= side1[it][e]; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) sumsq += v[e] * v[e];
                    store_bf16(p.C, idx, v);
                } break;
                case EPI_DGELU: {
                    const uint32_t w[4] = {__float_as_uint(side0[it][0]), __float_as_uint(side0[it][1]), __float_as_uint(side0[it][2]),
                                           __float_as_uint(side0[it][3])};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {      // aux = gelu'(pre), saved by the forward epilogue
                        v[2 * e] *= __uint_as_float(w[e] << 16);
                        v[2 * e + 1] *= __uint_as_float(w[e] & 0xffff0000u);
                    }
                    store_bf16(p.C, idx, v);
                } break;
                case EPI_F32_BF16: {
                    store_f32(reinterpret_cast<float*>(p.C) + idx);
                    store_bf16(p.C2, idx, v);
                } break;
                case EPI_RELU: {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    store_bf16(p.C, idx, v);
                } break;
                case EPI_DRELU: {   // aux = the forward ReLU output: gradient passes where it was positive
                    const uint32_t w[4] = {__float_as_uint(side0[it][0]), __float_as_uint(side0[it][1]), __float_as_uint(side0[it][2]),
                                           __float_as_uint(side0[it][3])};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!((w[e] & 0x7fffu) && !(w[e] & 0x8000u))) v[2 * e] = 0.f;
                        if (!((w[e] & 0x7fff0000u) && !(w[e] & 0x80000000u))) v[2 * e + 1] = 0.f;
                    }
                    store_bf16(p.C, idx, v);
                } break;
                case EPI_NCE: {     // v = cos/T (alpha = 1/T): partial sums of exp(v - 1/T) over negatives and of v over positives
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int dj = n + e - m;
                        if (dj == 1 || dj == -1) possum += v[e];
                        else if (dj != 0) sumsq += __expf(v[e] - alpha);
                    }
                } break;
                case EPI_NCE_BWD: { // labels = {lse, pos_coef}: C bf16 = d loss / d v
                    const float lse = p.labels[0], pc = p.labels[1];
                    float w[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int dj = n + e - m;
                        w[e] = dj == 0 ? 0.f : (dj == 1 || dj == -1) ? pc : __expf(v[e] - lse);
                    }
                    store_bf16(p.C, idx, w);
                } break;
                default: break;
            }
        }
    }
    if (do_rowsum && nt > 0 && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WM + 16 * i + lane;
            if (m < p.M) atomicAdd(p.rowsum + m, accb[i][0] * alpha);
        }
    }
    if (epi == EPI_NCE) {    // two partials per tile: [2 tile] = sum over negatives, [2 tile + 1] = sum over positives
        float* red = reinterpret_cast<float*>(smem);
        const float w0 = wave_sum(sumsq), w1 = wave_sum(possum);
        __syncthreads();
        if (lane == 0) { red[wave] = w0; red[4 + wave] = w1; }
        __syncthreads();
        if (tid == 0) {
            p.partial[2 * tile] = (red[0] + red[1]) + (red[2] + red[3]);
            p.partial[2 * tile + 1] = (red[4] + red[5]) + (red[6] + red[7]);
        }
    }
    if (epi == EPI_LOSS) {   // uniform per workgroup: deterministic per-tile partial of sum (logit-label)^2
        float* red = reinterpret_cast<float*>(smem);
        const float w = wave_sum(sumsq);
        __syncthreads();   // every wave is done with its tile in LDS
        if (lane == 0) red[wave] = w;
        __syncthreads();
        if (tid == 0) p.partial[tile] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ------------------------------------------------------------------ host side
static void tile_dims(int cfg, int& bm, int& bn) {
    if (cfg == 10 || cfg == 11) { bm = 256; bn = cfg == 10 ? 256 : 128; return; }    // gemm8.hip
    if (cfg == 12) { bm = 128; bn = 384; return; }                                     // gemm8.hip, weight gradients of 384-multiples
    if (cfg == 14) { bm = 128; bn = 256; return; }                                     // gemm_pp.hip: 128 x 256 units, the two wave rows take turns
    bm = cfg == 2 ? 64 : 128;
    bn = cfg == 0 ? 128 : 64;
}

static int tiles_for(const GemmProblem& p, int cfg) {
    int bm, bn;
    tile_dims(cfg, bm, bn);
    return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
}

// Can the LayerNorm next to a Linear with N output columns ride in that product's epilogue (EPI_RESID_LN / EPI_DLN)?  Full rows in one
// 128 x 384 tile, whole-row offsets inside 32 bits.
bool gemm_row_ln_ok(int M, int N, int K) { return N == 384 && K % 64 == 0 && K >= 128 && M > 0 && (double)M * 1536.0 < 4294000000.0; }

int gemm_pick_tile(const GemmProblem* probs, int nprob, int tile_cfg) {
    if (tile_cfg == 6 || tile_cfg == 7) return tile_cfg - 6;     // the persistent kernel: 128x128 / 128x64 tiles
    if (tile_cfg == 9) return 0;                                  // persistent 128x128 with deferred stores
    if (tile_cfg >= 10 && tile_cfg <= 12) return tile_cfg;        // gemm8.hip: 256x256 / 256x128 / 128x384
    if (tile_cfg == 14) return tile_cfg;                          // gemm_pp.hip
    if (tile_cfg >= 15 && tile_cfg <= 18) return 0;               // gemm_as.hip: 128 x 128 output tiles
    if (tile_cfg >= 0) return tile_cfg;
    // Measured on MI355X (profiles/r01_b_microbench.json): a workgroup's speed is set by its L2->LDS fill
    // rate (~70 GB/s per CU), so the big tile (64 FLOP/B) wins once it alone covers the 256 CUs ~1.5x;
    // below that, more and smaller workgroups win.  Narrow outputs (N <= 384) prefer 128x64 (3 workgroups/CU).
    int t0 = 0, t1 = 0, nmax = 0;
    for (int i = 0; i < nprob; ++i) {
        t0 += tiles_for(probs[i], 0) * probs[i].split_k;
        t1 += tiles_for(probs[i], 1) * probs[i].split_k;
        nmax = probs[i].N > nmax ? probs[i].N : nmax;
    }
    // ... unless there are at least two full rounds of 128x128 tiles anyway (B >= ~40 decoder shapes): then the big tile is as
    // fast at K = 384 and 15 % faster at K = 1536 (profiles/r01_e_gemm_ksweep_b64.txt, N = 384 rows), and it can be chained
    if (nmax <= 384) return t0 >= 1024 ? 0 : (t1 >= 256 ? 1 : 2);
    if (t0 >= 400) return 0;
    if (t1 >= 400) return 1;
    return 2;
}

// Column tiles per panel for the kernel's tile walk (see gemm_kernel).  Model: an XCD owns rows_x = tiles_m / 8 tile rows of
// the problem; with panels of G column tiles the A rows are fetched once per panel, and the B panel is fetched once if it fits
// ~2 MiB of the L2 and otherwise once per resident wave of 64 workgroups.  Candidates: no panels, or the widest panel that fits.
static int pick_panel(const GemmProblem& p, int cfg, GemmLayout layout) {
    if (BVC_EXP_ENV("BVC_GEMM_LEGACY_WALK") != nullptr) return 0;     // experiments build: read per launch for same-process A/Bs
    if (const char* pe = BVC_EXP_ENV("BVC_GEMM_PANEL")) {             // experiments build: force the panel width (A/B of the walk model)
        const int tn = (p.N + (cfg == 10 ? 256 : 128) - 1) / (cfg == 10 ? 256 : 128), gq = atoi(pe);
        if (layout != GEMM_TN && gq > 0) return gq < tn ? gq : tn;
    }
    // Same-box A/B at B=64 (profiles/r01_e_walk_ab_b64.txt): the panel walk is worth +5 % on the encoder fc1 shape and is
    // neutral elsewhere for NT / NN; the split-K weight-gradient launches are 2-10 % FASTER with the legacy walk (splits
    // fastest, short side first) although it fetches more - the Infinity Cache absorbs the re-reads - so TN keeps it.
    if (layout == GEMM_TN) return 0;
    int bm, bn;
    tile_dims(cfg, bm, bn);
    const int tiles_m = (p.M + bm - 1) / bm, tiles_n = (p.N + bn - 1) / bn;
    const double kper = (double)((p.K + p.split_k - 1) / p.split_k);
    const double a_slab = bm * kper * 2.0, b_slab = bn * kper * 2.0, cap = 2.0 * 1024 * 1024;
    const double rows_x = tiles_m / 8.0 > 1.0 ? tiles_m / 8.0 : 1.0;
    auto cost = [&](int G) {
        const double npan = (double)((tiles_n + G - 1) / G);
        const double waves = rows_x * G / 64.0 > 1.0 ? rows_x * G / 64.0 : 1.0;
        return rows_x * a_slab * npan + tiles_n * b_slab * (G * b_slab <= cap ? 1.0 : waves);
    };
    int gmax = (int)(cap / b_slab);
    if ((cfg == 10 || cfg == 11) && gmax < 2) {
        // Long K on the one-workgroup-per-CU kernel (round 4): not even two column tiles' B slabs fit the L2 budget, so no panel is
        // kept across rounds whatever G is - what counts is the set of tiles RESIDENT on the XCD (32 workgroups that started together
        // and advance along K in step): with G = 1 they are 32 row tiles of one column, 33 distinct operand tiles per K step for 64
        // fetched - rocprofv3 measures an L2 hit rate of 0.485 on the 8192^3 square (profiles/r04_k_pmc_mem_g8.txt) and half of all
        // fills cross the fabric.  A square-ish resident set (G = 6: 5.3 x 6 tiles, 11.3 distinct per 64 fetched) brings the square
        // from 907 to 742 us = 1.48 PFLOP/s (profiles/r04_k_panel_ab.txt); the step's K >= 1536 products (tiles_n = 3) are neutral.
        const int gq = cfg == 10 ? 6 : 8;
        return gq < tiles_n ? gq : tiles_n;
    }
    if (gmax < 1) gmax = 1;
    if (gmax >= tiles_n) return tiles_n;
    const int npan = (tiles_n + gmax - 1) / gmax;
    const int G = (tiles_n + npan - 1) / npan;
    return cost(G) < cost(tiles_n) ? G : tiles_n;
}

static int pick_gemm8(const GemmProblem* probs, int nprob, GemmLayout layout);
// (loss partials are written per tile of the kernel that launch_gemm picks for an NT problem: the same selection, gemm8 included)
int gemm_num_tiles(const GemmProblem& p, int tile_cfg) {
    if (tile_cfg < 0) {
        const int g8 = pick_gemm8(&p, 1, GEMM_NT);
        if (g8 > 0) tile_cfg = g8;
    }
    return tiles_for(p, gemm_pick_tile(&p, 1, tile_cfg));
}

template <int BM, int BN, bool AT, bool BT, int NS, int LB = 2, bool EARLY = true>
static int launch_one(const GemmGroup& g, int nblocks, hipStream_t stream) {
    constexpr size_t lds = (size_t)NS * (BM + BN) * 64 * 2;
    if (dry_run().on) {
        snprintf(dry_run().name, sizeof(dry_run().name), "bvc::gemm_kernel<%d, %d, %s, %s, %d, %d, %s>", BM, BN, AT ? "true" : "false",
                 BT ? "true" : "false", NS, LB, EARLY ? "true" : "false");
        return BVC_OK;
    }
    static bool attr_set = false;   // > 64 KiB of dynamic LDS needs the attribute once per kernel
    if (lds > 65536 && !attr_set) {
        BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, AT, BT, NS, LB, EARLY>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, AT, BT, NS, LB, EARLY>), dim3(nblocks), dim3(256), lds, stream, g);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

template <int BM, int BN, int NS>
static int launch_cfg(const GemmGroup& g, GemmLayout layout, int nblocks, hipStream_t stream) {
    switch (layout) {
        case GEMM_NT: return launch_one<BM, BN, false, false, NS>(g, nblocks, stream);
        case GEMM_NN: return launch_one<BM, BN, false, true, NS>(g, nblocks, stream);
        case GEMM_TN: return launch_one<BM, BN, true, true, NS>(g, nblocks, stream);
        default: set_error("launch_gemm: bad layout %d", (int)layout); return BVC_ERR_INVALID;
    }
}

template <int BM, int BN>
static int launch_stages(const GemmGroup& g, GemmLayout layout, int stages, int nblocks, hipStream_t stream) {
    // experiment hooks (same-run A/B; box-to-box variance on the pool is +-40 %): stages 3 = AGPR-form MFMA for TN (measured
    // 5-25 % slower), stages 4 = the other K-loop variant for the layout
    if (stages == 3 && layout == GEMM_TN) return launch_one<BM, BN, true, true, 2, 1>(g, nblocks, stream);
    if (stages == 4) {
        if (layout == GEMM_NT) return launch_one<BM, BN, false, false, 2, 2, false>(g, nblocks, stream);
        if (layout == GEMM_NN) return launch_one<BM, BN, false, true, 2, 2, false>(g, nblocks, stream);
        return launch_one<BM, BN, true, true, 2, 2, false>(g, nblocks, stream);
    }
    return launch_cfg<BM, BN, 2>(g, layout, nblocks, stream);
}

// The K loop keeps two K-steps of LDS-DMA in flight out of two LDS slots (see gemm_kernel); deeper rings were measured
// slower (they cost the second resident workgroup per CU), so `stages` is accepted for API stability and ignored.
int gemm_pick_stages(int, GemmLayout, int, int stages) { return (stages == 3 || stages == 4) ? stages : 2; }

int launch_gemm_big_nt(const GemmProblem& p, int cfg, hipStream_t stream);   // experiments/gemm_big.hip (tile configs 3-5, experiments build only)
// gemm_persist.hip; returns 1 when the problem is not eligible.  defer: 0 = stores in the epilogue, 1 = deferred where possible,
// 2 = deferred or not at all (tile config 9, tests)
int launch_gemm_persist(const GemmGroup& g, GemmLayout layout, int cfg, hipStream_t stream, int defer);
// gemm8.hip: 256 x bn tiles, one 512-thread workgroup per CU; returns 1 when the group is not eligible
int launch_gemm8(const GemmGroup& g, GemmLayout layout, int bn, hipStream_t stream);
// gemm_as.hip: A-stationary kernel for K = 384 products with bf16 outputs (tile configs 15 / 16); returns 1 when the problem is not eligible
int launch_gemm_as(const GemmProblem& p, GemmLayout layout, int variant, hipStream_t stream);
bool gemm_as_ok(const GemmProblem& p, GemmLayout layout);
// experiments/gemm_pp.hip (experiments build only): 128 x 256 units, K loop of one wave row under the epilogue of the other;
// built and measured in round 4 (profiles/r04_d_*): correct, bit-identical, and NOT faster - see its header
int launch_gemm_pp(const GemmGroup& g, GemmLayout layout, hipStream_t stream);

// Which products go to the 256-row persistent kernel (gemm8.hip) when the caller leaves the tile choice open.  Fitted to the
// same-process A/Bs of every product of the step at 16, 64 and 256 clips (profiles/r02_e_gemm8_ab_b{16,64,256}.txt, tools/ab/gemm8_ab.py):
// its K loop runs ~1.2 PFLOP/s against ~0.8 for the 128 x 128 kernels, but it is ONE workgroup per CU - a launch needs about two
// full rounds of tiles and ~45 GFLOP to amortise prologue and tail, and an epilogue is not hidden by a second resident workgroup:
//   * 256 x 256: plain bf16-output epilogues (BF16 / GELU / RELU) and f32-output ones (residual, positional, plain) with >= 448
//     tiles whose last column tile is at least 85 % full (encoder qkv / fc1 / dX at >= 64 clips, decoder qkv / fc1: -20 ... -27 %
//     at 256 clips, -5 ... -10 % at 64; encoder proj / fc2 at 256 clips);
//   * 256 x 128: input-gradient products (NN, bf16 out) and residual products (f32 + residual) with K >= 1024 and >= 224 tiles
//     when the 256-wide tile would be half empty (decoder N = 384: dX-fc1, dX-qkv, fc2);
//   * never: GELU' / ReLU' (their side-input epilogue still parks through LDS and drains the prefetch: no gain measured),
//     launches below 45 GFLOP (at 16 clips gemm8 loses 5-50 % on every product).
// Returns 10 / 11 (tile configs) or -1.
static int pick_gemm8(const GemmProblem* probs, int nprob, GemmLayout layout) {
    const int mode = options().gemm8;
    if (mode < 0 || nprob != 1 || layout == GEMM_TN) return -1;
    const GemmProblem& p = probs[0];
    if (p.split_k != 1 || p.K % 64 != 0) return -1;
    if (p.a_bytes >= 0x80000000u || p.b_bytes >= 0x80000000u) return -1;    // gemm8 addresses operands below 2 GiB (its out-of-range sentinel)
    // (the gated epilogues - GELU' / ReLU' - joined once the forward saved gelu' itself: one multiply per element instead of ~14
    //  vector instructions nothing hid on this kernel; decoder dX-fc2 at 256 clips 769 vs 878 us - profiles/r02_i_gelu_grad_saved.txt)
    const bool gated = p.epi == EPI_DGELU || p.epi == EPI_DRELU;
    const bool bf = p.epi == EPI_BF16 || p.epi == EPI_GELU || p.epi == EPI_RELU;
    // f32 out (+ f32 side input); the head + MSE product (f32 labels in, bf16 difference out, per-tile loss partials) rides the same
    // class since round 3: 868 vs 1120 us at 256 clips, 256 vs 303 at 64 (tools/debug/head_loss_ab.py)
    const bool resid = (p.epi == EPI_RESID || p.epi == EPI_POS || p.epi == EPI_F32 || p.epi == EPI_LOSS) && layout == GEMM_NT;
    if (!bf && !resid && !gated) return -1;
    const int tm = (p.M + 255) / 256, tn256 = (p.N + 255) / 256, tn128 = (p.N + 127) / 128;
    // the register-epilogue class keeps the tile-padded bias vector in 32 KiB of LDS (launch_gemm8 refuses wider outputs)
    const bool fits256 = !bf || (size_t)tn256 * 256 * 4 <= 32768, fits128 = !bf || (size_t)tn128 * 128 * 4 <= 32768;
    if (mode > 0) {      // forced (tests, A/B tools): the widest tile the output fills at least half of
        if (fits256 && (gated || p.N > 128)) return 10;
        return fits128 && !gated ? 11 : -1;
    }
    if (2.0 * p.M * p.N * p.K < 45e9) return -1;
    const bool full256 = (double)p.N >= 0.85 * 256.0 * tn256;
    // (the f32 class runs its side inputs in four passes on 256 x 256 tiles: encoder proj / fc2 / patch embedding at 256 clips
    //  -10 / -21 / -17 %, a loss below 448 tiles - profiles/r02_g_gemm8_resid_ab.txt)
    if (full256 && fits256 && tm * tn256 >= 448) return 10;
    if (gated) return -1;                   // 256 x 128 tiles lose on them at every size measured
    if (p.epi != EPI_LOSS && fits128 && (bf ? layout == GEMM_NN : p.N <= 384) && p.K >= 1024 && tm * tn128 >= (resid ? 1024 : 224)) return 11;
    return -1;
}

int launch_gemm(const GemmProblem* probs, int nprob, GemmLayout layout, int tile_cfg, hipStream_t stream, int stages) {
    BVC_REQUIRE(nprob >= 1 && nprob <= kMaxGroup, "launch_gemm: nprob %d out of range", nprob);
    static thread_local bool skip_g8 = false;      // set while an auto-picked gemm8 launch that turned the problem down is re-planned
    bool auto_g8 = false;
    if (probs[0].epi == EPI_RESID_LN || probs[0].epi == EPI_DLN) {
        // LayerNorm fused into a 384-wide product: exists on the full-row tile of gemm8.hip only (the caller asks gemm_row_ln_ok first)
        const GemmProblem& p = probs[0];
        BVC_REQUIRE(nprob == 1 && (tile_cfg < 0 || tile_cfg == 12), "launch_gemm: the LayerNorm epilogues take one problem on tile config 12");
        BVC_REQUIRE(gemm_row_ln_ok(p.M, p.N, p.K) && p.ldc == p.N && p.split_k == 1 && p.a_bytes < 0x80000000u && p.b_bytes < 0x80000000u,
                    "launch_gemm: the LayerNorm epilogues need N == ldc == 384, K %% 64 == 0, operands below 2 GiB (M=%d N=%d K=%d)", p.M, p.N, p.K);
        BVC_REQUIRE(p.C && p.C2 && p.ln_mean && p.ln_rstd && p.ln_gamma, "launch_gemm: LayerNorm epilogue with a null output / statistic / scale");
        if (p.epi == EPI_RESID_LN) BVC_REQUIRE(layout == GEMM_NT && p.resid && p.ln_beta, "launch_gemm: RESID_LN is an NT product with a residual and a LayerNorm bias");
        if (p.epi == EPI_DLN) BVC_REQUIRE(layout == GEMM_NN && p.ln_x && p.ln_part && p.ln_dgamma && p.ln_dbeta, "launch_gemm: DLN is an NN product with the LayerNorm input, partial scratch and parameter gradients");
        tile_cfg = 12;
    }
    // K = 384 products with a plain bf16 output (decoder / predictor qkv): the A-stationary kernel (gemm_as.hip), whose epilogue runs under
    // the next N tile's MFMAs.  Same-process A/B against the kernels picked below (profiles/r05_l_as_ab_batches.txt, r05_i_as_ab_b256.txt):
    // decoder qkv -6.5 % at 16 clips, -10 % at 32 ... 128, -11.5 % at 256.  Its GELU form does not win (the GELU arithmetic beside the
    // MFMAs costs more than it hides: +13 ... +22 % overlapped, -6 ... +5 % behind the tile) and stays a tile config for A/Bs.
    if (tile_cfg < 0 && stages < 0 && nprob == 1 && options().gemm8 >= 0 && probs[0].epi == EPI_BF16 && gemm_as_ok(probs[0], layout) &&
        (options().gemm8 > 0 || probs[0].M >= 16384) && BVC_EXP_ENV("BVC_GEMM_NO_AS") == nullptr)
        return launch_gemm_as(probs[0], layout, 0, stream);
    if (tile_cfg < 0 && stages < 0 && !skip_g8) {
        const int g8 = pick_gemm8(probs, nprob, layout);
        if (g8 > 0) { tile_cfg = g8; auto_g8 = true; }
    }
    if (tile_cfg >= 15 && tile_cfg <= 18) {      // gemm_as.hip: A-stationary kernel for K = 384 (16: epilogue behind its own tile; 17 / 18: the same two with late LDS-DMA; A/Bs)
        BVC_REQUIRE(nprob == 1, "launch_gemm: tile configs 15 - 18 take one problem");
        const int rc = launch_gemm_as(probs[0], layout, tile_cfg - 15, stream);
        BVC_REQUIRE(rc != 1, "launch_gemm: tile configs 15 - 18 (A-stationary kernel) take NT products with K = 384, N %% 128 == 0, BF16 / GELU epilogues");
        return rc;
    }
    if ((tile_cfg >= 3 && tile_cfg <= 5) || tile_cfg == 8) {
#ifdef BVC_EXPERIMENTS
        BVC_REQUIRE(nprob == 1 && layout == GEMM_NT, "launch_gemm: tile configs 3-5 (32-deep K steps) are NT, one problem");
        return launch_gemm_big_nt(probs[0], tile_cfg, stream);
#else
        BVC_REQUIRE(false, "launch_gemm: tile configs 3-5 / 8 exist only in a -DBVC_EXPERIMENTS build (csrc/experiments/gemm_big.hip)");
#endif
    }
    // tile config 13 = tile config 10 (256 x 256, gemm8.hip) for weight gradients whose outputs are ACCUMULATED: C += dY^T X by f32
    // atomics whether K is split or not (C pre-zeroed, as for split_k > 1).  An unsplit group that fills only part of the chip can
    // then take the balanced walk (plan_balance in gemm8.hip); plan_dw returns it for such groups.
    bool accum = false;
    if (tile_cfg == 13) {
        BVC_REQUIRE(layout == GEMM_TN, "launch_gemm: tile config 13 (accumulating 256 x 256 tiles) is for weight gradients (TN)");
        for (int i = 0; i < nprob; ++i) BVC_REQUIRE(probs[i].epi == EPI_F32, "launch_gemm: tile config 13 takes plain f32 outputs");
        accum = true;
        tile_cfg = 10;
    }
    const int cfg = gemm_pick_tile(probs, nprob, tile_cfg);
    GemmGroup g;
    g.nprob = nprob;
    g.bal_units = g.bal_lb = g.bal_tiles = 0;
    g.accum = accum ? 1 : 0;
    g.stagger = 0;
    {
        const char* e = BVC_EXP_ENV("BVC_GEMM_DEBUG");
        g.dbg = e ? atoi(e) : 0;
    }
    int total = 0;
    for (int i = 0; i < nprob; ++i) {
        const GemmProblem& p = probs[i];
        BVC_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "launch_gemm: empty problem %d (%d,%d,%d)", i, p.M, p.N, p.K);
        BVC_REQUIRE(p.N % 8 == 0, "launch_gemm: N=%d must be a multiple of 8", p.N);
        BVC_REQUIRE(p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 8 == 0, "launch_gemm: leading dims must be multiples of 8");
        if (layout != GEMM_TN) BVC_REQUIRE(p.K % 64 == 0, "launch_gemm: K=%d must be a multiple of 64 for k-contiguous operands", p.K);
        if (layout == GEMM_TN) BVC_REQUIRE(p.M % 8 == 0, "launch_gemm: TN needs M %% 8 == 0 (M=%d)", p.M);
        BVC_REQUIRE(p.split_k >= 1, "launch_gemm: split_k must be >= 1");
        if (p.rowsum) BVC_REQUIRE(layout == GEMM_TN, "launch_gemm: rowsum (bias gradient) is fused into TN products only");
        if (p.split_k > 1)
            BVC_REQUIRE(p.epi == EPI_F32 || (p.epi == EPI_RESID && p.resid == p.C),
                        "launch_gemm: split_k needs an accumulating f32 epilogue");
        g.prob[i] = p;
        g.panel[i] = pick_panel(p, cfg, layout);
        if (cfg >= 10 && cfg <= 12 && layout == GEMM_TN && BVC_EXP_ENV("BVC_G8_TN_LEGACY_WALK") == nullptr) {
            // weight gradients on the persistent kernel: K splits SLOWEST, tiles row-major inside a split.  An XCD's ~32 resident
            // units are then the tiles of one or two K ranges of one problem, which share their dY / X slices through its L2; with
            // the splits fastest (the 128 x 128 kernel's walk) neighbours share nothing and every unit streams its own slices
            // from HBM: 15.6 GB per decoder layer at 256 clips for 4.5 GB of operands (profiles/r02_f_dw_walk_ab.txt).
            int bm, bn;
            tile_dims(cfg, bm, bn);
            g.panel[i] = (p.N + bn - 1) / bn;
            if (BVC_EXP_ENV("BVC_G8_TN_SHORT_FAST") != nullptr) g.panel[i] = -1;
        }
        g.tile_start[i] = total;
        total += tiles_for(p, cfg) * p.split_k;
    }
    g.tile_start[nprob] = total;
    for (int i = nprob; i < kMaxGroup; ++i) { g.prob[i] = probs[0]; g.panel[i] = g.panel[0]; g.tile_start[i + 1] = total; }
    if (cfg == 14) {
#ifdef BVC_EXPERIMENTS
        const int rc = launch_gemm_pp(g, layout, stream);
        BVC_REQUIRE(rc != 1, "launch_gemm: tile config 14 (ping-pong kernel) does not take this problem");
        return rc;
#else
        BVC_REQUIRE(false, "launch_gemm: tile config 14 exists only in a -DBVC_EXPERIMENTS build (csrc/experiments/gemm_pp.hip)");
#endif
    }
    if (cfg >= 10 && cfg <= 12) {
        const int rc = launch_gemm8(g, layout, cfg == 10 ? 256 : cfg == 11 ? 128 : 384, stream);
        if (rc == 1 && auto_g8) {     // the selection and the kernel's own eligibility test disagree: never an error for the caller
            skip_g8 = true;
            const int rc2 = launch_gemm(probs, nprob, layout, -1, stream, stages);
            skip_g8 = false;
            return rc2;
        }
        BVC_REQUIRE(rc != 1, "launch_gemm: tile configs 10 - 12 (persistent one-workgroup-per-CU kernel) do not take this problem");
        return rc;
    }
    int kmax = 0;
    for (int i = 0; i < nprob; ++i) kmax = probs[i].K > kmax ? probs[i].K : kmax;
    int ns = gemm_pick_stages(cfg, layout, kmax, stages);
    // One round of 128x128 tiles (the encoder's proj / fc2 / dX-proj at B=64: 480 workgroups): the plain double buffer beat the
    // early-refill loop by 6-11 % for NT and for NN at K <= 768 in the same-process tile sweep (profiles/r01_f_tile_sweep_b64.txt),
    // and tied elsewhere - with every workgroup resident at once there is no second round whose prologue the deeper prefetch hides.
    if (stages < 0 && nprob == 1 && cfg == 0 && total <= 512 && probs[0].split_k == 1 &&
        (layout == GEMM_NT || (layout == GEMM_NN && probs[0].K <= 768)))
        ns = 4;
    // short-K single products with many tile rounds go to the persistent kernel (tile config 6 forces it, for tests);
    // BVC_GEMM_NO_PERSIST=1 keeps them on gemm_kernel (same-process A/B)
    // Same-box A/B at B=64 (profiles/r01_f_persist_ab_b64.txt): 128x128 persistent -12 ... -15 % on every eligible product; the
    // 128x64 form wins only at short K (decoder proj, K=384: -5 %) and LOSES 6-16 % at K >= 1152, where the per-tile kernel's
    // third resident workgroup per CU matters more than the chaining - so it is taken up to K = 512 only.
    const bool auto_persist = tile_cfg < 0 && stages < 3 && g.dbg == 0 && (cfg == 0 || (cfg == 1 && probs[0].K <= 512)) &&
                              BVC_EXP_ENV("BVC_GEMM_NO_PERSIST") == nullptr;
    if (nprob == 1 && (tile_cfg == 6 || tile_cfg == 7 || tile_cfg == 9 || auto_persist)) {
        // deferred stores (gemm_persist.hip) measured on the decoder shapes at B=64 (profiles/r01_f_gemm_ksweep_b64.txt, tile 9 vs 6):
        // GELU' epilogue -8 ... -18 % at K <= 384, -2 % at K = 768; plain bf16 +-0; GELU (two outputs, 253 VGPRs) +8 % slower.
        // So: the GELU' products only.  BVC_GEMM_DEFER=1 / =0 force it on (where possible) / off for A/Bs.
        const char* dv = BVC_EXP_ENV("BVC_GEMM_DEFER");
        const int defer = tile_cfg == 9 ? 2 : tile_cfg >= 0 ? 0 : dv ? (dv[0] == '1' ? 1 : 0) : (probs[0].epi == EPI_DGELU ? 1 : 0);
        const int rc = launch_gemm_persist(g, layout, cfg, stream, defer);
        if (rc != 1) return rc;
        BVC_REQUIRE(tile_cfg < 6, "launch_gemm: tile configs 6 / 7 / 9 (persistent kernels) do not take this problem");
    }
    switch (cfg) {
        case 0: return launch_stages<128, 128>(g, layout, ns, total, stream);
        case 1: return launch_stages<128, 64>(g, layout, ns, total, stream);
        default: return launch_stages<64, 64>(g, layout, ns, total, stream);
    }
}

}  // namespace bvc
