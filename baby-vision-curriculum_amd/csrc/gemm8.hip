// Deep-pipelined persistent bf16 GEMM for gfx950: ONE 512-thread workgroup per CU, 256 x BN output tiles (BN = 256 / 128),
// 64-deep K tiles, four phases per K tile, LDS-DMA prefetch that stays in flight across raw barriers behind COUNTED vmcnt waits.
//
// Why a second GEMM structure (profiles/r01_e_gemm_ksweep_b64.txt, r02_*): the 128 x 128 kernels of gemm.hip / gemm_persist.hip
// need one byte of L2 -> LDS fill per 64 FLOP and the CU's fill path (~70 GB/s) then caps them near 0.8 PFLOP/s.  A 256 x 256
// tile needs one byte per 128 FLOP.  It only fits as one workgroup per CU (128 KiB of staging), so latency has to be hidden
// inside the workgroup instead of by a second resident workgroup:
//   * 8 waves as 2 (M) x 4 (N); wave tile 128 x BN/4; accumulators 8 x BN/64 MFMA tiles of v_mfma_f32_16x16x32_bf16;
//   * the two wave rows run STAGGERED by one barrier (wave row 1 executes one extra s_barrier at entry, wave row 0 one at
//     exit): while one wave of a SIMD issues its 16 MFMAs of a phase the other one issues its LDS reads and its LDS-DMA;
//   * per phase: {LDS fragment reads, one group of LDS-DMA pieces of a K tile one-to-two tiles ahead, [counted vmcnt],
//     s_barrier, lgkmcnt(0), 16 MFMAs, s_barrier}.  vmcnt never reaches 0 inside the stream of K tiles;
//   * an LDS region is re-staged no earlier than two phases after its last fragment read (with the stagger, the other wave
//     row's reads of phase p retire only after this row's first barrier of phase p+1), and read no earlier than the phase
//     after the wait that retires it;
//   * persistent: a workgroup walks its units (output tile x K split) as ONE stream of K tiles, the prefetch runs across unit
//     boundaries, so the next unit's first K tiles arrive under the current unit's last MFMAs and its epilogue.
// Phase order inside a K tile:
//   * k-contiguous A (NT / NN): "m-major" - all B fragments of the K tile are read in phase 0 and kept, phase q multiplies
//     rows 32q..32q+31 of the wave tile.  A pieces (64 rows) free up after phases 1 and 3, the B tile after phase 0;
//   * transposed A (TN, weight gradients): "k-major" - phases (k half, row half); the [64 k][128] images free up by k half.
// Either way a K tile is staged as four groups g0..g3, one per phase, in the order they become free, and phase q issues
// group (q + 2) & 3; the stream of groups never skips: past the last unit the loads are issued with an out-of-range offset
// (the buffer descriptor drops them, the counter still counts them), which keeps every counted wait exact.
// Results are bit-identical to gemm_kernel's for the same problem (same K order per output element, same epilogue math).
//
// Round 3: a third geometry for weight gradients, 128 x 384 tiles (BM = 128, BN = 384; tile config 12).  Every width of the
// decoder and of the JEPA predictor is a multiple of 384 (384, 1152, 1536): on 256-wide tiles the four weight gradients of a
// decoder layer fill 71 % of the MFMA work they execute (38 tiles of 65536 outputs for 1.77 M outputs), on 128 x 384 tiles 100 %
// (36 tiles of 49152).  Same stream, same phases: wave tile 64 x 96 (4 x 6 MFMA tiles, 12 MFMAs per phase), one A half and three
// B halves per K tile (again 128 KiB of staging), groups of 3 + 1 + 3 + 1 LDS-DMA pieces.
//
// Round 5: the 128 x 384 geometry for k-contiguous A (NT / NN) with ROW epilogues (EC = 4 / 5).  The decoder and the JEPA predictor are
// 384 wide, so a 128 x 384 tile holds COMPLETE rows of a Linear's output and the LayerNorm next to that Linear can run in its epilogue:
//   EC = 4 (NT, proj / fc2):  v = alpha acc + bias + residual -> C (f32 residual stream); mean / rstd of every row (wave-local sums
//     over the wave's 96 columns, Chan's combination of the four wave columns through 4 KiB of LDS, one barrier) and the bf16
//     LayerNorm output (the NEXT product's A operand) -> C2: the separate ln_fwd pass (read 1.5 KB, write 0.75 KB per row) is gone;
//   EC = 5 (NN, dX of fc1 / qkv):  g = alpha acc is d/d(LayerNorm output); the epilogue reads the LayerNorm's f32 input and the incoming
//     residual gradient, reduces mean(g gamma) and mean(g gamma xhat) per row, writes dres += rstd (g gamma - s1 - xhat s2) in f32 and as
//     the bf16 operand copy, and folds the dgamma / dbeta column sums (16-lane DPP reductions into wave-private LDS accumulators, one
//     partial row per workgroup at kernel end): the separate ln_bwd pass (d(ln out) written and re-read, 3.8 KB read per row) is gone.
// Same stream and phases as the 256-row tiles; a wave ROW takes the row tiles 32 q + 16 wm (q = phase), so that A piece 0 (rows 0-63)
// is read in phases 0-1 and piece 1 in phases 2-3 by both wave rows - the freeing order the staging groups assume.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm_tile.h"
#include "rowops.h"

namespace bvc {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// cache policy of the epilogue's output stores (aux operand of the buffer store: bit 1 = nt, "non-temporal").  Outputs are written
// once and read by a LATER kernel; left at the default policy they sit in the XCD's 4 MiB L2 next to the operand slabs the
// resident units share (a round of 32 units writes 4-8 MiB).  -DBVC_G8_ST_AUX=2 marks them streaming (A/B: profiles/r03_i_*).
#ifndef BVC_G8_ST_AUX
#define BVC_G8_ST_AUX 0
#endif
// the same switch for the row epilogues (EC 4 / 5), whose outputs are the bulk of their HBM traffic (A/B: profiles/r05_p_*)
#ifndef BVC_G8_ROW_ST_AUX
#define BVC_G8_ROW_ST_AUX 0
#endif
constexpr uint32_t kInvalidBase = 0x80000000u;   // beyond every operand this kernel accepts (extents < 2 GiB)

struct Unit {
    int pi, m0, n0, kt0, nkt, split, tile;
};

// Transposed-operand fragment out of a [64 k][128] LDS image (layout and swizzle of gemm_tile.h read_frag<128, true>), with the
// address arithmetic spelled out so that it costs two vector instructions per fragment instead of one live address register
// per (row tile, LDS slot): byte address = region + 8192 ks + lane_base + ((32 row_tile) ^ lane_swz), second half 1024 further.
//   lane_base = (8 (l >> 4) + ((l >> 2) & 3)) * 256 + ((l & 3) >> 1) * 16 + (l & 1) * 8,   lane_swz = swz_tr<128>(k) << 4
// (the swizzle only touches address bits 5-7, exactly the bits 32 * row_tile occupies).  The kernel passes lane_swz through an
// empty asm once per phase: hipcc then recomputes the addresses per phase instead of hoisting ~24 of them out of the K loop,
// which is what pushed the 256 x 256 weight-gradient instantiation over 256 registers.
__device__ __forceinline__ bf16x8 tr_frag(uint32_t region, uint32_t lane_base, uint32_t lane_swz, uint32_t chunk16, int ks) {
    const uint32_t a = region + lane_base + (chunk16 ^ lane_swz) + (uint32_t)ks * 8192u;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 bf16x4*)(size_t)a);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 bf16x4*)(size_t)(a + 1024u));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// sum over the 16 lanes of a DPP row (lanes with equal l >> 4): four rotations, every lane ends with the full sum
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));   // row_ror:1
    return v;
}

// unit id -> problem, tile, K range.  Uniform (kernel arguments and blockIdx only).
template <int BM, int BN>
__device__ __forceinline__ void decode_unit(const GemmGroup& g, int uid, Unit& u) {
    // balanced weight-gradient walk (GemmGroup::bal_*): ids below bal_units are the full-length units, id bal_units + t is the tail of
    // tile t (tiles counted problem after problem, row-major); tile_start counts units, i.e. tiles x split_k (uniform over the group)
    const bool tail = g.bal_units > 0 && uid >= g.bal_units;
    const int S = tail ? g.prob[0].split_k : 1;
    const int key = tail ? (uid - g.bal_units) * S : uid;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if (i < g.nprob && key >= g.tile_start[i]) pi = i;
    const GemmProblem& p = g.prob[pi];
    const int lid = tail ? (key - g.tile_start[pi]) / S : uid - g.tile_start[pi];
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const int G = g.panel[pi];
    int split, tm, tn;
    if (tail) {
        split = S;
        tm = lid / tiles_n; tn = lid - tm * tiles_n;
    } else if (G > 0) {
        split = lid / ntiles;
        tile_of(lid - split * ntiles, tiles_m, tiles_n, G, tm, tn);
    } else if (G == -1) {   // K splits slowest, tiles with the SHORTER side fastest: a run of consecutive units (one XCD's share) is a compact block
        split = lid / ntiles;
        const int w = lid - split * ntiles;
        if (tiles_n > tiles_m) { tm = w % tiles_m; tn = w / tiles_m; } else { tm = w / tiles_n; tn = w % tiles_n; }
    } else {   // K splits fastest, then along the shorter side (the weight-gradient walk of gemm_kernel)
        split = lid % p.split_k;
        const int tl = lid / p.split_k;
        const bool m_fast = tiles_n > tiles_m;
        tm = m_fast ? tl % tiles_m : tl / tiles_n;
        tn = m_fast ? tl / tiles_m : tl % tiles_n;
    }
    const int nt_all = (p.K + 63) / 64;
    int per = (nt_all + p.split_k - 1) / p.split_k;
    per = (per + 1) & ~1;                        // the K loop is unrolled over two K tiles (LDS slot parity)
    int kt0 = split * per;
    if (g.bal_units > 0) {                       // splits of bal_lb K tiles (even), the tail takes what is left of the tile
        kt0 = split * g.bal_lb;
        per = tail ? (nt_all - kt0 + 1) & ~1 : g.bal_lb;
    }
    // integer divisions run on the vector ALU: pin the (uniform) results to scalar registers
    u.pi = pi;
    u.m0 = __builtin_amdgcn_readfirstlane(tm * BM);
    u.n0 = __builtin_amdgcn_readfirstlane(tn * BN);
    u.kt0 = __builtin_amdgcn_readfirstlane(kt0);
    u.nkt = __builtin_amdgcn_readfirstlane(per);
    u.split = __builtin_amdgcn_readfirstlane(split);
    u.tile = __builtin_amdgcn_readfirstlane(tm * tiles_n + tn);
}

}  // namespace

// EC (epilogue class): 0 = bf16 outputs without side inputs (BF16, GELU, RELU); 3 = bf16 outputs gated by a bf16 side input (DGELU,
// DRELU); 1 = f32 side inputs / outputs (F32, RESID, POS, E2D, LOSS, F32_BF16); 2 = weight gradients (TN): f32 store or split-K
// atomics, fused bias gradient.  Classes are separate instantiations because the side inputs of a whole unit sit in registers.
template <int BM, int BN, bool AT, bool BT, int EC>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(const GemmGroup g, const int total_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(BM == 256 || (BM == 128 && BN == 384), "128-row tiles: the 128 x 384 geometry only");
    static_assert(BN == 256 || BN == 128 || (BN == 384 && BM == 128), "column tiles of 256 / 128, or 384 with 128 rows");
    static_assert((EC >= 4) == (BM == 128 && !AT), "row epilogues (EC 4 / 5) are the k-contiguous-A instantiations of the 128 x 384 tile, and only they");
    static_assert(BM == 256 || EC >= 4 || (AT && BT && EC == 2), "128 x 384 with transposed A: weight gradients");
    static_assert(EC != 4 || !BT, "EC 4 (residual + LayerNorm forward) is an NT product");
    constexpr int A_BYTES = BM * 64 * 2, B_BYTES = BN * 64 * 2, TILE = A_BYTES + B_BYTES;
    constexpr int WMR = BM / 2;                          // rows of a wave row
    constexpr int WN = BN / 4, TN = WN / 16, TM = WMR / 16, TMH = TM / 2;
    constexpr int NB = BN / 64, NBH = BN / 128, NAH = BM / 128;
    constexpr int n0c = AT ? NBH : (NB == 6 ? 3 : NB < 2 ? NB : 2), n1c = AT ? NAH : NB - n0c, n2c = AT ? NBH : BM / 128, n3c = AT ? NAH : BM / 128;
    constexpr int RT = BM / 128;                         // k-contiguous A: row tiles of a wave per phase
    constexpr int KMAX = (n0c > 2 || n1c > 2 || n2c > 2 || n3c > 2) ? 3 : 2;      // most LDS-DMA pieces in one group
    constexpr int PT = n0c + n1c + n2c + n3c;          // LDS-DMA instructions per wave per K tile
    // Where a phase issues its group.  EARLY (default): in the read segment, after the fragment reads and before the counted wait -
    // the wave stalls on the LDS-DMA issue (100-185 cycles per instruction there) while the OTHER wave row of its SIMD issues
    // MFMAs.  LATE (-DBVC_G8_DMA_LATE, kept as a measured alternative): between the MFMAs of the phase; a wave issues in order, so
    // its MFMAs queue behind the stalled DMA issue - same-box A/B (profiles/r02_c_gemm8_dma_placement_ab.txt): 3-9 % slower on
    // every product.  The counted waits sit in the read segments of phases 1 and 3 either way; what is younger than the group
    // they retire differs.
#ifdef BVC_G8_DMA_LATE
    constexpr bool LATE = true;
#else
    constexpr bool LATE = false;
#endif
    // row epilogues (EC 4 / 5), bytes past the two staging slots: per-column parameters (bias | gamma | beta, or gamma), 4 KiB of
    // row-statistics exchange [128 rows][4 wave columns][2], and for EC 5 the wave-private dgamma / dbeta accumulators [8][2][96]
    constexpr int PAR_BYTES = EC == 4 ? 3 * BN * 4 : EC == 5 ? BN * 4 : 0, STAT_OFF = PAR_BYTES, COL_OFF = STAT_OFF + 4096;
    constexpr int CPR = BN == 384 ? 8 : WN / 8;   // epilogue geometry: 8-column chunks per wave-tile row (unused by the 384-wide tile)
    constexpr int NSIDE = TM * (16 / (64 / CPR));  // 16-byte chunks (= bf16 store instructions per output) per lane and unit
    static_assert(PT + 2 * NSIDE < 64, "vmcnt is a 6-bit counter");
    constexpr int WP = AT ? PT : n3c + n0c + n1c;                                        // prologue: K tile 0's phase-0 operands
    constexpr int W1 = LATE ? n0c + n1c + n2c : PT;
    constexpr int W3 = LATE ? (AT ? n2c + n3c + n0c : n3c + n0c) : WP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // XCD x owns a contiguous run of unit ids; its gridDim.x / 8 workgroups take them round robin
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int xq = total_units >> 3, xr = total_units & 7;
    const int x_lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int x_hi = x_lo + xq + (xcd < xr ? 1 : 0);
    int uid = x_lo + slot_id, u_step = nslots, u_hi = x_hi;
    if (EC == 2 && g.bal_units > 0) {
        // balanced walk: the XCD's share of the bal_units full-length units goes one to a workgroup (total_units = bal_units here);
        // the workgroups left over take consecutive tails, [J T / I, (J + 1) T / I) for the J-th of the I idle workgroups
        // (numbered XCD after XCD, so that an XCD's tails are neighbouring tiles)
        u_step = 1;
        if (uid < x_hi) {
            u_hi = uid + 1;
        } else {
            const int J = xcd * nslots - x_lo + (uid - x_hi), I = 8 * nslots - g.bal_units;
            uid = g.bal_units + (J * g.bal_tiles) / I;
            u_hi = g.bal_units + ((J + 1) * g.bal_tiles) / I;
        }
    }
    if (uid >= u_hi) {             // uniform per workgroup, before any barrier
        if constexpr (EC == 5) {   // every workgroup of the grid owns one row of column partials
            float* pr = g.prob[0].ln_part + (size_t)blockIdx.x * 2 * BN;
            for (int i = tid; i < 2 * BN; i += 512) pr[i] = 0.f;
        }
        return;
    }
    if constexpr (EC >= 4) {
        // Row epilogues move 490 - 690 KB per unit through HBM while a unit's K loop needs only its A rows: with every workgroup
        // starting together the epilogues of a round fall together - all CUs queue on HBM, then all CUs compute and leave it idle
        // (measured: unit time = K loop + epilogue at the fair share of 5.5 TB/s, profiles/r05_d_*).  The workgroups of an XCD that
        // own one unit FEWER than the busiest ones (the last round is partial) therefore start late by a graded fraction of one unit
        // time: the launch ends no later, and the epilogues are spread over the cycle.  When all own the same number, half a unit.
        const int cnt = x_hi - x_lo, r = cnt % nslots, nmax = (cnt + nslots - 1) / nslots;
        int num = 0, den = 1;
        if (r > 0 && slot_id >= r) { num = slot_id - r + 1; den = nslots - r + 1; }
        else if (r == 0 && nmax >= 4) { num = slot_id; den = 2 * nslots; }
        const int quanta = __builtin_amdgcn_readfirstlane((int)((long long)g.stagger * num / den));
        for (int i = 0; i < quanta / 127; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // experiments build (BVC_GEMM_DEBUG = 1024 + (n << 12)): every other workgroup of an XCD starts n x 3.4 us late.  All workgroups
    // walk units of the same length from the same start, so their epilogues - the only phase with store / side-input traffic - hit
    // the memory system together; a start offset persists for the whole launch and puts one half's epilogues under the other half's K loops.
    if (BVC_DBG(g, 1024) && (slot_id & 1)) {
        for (int i = 0; i < ((g.dbg >> 12) & 63); ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ------------------------------------------------------------------ the staging cursor (runs ahead of the compute)
    Unit su;
    decode_unit<BM, BN>(g, uid, su);
    int s_uid = uid, s_kt = 0;
    bool s_valid = true;
    __amdgpu_buffer_rsrc_t s_ra, s_rb;
    uint32_t s_la, s_lb, s_baseA, s_baseB, s_strideA, s_strideB;   // per-lane source offsets; per-K-tile bases; per-piece strides
    auto s_problem = [&]() {
        const GemmProblem& p = g.prob[su.pi];
        s_ra = make_rsrc(p.A, p.a_bytes);
        s_rb = make_rsrc(p.B, p.b_bytes);
        if constexpr (!AT) {
            const int r = 8 * wave + (lane >> 3);
            s_la = (uint32_t)((r * p.lda + (((lane & 7) ^ swz_rows(r)) << 3)) * 2);
            s_strideA = (uint32_t)(64 * p.lda * 2);
        } else {
            const int kr = 4 * wave + (lane >> 4);
            s_la = (uint32_t)((kr * p.lda + (((lane & 15) ^ swz_tr<128>(kr)) << 3)) * 2);
            s_strideA = (uint32_t)(32 * p.lda * 2);
        }
        if constexpr (!BT) {
            const int r = 8 * wave + (lane >> 3);
            s_lb = (uint32_t)((r * p.ldb + (((lane & 7) ^ swz_rows(r)) << 3)) * 2);
            s_strideB = (uint32_t)(64 * p.ldb * 2);
        } else {
            const int kr = 4 * wave + (lane >> 4);
            s_lb = (uint32_t)((kr * p.ldb + (((lane & 15) ^ swz_tr<128>(kr)) << 3)) * 2);
            s_strideB = (uint32_t)(32 * p.ldb * 2);
        }
    };
    auto s_ktile = [&]() {       // bases of the cursor's K tile
        const GemmProblem& p = g.prob[su.pi];
        const int kt = su.kt0 + s_kt;
        const bool ok = s_valid && kt * 64 < p.K;
        const int k0 = kt * 64;
        s_baseA = !ok ? kInvalidBase : AT ? (uint32_t)((k0 * p.lda + su.m0) * 2) : (uint32_t)((su.m0 * p.lda + k0) * 2);
        s_baseB = !ok ? kInvalidBase : BT ? (uint32_t)((k0 * p.ldb + su.n0) * 2) : (uint32_t)((su.n0 * p.ldb + k0) * 2);
    };
    auto s_advance = [&]() {
        if (++s_kt == su.nkt) {
            s_kt = 0;
            s_uid += u_step;
            if (s_uid < u_hi) {
                const int old = su.pi;
                decode_unit<BM, BN>(g, s_uid, su);
                if (su.pi != old) s_problem();
            } else {
                s_valid = false;
            }
        }
        s_ktile();
    };
    // piece j of an operand tile -> its 8 KiB of LDS at region + j * 8192, this wave's 1 KiB at + wave * 1024
    //   k-contiguous operand: rows 64 j .. 64 j + 63;   transposed operand: half h = j >> 1 (128 columns), k rows 32 (j & 1) ..
    const uint32_t lds_base = (uint32_t)(size_t)((AS3 char*)smem);
    const uint32_t lds0 = lds_base + (uint32_t)wave * 1024u;   // this wave's 1 KiB of piece 0 of slot 0
    // lane constants of the transposed fragment reads (tr_frag)
    const uint32_t tr_base = (uint32_t)((8 * (lane >> 4) + ((lane >> 2) & 3)) * 256 + ((lane & 3) >> 1) * 16 + (lane & 1) * 8);
    const uint32_t tr_swz = (uint32_t)swz_tr<128>(8 * (lane >> 4) + ((lane >> 2) & 3)) << 4;
    // fragment j of this wave's transposed B columns (wn WN + 16 j): which 128-column half, and the column in address bits.  A 64- or
    // 32-wide wave tile sits inside one half; the 96-wide one of the 384-column tile straddles (uniform arithmetic, scalar registers)
    auto b_region = [&](int j) -> uint32_t { return (uint32_t)(A_BYTES + ((wn * WN + 16 * j) >> 7) * 16384); };
    auto b_chunk16 = [&](int j) -> uint32_t { return (uint32_t)(2 * ((wn * WN + 16 * j) & 127)); };
    // this wave row's A rows in a transposed A tile: half wm of a 256-row tile, rows 64 wm .. of the single half of a 128-row one
    const uint32_t a_region = BM == 256 ? (uint32_t)wm * 16384u : 0u;
    const uint32_t a_chunk0 = BM == 256 ? 0u : (uint32_t)wm * 128u;
    auto load_a = [&](char* region, int j) {
        const uint32_t off = s_la + s_baseA + (AT ? (uint32_t)(j & 1) * s_strideA + (uint32_t)(j >> 1) * 256u : (uint32_t)j * s_strideA);
        glds16(s_ra, off, lds0 + (uint32_t)(region - smem) + (uint32_t)j * 8192u);
    };
    auto load_b = [&](char* region, int j) {
        const uint32_t off = s_lb + s_baseB + (BT ? (uint32_t)(j & 1) * s_strideB + (uint32_t)(j >> 1) * 256u : (uint32_t)j * s_strideB);
        glds16(s_rb, off, lds0 + (uint32_t)(region - smem) + (uint32_t)j * 8192u);
    };
    // load K (0 / 1) of group GI of the cursor's K tile into LDS slot `buf`; a group has at most two loads; after the second load
    // of g3 the cursor moves to the next K tile of the stream
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto stage_one = [&](auto gi_, auto k_, char* buf) {
        constexpr int GI = decltype(gi_)::value, K = decltype(k_)::value;
        char* ra_ = buf;
        char* rb_ = buf + A_BYTES;
        if constexpr (!AT) {
            if constexpr (GI == 0) { if constexpr (K < n0c) load_b(rb_, K); }
            else if constexpr (GI == 1) { if constexpr (K < n1c) load_b(rb_, n0c + K); }
            else if constexpr (GI == 2) { if constexpr (K < n2c) load_a(ra_, BM == 256 ? (K == 0 ? 0 : 2) : 0); }
            else { if constexpr (K < n3c) load_a(ra_, BM == 256 ? (K == 0 ? 1 : 3) : 1); }
        } else {       // k rows 0-31 of every B half, of every A half, then k rows 32-63 likewise (piece 2 h + (k >= 32) of half h)
            if constexpr (GI == 0) { if constexpr (K < NBH) load_b(rb_, 2 * K); }
            else if constexpr (GI == 1) { if constexpr (K < NAH) load_a(ra_, 2 * K); }
            else if constexpr (GI == 2) { if constexpr (K < NBH) load_b(rb_, 2 * K + 1); }
            else { if constexpr (K < NAH) load_a(ra_, 2 * K + 1); }
        }
        if constexpr (GI == 3 && K == KMAX - 1) s_advance();
    };
    auto stage_group = [&](auto gi_, char* buf) {
        stage_one(gi_, I0{}, buf); stage_one(gi_, I1{}, buf);
        if constexpr (KMAX == 3) stage_one(gi_, I2{}, buf);
    };
    // the group phase q issues: (q + 2) & 3 of the cursor's K tile, into the other slot for q = 0, 1 and into this one for q = 2, 3
    auto stage_phase = [&](auto q_, auto k_, char* cur, char* oth) {
        constexpr int Q = decltype(q_)::value;
        if constexpr (Q == 0) stage_one(I2{}, k_, oth);
        else if constexpr (Q == 1) stage_one(I3{}, k_, oth);
        else if constexpr (Q == 2) stage_one(I0{}, k_, cur);
        else stage_one(I1{}, k_, cur);
    };

    char* const buf0 = smem;
    char* const buf1 = smem + TILE;
    // wave-private epilogue parking: 16 rows x WN f32 (the 384-wide tile parks 16 rows x 48 columns at a time: 24 KiB for the
    // eight waves next to 128 KiB of staging)
    constexpr int PARK = BN == 384 ? 16 * 48 * 4 : 16 * WN * 4;
    AS3 char* const wl = (AS3 char*)smem + 2 * TILE + wave * PARK;

    if constexpr (EC == 0) {
        // the bias vector of the (single) problem, zero padded to whole tiles, into the LDS the other classes park accumulators
        // in; read by every unit's epilogue.  Plain loads: hipcc waits for them right here, before any LDS-DMA is in flight.
        const GemmProblem& p0 = g.prob[0];
        AS3 float* lbias = (AS3 float*)((AS3 char*)smem + 2 * TILE);
        const int npad = ((p0.N + BN - 1) / BN) * BN;
        for (int i = tid; i < npad; i += 512) lbias[i] = (p0.bias && i < p0.N) ? p0.bias[i] : 0.f;
        __syncthreads();
    }
    if constexpr (EC == 4 || EC == 5) {
        // per-column parameters of the (single) problem into LDS; the column accumulators of EC 5 start at zero.  Plain loads,
        // waited for right here, before any LDS-DMA is in flight.
        const GemmProblem& p0 = g.prob[0];
        AS3 float* lpar = (AS3 float*)((AS3 char*)smem + 2 * TILE);
        for (int i = tid; i < BN; i += 512) {
            if constexpr (EC == 4) {
                lpar[i] = p0.bias ? p0.bias[i] : 0.f;
                lpar[BN + i] = p0.ln_gamma[i];
                lpar[2 * BN + i] = p0.ln_beta[i];
            } else {
                lpar[i] = p0.ln_gamma[i];
            }
        }
        if constexpr (EC == 5) {
            AS3 float* lcol = (AS3 float*)((AS3 char*)smem + 2 * TILE + COL_OFF);
            for (int i = tid; i < 8 * 2 * WN; i += 512) lcol[i] = 0.f;
        }
        __syncthreads();
    }
    // ------------------------------------------------------------------ prologue: K tile 0 and groups 0, 1 of K tile 1
    s_problem();
    s_ktile();
    stage_group(I0{}, buf0); stage_group(I1{}, buf0); stage_group(I2{}, buf0); stage_group(I3{}, buf0);
    stage_group(I0{}, buf1); stage_group(I1{}, buf1);
    wait_vmcnt<WP>();
    asm volatile("s_barrier" ::: "memory");
#ifndef BVC_G8_NO_STAGGER
    if (wm == 1) asm volatile("s_barrier" ::: "memory");     // the stagger: wave row 1 runs one barrier behind wave row 0
#endif

    Unit cu;
    decode_unit<BM, BN>(g, uid, cu);
    // First K tile after an epilogue: what its phase-1 wait has to allow for.  0: the stream ran on, the plain counted wait;
    // -1: the epilogue drained every load before its stores, no wait needed; n > 0: the epilogue issued n stores BEHIND the
    // prefetched loads without draining them (vmcnt is one in-order counter), so the wait counts them as younger operations.
    int after_epi = 0;


    while (true) {
        const GemmProblem& p = g.prob[cu.pi];
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float rsum[2] = {0.f, 0.f};
        const bool do_rowsum = EC == 2 && p.rowsum != nullptr && cu.n0 == 0;   // bias gradient: wave wn takes row tiles wn and 4 + wn

        // one K tile out of LDS slot `cur`; `oth` is the other slot
        auto ktile = [&](auto rs_, char* cur, char* oth) {
            constexpr bool RS = decltype(rs_)::value;       // weight-gradient instantiation: may fuse the bias gradient
            const char* la = cur;
            const char* lb = cur + A_BYTES;
            if constexpr (!AT) {
                bf16x8 bfr[2][TN];
                auto phase = [&](auto q_) {
                    constexpr int q = decltype(q_)::value;
                    __builtin_amdgcn_sched_barrier(0);
#ifdef BVC_G8_DMA_FIRST      // measured alternative (profiles/r03_f_gemm8_loop_variants.txt): the phase's LDS-DMA pieces ahead of its fragment reads
                    if constexpr (!LATE) { stage_phase(q_, I0{}, cur, oth); stage_phase(q_, I1{}, cur, oth); if constexpr (KMAX == 3) stage_phase(q_, I2{}, cur, oth); }
#endif
                    uint32_t sw = tr_swz;
                    if constexpr (BT && q == 0) asm volatile("" : "+v"(sw));
                    if constexpr (q == 0) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                bfr[ks][j] = BT ? tr_frag(lds_base + (uint32_t)(cur - smem) + b_region(j), tr_base, sw, b_chunk16(j), ks)
                                                : read_frag<BN, false>(lb, wn * WN + 16 * j, ks, lane);
                    }
                    bf16x8 af[2][RT];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int ii = 0; ii < RT; ++ii)
                            af[ks][ii] = read_frag<256, false>(la, BM == 256 ? wm * WMR + 16 * (2 * q + ii) : 32 * q + 16 * wm, ks, lane);
#ifndef BVC_G8_DMA_FIRST
                    if constexpr (!LATE) { stage_phase(q_, I0{}, cur, oth); stage_phase(q_, I1{}, cur, oth); if constexpr (KMAX == 3) stage_phase(q_, I2{}, cur, oth); }
#endif
                    if constexpr (q == 1) {
                        if (after_epi == 0) wait_vmcnt<W1>();
                        else if (after_epi == NSIDE) wait_vmcnt<W1 + NSIDE>();
                        else if (after_epi == 2 * NSIDE) wait_vmcnt<W1 + 2 * NSIDE>();
                    }
                    if constexpr (q == 3) wait_vmcnt<W3>();
                    asm volatile("s_barrier" ::: "memory");   // fragment reads are builtins: hipcc places counted lgkmcnt waits in front of the MFMAs that use them
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
                    constexpr int NM = 2 * RT * TN;       // MFMAs of the phase: the two LDS-DMA loads go after the first and the second quarter
#pragma unroll
                    for (int t = 0; t < NM; ++t) {
                        const int ks = t / (RT * TN), ii = (t / TN) % RT, j = t % TN;
                        acc[RT * q + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][ii], acc[RT * q + ii][j], 0, 0, 0);
                        if constexpr (LATE) {
                            if (t == NM / 4 - 1) stage_phase(q_, I0{}, cur, oth);
                            if (t == NM / 2 - 1) stage_phase(q_, I1{}, cur, oth);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_barrier" ::: "memory");
                };
                phase(I0{}); phase(I1{}); phase(I2{}); phase(I3{});
            } else {
                bf16x8 bfr[TN];      // the B fragments of one k half: read in the half's first phase, kept for its second
                auto phase = [&](auto q_) {
                    constexpr int q = decltype(q_)::value;
                    constexpr int ks = q >> 1, mh = q & 1;
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 af[TMH];
                    uint32_t sw = tr_swz;
                    asm volatile("" : "+v"(sw));
                    const uint32_t slot = lds_base + (uint32_t)(cur - smem);
                    if constexpr (mh == 0) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) bfr[j] = tr_frag(slot + b_region(j), tr_base, sw, b_chunk16(j), ks);
                    }
#pragma unroll
                    for (int ii = 0; ii < TMH; ++ii) af[ii] = tr_frag(slot + a_region, tr_base, sw, a_chunk0 + 32u * (TMH * mh + ii), ks);
                    if constexpr (!LATE) {
                        stage_phase(q_, I0{}, cur, oth); stage_phase(q_, I1{}, cur, oth);
                        if constexpr (KMAX == 3) stage_phase(q_, I2{}, cur, oth);
                    }
                    if constexpr (q == 1) {
                        if (after_epi == 0) wait_vmcnt<W1>();
                        else if (after_epi == NSIDE) wait_vmcnt<W1 + NSIDE>();
                        else if (after_epi == 2 * NSIDE) wait_vmcnt<W1 + 2 * NSIDE>();
                    }
                    if constexpr (q == 3) wait_vmcnt<W3>();
                    asm volatile("s_barrier" ::: "memory");   // fragment reads are builtins: hipcc places counted lgkmcnt waits in front of the MFMAs that use them
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
                    constexpr int NM = TMH * TN;
#pragma unroll
                    for (int t = 0; t < NM; ++t) {
                        const int ii = t / TN, j = t % TN;
                        acc[TMH * mh + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[ii], acc[TMH * mh + ii][j], 0, 0, 0);
                        if constexpr (LATE) {
                            if (t == NM / 4 - 1) stage_phase(q_, I0{}, cur, oth);
                            if (t == NM / 2 - 1) stage_phase(q_, I1{}, cur, oth);
                        }
                    }
                    if (RS && do_rowsum && wn < TMH) {
                        // bias gradient db[m] = sum_k A(m, k): wave column wn sums the fragment of row tile TMH mh + wn it has in
                        // registers anyway (16 vector instructions per phase, no extra MFMA, no extra accumulator tile: the
                        // all-ones-operand MFMA this replaces cost ~45 live registers and made the 256 x 256 instantiation
                        // spill).  Lane l holds 8 k values of row l & 15; the k groups are folded by two shuffles at the end.
                        bf16x8 a;
                        if constexpr (TMH == 4) {
                            switch (wn) {
                                case 0: a = af[0]; break;
                                case 1: a = af[1]; break;
                                case 2: a = af[2]; break;
                                default: a = af[3]; break;
                            }
                        } else {
                            a = wn == 0 ? af[0] : af[1];
                        }
                        union { bf16x8 v; uint32_t u[4]; } w;
                        w.v = a;
                        float t0 = 0.f, t1 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            t0 += __uint_as_float(w.u[e] << 16);
                            t1 += __uint_as_float(w.u[e] & 0xffff0000u);
                        }
                        rsum[mh] += t0 + t1;
                    }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_barrier" ::: "memory");
                };
                phase(I0{}); phase(I1{}); phase(I2{}); phase(I3{});
            }
            after_epi = 0;
        };

        for (int kt = 0; kt < cu.nkt; kt += 2) {
            ktile(std::bool_constant<EC == 2>{}, buf0, buf1);
            ktile(std::bool_constant<EC == 2>{}, buf1, buf0);
        }

        // ------------------------------------------------------------------ epilogue of this unit
        // Branch-free: every side-input load and every store goes through a buffer descriptor with an out-of-range offset
        // for rows / columns past the matrix, and every side input of the unit is in registers BEFORE its first store.
        // (With per-chunk branches hipcc's wait-count pass put an `s_waitcnt vmcnt(0)` behind every store - one store round
        // trip per chunk, ~14 us per 256 x 256 tile.)
        const int m0 = cu.m0, n0 = cu.n0;
        const int epi = p.epi, Mrows = p.M, Ncols = p.N, ldc = p.ldc;
        const float alpha = p.alpha_dev ? p.alpha * p.alpha_dev[0] : p.alpha;
        constexpr int UNITS = WN / 4;                 // 16-B units per parked row
        constexpr int RPU = 64 / CPR;                 // rows covered by the 64 lanes in one pass
        constexpr int U = 16 / RPU;                   // passes per 16-row round
        static_assert(NSIDE == TM * U, "epilogue geometry");
        int nstores = 0;
        constexpr uint32_t kDrop = 0xFFFFFFF0u;       // >= every descriptor's extent: the access is dropped / reads 0
        auto park = [&](int i) {
            const int row = lane & 15;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int unit = (4 * j + (lane >> 4)) ^ (row & (UNITS - 1));
                *reinterpret_cast<AS3 f32x4*>(wl + row * (WN * 4) + unit * 16) = acc[i][j];
            }
        };
        const bool atomic = p.split_k > 1 || g.accum != 0;
        if constexpr (EC == 0) {
            // bf16 outputs without side inputs, entirely in registers: v = alpha acc + bias in the MFMA layout (lane = row l & 15,
            // 4 consecutive columns at 4 (l >> 4) of each 16 x 16 tile), bias out of the LDS copy made at kernel entry (no
            // vector-memory load, so NOTHING has to be drained: the next unit's prefetch stays in flight and the stores queue
            // behind it), packed to bf16, and v_permlane16_swap between the tiles of a column pair turns the 8-byte pieces of
            // lanes l / l + 16 into 16 bytes per lane: every store instruction writes 16 rows x 64 contiguous bytes.
            // (experiments build, BVC_GEMM_DEBUG bit 1: zero-record descriptors - every store is dropped by the range check while
            //  the instruction stream, the counters and the waits stay: prices the stores)
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, BVC_DBG(g, 1) ? 0u : kDrop);
            const __amdgpu_buffer_rsrc_t rc2 = make_rsrc(p.C2 ? p.C2 : p.C, BVC_DBG(g, 1) ? 0u : kDrop);
            const AS3 float* lbias = (const AS3 float*)((AS3 char*)smem + 2 * TILE);
            const int q4 = lane >> 4;
            f32x4 bj[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + wn * WN + 16 * j + 4 * q4;            // < N rounded up to the tile: inside the LDS copy
                bj[j] = *reinterpret_cast<const AS3 f32x4*>(lbias + nb);
            }
            const bool two_out = epi == EPI_GELU;
            const bool relu = epi == EPI_RELU;
            // this lane's 8 columns after the swap: tile 2 jp + (q4 & 1), columns 8 (q4 >> 1) .. + 7
            const int ncol = n0 + wn * WN + 16 * (q4 & 1) + 8 * (q4 >> 1);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * WMR + 16 * i + (lane & 15);
#pragma unroll
                for (int jp = 0; jp < TN / 2; ++jp) {
                    float va[4], vb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        va[e] = acc[i][2 * jp][e] * alpha + bj[2 * jp][e];
                        vb[e] = acc[i][2 * jp + 1][e] * alpha + bj[2 * jp + 1][e];
                    }
                    if (relu) {      // (a uniform branch: the GELU path does not carry the max / select pairs)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], 0.f); vb[e] = fmaxf(vb[e], 0.f); }
                    }
                    const int n = ncol + 32 * jp;
                    const uint32_t o = (m < Mrows && n < Ncols) ? (uint32_t)(((size_t)m * ldc + n) * 2) : kDrop;
                    auto emit = [&](__amdgpu_buffer_rsrc_t r) {
                        uint32_t a0 = pack2bf(va[0], va[1]), a1 = pack2bf(va[2], va[3]);
                        uint32_t b0 = pack2bf(vb[0], vb[1]), b1 = pack2bf(vb[2], vb[3]);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                        // even lane rows: own tile-2jp columns, then lane + 16's; odd lane rows: lane - 16's tile-(2jp+1) columns, then own
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, r, o, 0, BVC_G8_ST_AUX);
                    };
                    if (two_out) {      // EPI_GELU: C <- gelu'(pre), C2 <- gelu(pre)
                        float vv[8], gg[8];      // the four pairs of both tiles side by side
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vv[e] = va[e]; vv[4 + e] = vb[e]; }
                        gelu_split(vv, gg);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = vv[e]; vb[e] = vv[4 + e]; }
                        emit(rc);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = gg[e]; vb[e] = gg[4 + e]; }
                        emit(rc2);
                    } else {
                        emit(rc);
                    }
                }
            }
            nstores = two_out ? 2 * NSIDE : NSIDE;
        } else if constexpr (EC == 4) {
            // ---- residual + LayerNorm forward on complete rows (N == ldc == BN).  Lane l holds, of row tile i, row 32 i + 16 wm + (l & 15)
            // and the 4 columns wn 96 + 16 j + 4 (l >> 4) .. + 3 of each of the 6 column tiles j.  Descriptors carry the exact extent
            // M x 384 elements: a row past M starts past it and every access to it is dropped (whole rows, no column test).
            // Row tile by row tile, the residual rows of tile i + 1 in flight while tile i is finished (24 registers each: all four at
            // once next to the accumulators made hipcc spill lane constants of the K loop into scratch).
            const uint32_t ext = (uint32_t)Mrows * (uint32_t)(BN * 4);
            const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.resid, ext), rc = make_rsrc(p.C, ext), rc2 = make_rsrc(p.C2, ext / 2);
            const __amdgpu_buffer_rsrc_t rmu = make_rsrc(p.ln_mean, (uint32_t)Mrows * 4u), rrs = make_rsrc(p.ln_rstd, (uint32_t)Mrows * 4u);
            const AS3 float* lpar = (const AS3 float*)((AS3 char*)smem + 2 * TILE);
            AS3 float* lstat = (AS3 float*)((AS3 char*)smem + 2 * TILE + STAT_OFF);
            const int q4 = lane >> 4, r16 = lane & 15;
            const int cw = wn * WN + 4 * q4;
            const float eps = p.ln_eps;
            const uint32_t obase = (uint32_t)(m0 + 16 * wm + r16) * (uint32_t)(BN * 4) + (uint32_t)cw * 4u;     // row tile i: + i * 32 rows
            f32x4 rs[2][TN];
            auto issue = [&](auto i_) {
                constexpr int i = decltype(i_)::value;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    rs[i & 1][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, obase + (uint32_t)(i * 32 * BN * 4 + 64 * j), 0, 0));
            };
            auto tile = [&](auto i_) {
                constexpr int i = decltype(i_)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i + 1 < TM) issue(std::integral_constant<int, i + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 bj = *reinterpret_cast<const AS3 f32x4*>(lpar + cw + 16 * j);
                    acc[i][j] = (acc[i][j] * alpha + bj) + rs[i & 1][j];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rc, obase + (uint32_t)(i * 32 * BN * 4 + 64 * j), 0, BVC_G8_ROW_ST_AUX);
                    sum += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
                }
                // statistics of the wave's 96 columns of this row: sum, and the squares about the wave's own mean (Chan's form: the four
                // wave columns combine without cancellation)
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float mw = sum * (1.f / WN);
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = acc[i][j][e] - mw; sq += d * d; }
                sq += __shfl_xor(sq, 16, 64);
                sq += __shfl_xor(sq, 32, 64);
                if (q4 == 0) *reinterpret_cast<AS3 f32x2*>(lstat + ((wm * 64 + 16 * i + r16) * 4 + wn) * 2) = f32x2{sum, sq};
            };
            issue(I0{});
            wait_vmcnt<0>();      // one drain: the first residual rows and the next unit's prefetched K tiles (see the f32 class below)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(rs[0][j]));
            tile(I0{}); tile(I1{}); tile(I2{}); tile(I3{});
            // the four waves of a wave row run in step (the stagger is between wave rows): one barrier orders their exchange
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                asm volatile("" ::: "memory");     // (keeps the parameter reads below per row tile: hoisted, they are 48 live registers)
                const AS3 f32x4* sp = reinterpret_cast<const AS3 f32x4*>(lstat + (wm * 64 + 16 * i + r16) * 8);
                const f32x4 a = sp[0], b = sp[1];          // {sum, sq} of wave columns 0, 1 | 2, 3
                const float mean = ((a[0] + a[2]) + (b[0] + b[2])) * (1.f / BN);
                const float d0 = a[0] * (1.f / WN) - mean, d1 = a[2] * (1.f / WN) - mean, d2 = b[0] * (1.f / WN) - mean, d3 = b[2] * (1.f / WN) - mean;
                const float m2 = ((a[1] + a[3]) + (b[1] + b[3])) + (float)WN * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                const float rstd = rsqrtf(m2 * (1.f / BN) + eps);
                const int m = m0 + 32 * i + 16 * wm + r16;
                if (wn == 0 && q4 == 0) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mean), rmu, (uint32_t)m * 4u, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rstd), rrs, (uint32_t)m * 4u, 0, 0);
                }
                // bf16 LayerNorm output, 16 bytes per lane after the v_permlane16_swap of a column-tile pair (as the bf16 class above)
                const uint32_t o2 = ((uint32_t)m * (uint32_t)BN + (uint32_t)(wn * WN + 16 * (q4 & 1) + 8 * (q4 >> 1))) * 2u;
#pragma unroll
                for (int jp = 0; jp < TN / 2; ++jp) {
                    const f32x4 ga = *reinterpret_cast<const AS3 f32x4*>(lpar + BN + cw + 32 * jp), gb = *reinterpret_cast<const AS3 f32x4*>(lpar + BN + cw + 32 * jp + 16);
                    const f32x4 ba = *reinterpret_cast<const AS3 f32x4*>(lpar + 2 * BN + cw + 32 * jp), bb = *reinterpret_cast<const AS3 f32x4*>(lpar + 2 * BN + cw + 32 * jp + 16);
                    const f32x4 ya = ((acc[i][2 * jp] - mean) * rstd) * ga + ba;
                    const f32x4 yb = ((acc[i][2 * jp + 1] - mean) * rstd) * gb + bb;
                    const uint32_t a0 = pack2bf(ya[0], ya[1]), a1 = pack2bf(ya[2], ya[3]), b0 = pack2bf(yb[0], yb[1]), b1 = pack2bf(yb[2], yb[3]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rc2, o2 + 64u * jp, 0, BVC_G8_ROW_ST_AUX);
                }
            }
        } else if constexpr (EC == 5) {
            // ---- LayerNorm backward on complete rows: acc = g = d/d(LayerNorm output).  Row tile by row tile; the LayerNorm input rows and
            // the incoming residual gradient of tile i + 1 are in flight while tile i is reduced, exchanged and written.
            const uint32_t ext = (uint32_t)Mrows * (uint32_t)(BN * 4);
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.ln_x, ext), rd = make_rsrc(p.C, ext), rc2 = make_rsrc(p.C2, ext / 2);
            const __amdgpu_buffer_rsrc_t rmu = make_rsrc(p.ln_mean, (uint32_t)Mrows * 4u), rrs = make_rsrc(p.ln_rstd, (uint32_t)Mrows * 4u);
            const AS3 float* lpar = (const AS3 float*)((AS3 char*)smem + 2 * TILE);
            AS3 float* lstat = (AS3 float*)((AS3 char*)smem + 2 * TILE + STAT_OFF);
            AS3 float* lcol = (AS3 float*)((AS3 char*)smem + 2 * TILE + COL_OFF) + wave * 2 * WN;     // this wave's [dgamma | dbeta][96]
            const int q4 = lane >> 4, r16 = lane & 15;
            const int cw = wn * WN + 4 * q4;
            const int mrow = m0 + 16 * wm + r16;                                                              // row tile i: + 32 i
            const uint32_t obase = (uint32_t)mrow * (uint32_t)(BN * 4) + (uint32_t)cw * 4u;
            const uint32_t o2base = ((uint32_t)mrow * (uint32_t)BN + (uint32_t)(wn * WN + 16 * (q4 & 1) + 8 * (q4 >> 1))) * 2u;
            // (the residual-gradient rows of a tile are fetched at the START of that tile - they are consumed after its exchange - and only the
            //  LayerNorm input rows one tile ahead: with both one tile ahead hipcc spilled lane constants of the K loop into scratch)
            f32x4 xr[2][TN], dr[TN];
            float mu[2], rsd[2];
            auto issue = [&](auto i_) {
                constexpr int i = decltype(i_)::value;
                mu[i & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmu, (uint32_t)(mrow + 32 * i) * 4u, 0, 0));
                rsd[i & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, (uint32_t)(mrow + 32 * i) * 4u, 0, 0));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    xr[i & 1][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, obase + (uint32_t)(i * 32 * BN * 4 + 64 * j), 0, 0));
            };
            auto tile = [&](auto i_) {
                constexpr int i = decltype(i_)::value, b = i & 1;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    dr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, obase + (uint32_t)(i * 32 * BN * 4 + 64 * j), 0, 0));
                if constexpr (i + 1 < TM) issue(std::integral_constant<int, i + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 gam = *reinterpret_cast<const AS3 f32x4*>(lpar + cw + 16 * j);
                    const f32x4 gv = acc[i][j] * alpha;
                    const f32x4 xh = (xr[b][j] - mu[b]) * rsd[b];
                    const f32x4 gg = gv * gam;
                    f32x4 tg = gv * xh, tb = gv;
                    s1 += (gg[0] + gg[1]) + (gg[2] + gg[3]);
                    const f32x4 gx = gg * xh;
                    s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
                    xr[b][j] = xh;
                    acc[i][j] = gg;
                    // column sums over the 16 rows the lanes of a DPP row hold (rotations: every lane ends with the full sum), then the
                    // four lanes with (l & 15) == 0 add their 4 columns into the wave's accumulators in LDS
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        tg[e] = row16_sum(tg[e]);
                        tb[e] = row16_sum(tb[e]);
                    }
                    if (r16 == 0) {
                        AS3 f32x4* cg = reinterpret_cast<AS3 f32x4*>(lcol + 16 * j + 4 * q4);
                        AS3 f32x4* cb = reinterpret_cast<AS3 f32x4*>(lcol + WN + 16 * j + 4 * q4);
                        *cg = *cg + tg;
                        *cb = *cb + tb;
                    }
                }
                s1 += __shfl_xor(s1, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (q4 == 0) *reinterpret_cast<AS3 f32x2*>(lstat + ((wm * 64 + 16 * i + r16) * 4 + wn) * 2) = f32x2{s1, s2};
                // the four waves of a wave row run in step (the stagger is between wave rows): one barrier orders their exchange
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const AS3 f32x4* sp = reinterpret_cast<const AS3 f32x4*>(lstat + (wm * 64 + 16 * i + r16) * 8);
                const f32x4 sa = sp[0], sb = sp[1];
                const float c1 = ((sa[0] + sa[2]) + (sb[0] + sb[2])) * (1.f / BN), c2 = ((sa[1] + sa[3]) + (sb[1] + sb[3])) * (1.f / BN);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = ((acc[i][j] - c1) - xr[b][j] * c2) * rsd[b] + dr[j];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rd, obase + (uint32_t)(i * 32 * BN * 4 + 64 * j), 0, BVC_G8_ROW_ST_AUX);
                }
#pragma unroll
                for (int jp = 0; jp < TN / 2; ++jp) {
                    const f32x4 ya = acc[i][2 * jp], yb = acc[i][2 * jp + 1];
                    const uint32_t a0 = pack2bf(ya[0], ya[1]), a1 = pack2bf(ya[2], ya[3]), b0 = pack2bf(yb[0], yb[1]), b1 = pack2bf(yb[2], yb[3]);
                    const auto w0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto w1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rc2, o2base + (uint32_t)(i * 32 * BN * 2 + 64 * jp), 0, BVC_G8_ROW_ST_AUX);
                }
            };
            issue(I0{});
            wait_vmcnt<0>();         // the first row tile's rows and the next unit's prefetched K tiles
            asm volatile("" : "+v"(mu[0]), "+v"(rsd[0]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(xr[0][j]));
            tile(I0{}); tile(I1{}); tile(I2{}); tile(I3{});
        } else if constexpr (BN == 384) {
            // 128 x 384 weight-gradient tile: the wave's 64 x 96 outputs leave through 16 rows x 48 columns of parking at a time (two
            // passes per row tile), one dword per lane - runs of 64 consecutive floats of the row-major [16][48] image, i.e. 64-byte
            // aligned pieces of one or two output rows per wave-instruction; f32 atomics when K is split (the memory-side atomic units
            // take a wave-instruction as four 64-byte requests whatever the rows), plain stores otherwise.  Once per ~900 K tiles.
            const __amdgpu_buffer_rsrc_t rca = make_rsrc(p.C, kDrop);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int prow = lane & 15;
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj)
                        *reinterpret_cast<AS3 f32x4*>(wl + prow * 192 + (4 * jj + (lane >> 4)) * 16) = acc[i][3 * ps + jj];
#pragma unroll
                    for (int t = 0; t < 16 * 48 / 64; ++t) {
                        const int idx = t * 64 + lane;
                        const int row = idx / 48, col = idx % 48;
                        const int m = m0 + wm * WMR + 16 * i + row, n = n0 + wn * WN + 48 * ps + col;
                        const float v = *reinterpret_cast<const AS3 float*>(wl + row * 192 + col * 4) * alpha;
                        const uint32_t o = (m < Mrows && n < Ncols) ? (uint32_t)(((size_t)m * ldc + n) * 4) : kDrop;
                        if (atomic) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rca, o, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rca, o, 0, 0);
                    }
                }
            }
        } else if (EC == 2 && atomic) {
            // split-K: f32 atomics, one dword per lane, whole contiguous rows per wave-instruction (256 B / two 128-B rows: the shape
            // the memory-side atomic units take at full rate), through a buffer descriptor (32-bit offsets, dropped when out of range)
            const __amdgpu_buffer_rsrc_t rca = make_rsrc(p.C, kDrop);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                park(i);
#pragma unroll
                for (int t = 0; t < 16 * WN / 64; ++t) {
                    const int idx = t * 64 + lane;
                    const int row = idx / WN, col = idx % WN;
                    const int m = m0 + wm * WMR + 16 * i + row, n = n0 + wn * WN + col;
                    const int unit = (col >> 2) ^ (row & (UNITS - 1));
                    const float v = *reinterpret_cast<const AS3 float*>(wl + row * (WN * 4) + unit * 16 + (col & 3) * 4) * alpha;
                    const uint32_t o = (m < Mrows && n < Ncols) ? (uint32_t)(((size_t)m * ldc + n) * 4) : kDrop;
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rca, o, 0, 0);
                }
            }
        } else {
            const int cc = lane % CPR;
            const int n = n0 + wn * WN + cc * 8;
            const bool ncol_ok = n < Ncols;
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, kDrop);
            const __amdgpu_buffer_rsrc_t rc2 = make_rsrc((EC == 0 || EC == 1) && p.C2 ? p.C2 : p.C, kDrop);
            const bool have_c2 = p.C2 != nullptr;
            // element offset of (row m, this lane's 8 columns) or "dropped"; esz = bytes per element of the addressed tensor
            auto offs = [&](int m, int ld, int esz) -> uint32_t {
                return (m < Mrows && ncol_ok) ? (uint32_t)(((size_t)m * ld + n) * esz) : kDrop;
            };
            f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EC != 2) {
                const __amdgpu_buffer_rsrc_t rbias = make_rsrc(p.bias, p.bias ? kDrop : 0u);
                const uint32_t ob = ncol_ok ? (uint32_t)n * 4u : kDrop;
                bias0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, ob, 0, 0));
                bias1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, ob, 16, 0));
            }
            // Side inputs of the whole unit would be 128 registers for the f32 class on 256 x 256 tiles: that combination runs in
            // NP = 4 passes of 2 row rounds each (loads, drain, stores, loads, ...; two passes still spilled inside the K loop);
            // every later pass's wait also waits for the previous pass's stores (three store round trips per unit - acceptable
            // only for the long-K products it is selected for).
            constexpr int NP = (EC == 1 && BN == 256) ? 4 : 1;
            constexpr int NSP = NSIDE / NP, TMP = TM / NP;
            f32x4 side0[(EC == 1 || EC == 3) ? NSP : 1], side1[(EC == 1) ? NSP : 1];
            float sumsq = 0.f;
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
            if constexpr (EC == 3) {
                const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux, kDrop);
                const int ldaux = p.ldaux;
#pragma unroll
                for (int c = 0; c < NSP; ++c) {
                    const int m = m0 + wm * WMR + 16 * ((c + pass * NSP) / U) + ((((c + pass * NSP) % U) * 64 + lane) / CPR);
                    side0[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, offs(m, ldaux, 2), 0, 0));
                }
            } else if constexpr (EC == 1) {
                const bool side_f32 = epi == EPI_RESID || epi == EPI_POS || epi == EPI_E2D || epi == EPI_LOSS;
                const bool by_tok = epi == EPI_POS || epi == EPI_E2D;
                const void* sbase = epi == EPI_RESID ? (const void*)p.resid : epi == EPI_LOSS ? (const void*)p.labels : (const void*)p.pos;
                const __amdgpu_buffer_rsrc_t rside = make_rsrc(sbase, side_f32 ? kDrop : 0u);
                const __amdgpu_buffer_rsrc_t rtok = make_rsrc(p.rowtok, by_tok ? kDrop : 0u);
#pragma unroll
                for (int c = 0; c < NSP; ++c) {
                    const int m = m0 + wm * WMR + 16 * ((c + pass * NSP) / U) + ((((c + pass * NSP) % U) * 64 + lane) / CPR);
                    uint32_t o = offs(m, ldc, 4);
                    if (by_tok) {
                        const int tok = __builtin_amdgcn_raw_buffer_load_b32(rtok, m < Mrows ? (uint32_t)m * 4u : kDrop, 0, 0);
                        o = (m < Mrows && ncol_ok) ? (uint32_t)(((size_t)tok * Ncols + n) * 4) : kDrop;
                    }
                    side0[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rside, o, 0, 0));
                    side1[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rside, o, 16, 0));
                }
            }
            // one drain: the side inputs and the next unit's prefetched K tiles (the LDS-DMA is invisible to hipcc, so the
            // explicit wait stays; the empty statement makes hipcc wait for ITS loads here, once, and not behind every store)
            wait_vmcnt<0>();
            asm volatile("" : "+v"(bias0), "+v"(bias1));
            if constexpr (EC == 1 || EC == 3) {
#pragma unroll
                for (int c = 0; c < NSP; ++c) asm volatile("" : "+v"(side0[c]));
            }
            if constexpr (EC == 1) {
#pragma unroll
                for (int c = 0; c < NSP; ++c) asm volatile("" : "+v"(side1[c]));
            }
#pragma unroll
            for (int i = pass * TMP; i < (pass + 1) * TMP; ++i) {
                park(i);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = (u * 64 + lane) / CPR;
                    const int m = m0 + wm * WMR + 16 * i + row;
                    const f32x4 lo = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc) ^ (row & (UNITS - 1))) << 4));
                    const f32x4 hi = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc + 1) ^ (row & (UNITS - 1))) << 4));
                    const int c = U * i + u - pass * NSP;
                    const bool ok = m < Mrows && ncol_ok;
                    float v[8] = {lo[0] * alpha + bias0[0], lo[1] * alpha + bias0[1], lo[2] * alpha + bias0[2], lo[3] * alpha + bias0[3],
                                  hi[0] * alpha + bias1[0], hi[1] * alpha + bias1[1], hi[2] * alpha + bias1[2], hi[3] * alpha + bias1[3]};
                    const uint32_t o2 = offs(m, ldc, 2), o4 = offs(m, ldc, 4);
                    auto store_f32 = [&](__amdgpu_buffer_rsrc_t r, uint32_t o) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), r, o, 0, BVC_G8_ST_AUX);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), r, o, 16, BVC_G8_ST_AUX);
                    };
                    auto store_bf16 = [&](__amdgpu_buffer_rsrc_t r, uint32_t o) {
                        __builtin_amdgcn_raw_buffer_store_b128(
                            u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])}, r, o, 0, BVC_G8_ST_AUX);
                    };
                    if constexpr (EC == 0) {
                        // (handled above, in registers)
                    } else if constexpr (EC == 3) {
                        if (epi == EPI_DGELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t w = __float_as_uint(side0[c][e]);      // aux = gelu'(pre), saved by the forward epilogue
                                v[2 * e] *= __uint_as_float(w << 16);
                                v[2 * e + 1] *= __uint_as_float(w & 0xffff0000u);
                            }
                        } else {     // EPI_DRELU: aux = the forward ReLU output, the gradient passes where it was positive
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t w = __float_as_uint(side0[c][e]);
                                if (!((w & 0x7fffu) && !(w & 0x8000u))) v[2 * e] = 0.f;
                                if (!((w & 0x7fff0000u) && !(w & 0x80000000u))) v[2 * e + 1] = 0.f;
                            }
                        }
                        store_bf16(rc, o2);
                    } else if constexpr (EC == 1) {
                        if (epi == EPI_LOSS) {
                            if (have_c2) store_f32(rc2, o4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] -= side0[c][e]; v[4 + e] -= side1[c][e]; }
                            if (ok) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) sumsq += v[e] * v[e];
                            }
                            store_bf16(rc, o2);
                        } else {
                            if (epi == EPI_RESID || epi == EPI_POS || epi == EPI_E2D) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v[e] += side0[c][e]; v[4 + e] += side1[c][e]; }
                            }
                            uint32_t o = o4;
                            if (epi == EPI_E2D) {
                                const size_t orow = (size_t)(m / p.rin) * p.rout + (m % p.rin);
                                o = ok ? (uint32_t)((orow * ldc + n) * 4) : kDrop;
                            }
                            store_f32(rc, o);
                            if (epi == EPI_F32_BF16) store_bf16(rc2, o2);
                        }
                    } else {
                        store_f32(rc, o4);
                    }
                }
            }
            }   // pass
            if (EC == 1 && epi == EPI_LOSS) {
                // deterministic per-tile partial of sum (logit - label)^2: every wave leaves its sum in the first word of its own
                // parking rows, one lane folds the eight in a fixed order.  The wave rows run one barrier apart, so the fold
                // sits behind TWO barriers: after the second one wave row 1 has passed the first, i.e. has written.
                const float w = wave_sum(sumsq);
                // (a wave's parking area is 16 rows x WN floats: wave q's first word is float 16 WN q.  Round 2 indexed it 4 WN q - inside
                //  the rows OTHER waves were still reading back: one output element per affected tile came out as a partial sum; found by
                //  tools/g8_race_screen.py in round 3.  Since round 3 the head + MSE product of the training step runs on this epilogue from
                //  64 clips upward (pick_gemm8's `resid` class, tests/test_selection.py); tests/test_gpu_ops.py bit-compares C2 / diff /
                //  partials with the per-tile kernel.)
                float* red = reinterpret_cast<float*>(smem + 2 * TILE);
                if (lane == 0) red[wave * (16 * WN)] = w;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                if (tid == 0) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) s += red[q * (16 * WN)];
                    p.partial[cu.tile] = s;
                }
            }
        }
        if (do_rowsum) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                float v = rsum[mh];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int m = m0 + wm * WMR + 16 * (TMH * mh + wn) + lane;
                if (wn < TMH && (lane >> 4) == 0 && m < p.M) atomicAdd(p.rowsum + m, v * alpha);
            }
        }
        uid += u_step;
        if (uid >= u_hi) break;
        decode_unit<BM, BN>(g, uid, cu);
        // (atomics and the 384-wide tile's dword stores queue behind the prefetch like any store: the plain counted wait of the next
        //  K tile then waits for them as well - once per unit, conservative and exact)
        after_epi = EC >= 4 ? -1 : ((EC == 2 && atomic) || BN == 384) ? 0 : EC == 0 ? nstores : -1;
    }
    // drain the out-of-range tail of the stream, then pay back the stagger barrier
    wait_vmcnt<0>();
#ifndef BVC_G8_NO_STAGGER
    if (wm == 0) asm volatile("s_barrier" ::: "memory");
#endif
    if constexpr (EC == 5) {
        // one row of dgamma / dbeta partials per workgroup: the two wave rows' accumulators of every wave column, fixed order
        __syncthreads();
        const AS3 float* lcol = (const AS3 float*)((AS3 char*)smem + 2 * TILE + COL_OFF);
        float* pr = g.prob[0].ln_part + (size_t)blockIdx.x * 2 * BN;
        for (int c = tid; c < 2 * BN; c += 512) {
            const int which = c / BN, col = c % BN, w = col / WN, cc = col % WN;
            pr[c] = lcol[(w * 2 + which) * WN + cc] + lcol[((4 + w) * 2 + which) * WN + cc];
        }
    }
}

// ------------------------------------------------------------------ host side
static int g8_ncu() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        ncu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
        if (ncu < 8) ncu = 8;
    }
    return ncu;
}

// Balanced walk for a split-K weight-gradient group whose units do not fill the chip (the encoder layer's four gradients at 64 / 256
// clips: 108 tiles x 2 splits = 216 units on 256 CUs, 40 CUs idle for the whole launch).  Every tile gives up its last Lt K tiles: the
// splits shrink to Lb = (nt - Lt) / S, the T tails go to the I idle workgroups, q = ceil(T / I) each.  A tail pays its own epilogue
// (e, in K-tile times) - balance: Lb + e = q (c Lt + e).  Off (returns false) when the group is not of that shape or the gain is < 5 %.
static bool plan_balance(GemmGroup& g, int bm, int bn, int total, int ncu) {
    if (BVC_EXP_ENV("BVC_G8_NO_BALANCE") != nullptr) return false;
    const int S = g.prob[0].split_k, nt = (g.prob[0].K + 63) / 64;
    if ((S < 2 && !g.accum) || total >= ncu || total % S != 0) return false;     // (every unit of the walk adds into C)
    for (int i = 0; i < g.nprob; ++i) {
        const GemmProblem& p = g.prob[i];
        if (p.split_k != S || (p.K + 63) / 64 != nt || p.epi != EPI_F32) return false;
        if (g.tile_start[i] % S != 0 || g.panel[i] == 0) return false;
    }
    const int T = total / S, I = ncu - total, q = (T + I - 1) / I;
    double e = 6.0, c = 1.0;
    if (const char* v = BVC_EXP_ENV("BVC_G8_BALANCE_EPI")) e = atof(v);
    if (const char* v = BVC_EXP_ENV("BVC_G8_BALANCE_TAIL")) c = atof(v);
    const double lt = ((double)nt / S - (q - 1) * e) / (q * c + 1.0 / S);
    if (lt < 2.0) return false;
    const int per = (((nt + S - 1) / S) + 1) & ~1;
    const int lb = (int)((nt - lt) / S) & ~1;
    if (lb < 8 || nt - S * lb < 2 || lb > 0.95 * per) return false;
    g.bal_units = total;
    g.bal_lb = lb;
    g.bal_tiles = T;
    (void)bm; (void)bn;
    return true;
}

template <int BM, int BN, bool AT, bool BT, int EC>
static int launch_gemm8_one(const GemmGroup& g_in, int total, hipStream_t stream) {
    // two K-tile slots + the epilogue region: 16 parked rows per wave (classes 1-3; 16 x 48 for the 384-wide tile) or the bias
    // copy of class 0 (32 KiB: N <= 8192)
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 64 * 2 + (EC == 0 ? (size_t)32768 : EC == 4 ? (size_t)(3 * BN * 4 + 4096) : EC == 5 ? (size_t)(BN * 4 + 4096 + 8 * 2 * (BN / 4) * 4) :
                                                              BN == 384 ? (size_t)8 * 16 * 48 * 4 : (size_t)8 * 16 * (BN / 4) * 4);
    static_assert(lds <= 160 * 1024, "LDS per CU");
    if (dry_run().on) {
        snprintf(dry_run().name, sizeof(dry_run().name), "bvc::gemm8_kernel<%d, %d, %s, %s, %d>", BM, BN, AT ? "true" : "false", BT ? "true" : "false", EC);
        return BVC_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel<BM, BN, AT, BT, EC>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int ncu = g8_ncu();
    GemmGroup g = g_in;
    int grid = total < ncu ? ((total + 7) / 8) * 8 : ncu;      // one workgroup per CU, a multiple of the 8 XCDs
    if constexpr (EC == 2) {
        if (plan_balance(g, BM, BN, total, ncu)) grid = ncu;    // the full-length units partition over the XCDs as before; every CU gets a workgroup
    }
    if constexpr (EC >= 4) {
        // one unit's time in 64-cycle quanta at ~2.1 GHz: 1.45 us per 128 x 384 x 64 K tile + the epilogue's bytes at a CU's fair share of HBM
        const double us = 1.45 * ((g.prob[0].K + 63) / 64) + (EC == 4 ? 23.0 : 32.0);
        g.stagger = options().row_stagger && total > ncu ? (int)(us * 33.0) : 0;
    }
    hipLaunchKernelGGL((gemm8_kernel<BM, BN, AT, BT, EC>), dim3(grid), dim3(512), lds, stream, g, total);
    BVC_CHECK_HIP(hipGetLastError());
    (void)0;
    if constexpr (EC == 5)      // every workgroup of the grid left one row of dgamma / dbeta partials
        return launch_ln_param_reduce(g.prob[0].ln_part, grid, BN, g.prob[0].ln_dgamma, g.prob[0].ln_dbeta, stream);
    return BVC_OK;
}

static int epi_class(int epi, GemmLayout layout) {
    if (layout == GEMM_TN) return epi == EPI_F32 ? 2 : -1;
    if (epi == EPI_RESID_LN) return layout == GEMM_NT ? 4 : -1;
    if (epi == EPI_DLN) return layout == GEMM_NN ? 5 : -1;
    switch (epi) {
        case EPI_BF16: case EPI_GELU: case EPI_RELU: return 0;
        case EPI_DGELU: case EPI_DRELU: return 3;
        case EPI_F32: case EPI_RESID: case EPI_POS: case EPI_E2D: case EPI_LOSS: case EPI_F32_BF16: return 1;
        default: return -1;
    }
}

// Launcher hook used by launch_gemm (gemm.hip): bn = 256 / 128 (256-row tiles) or 384 (128-row tiles, weight gradients only).
// Returns BVC_OK after launching, 1 when the group is not eligible.
int launch_gemm8(const GemmGroup& g, GemmLayout layout, int bn, hipStream_t stream) {
    const int total = g.tile_start[g.nprob];
    int ec = -2;
    for (int i = 0; i < g.nprob; ++i) {
        const GemmProblem& p = g.prob[i];
        const int c = epi_class(p.epi, layout);
        if (c < 0 || (ec != -2 && c != ec)) return 1;
        ec = c;
        if (p.a_bytes >= kInvalidBase || p.b_bytes >= kInvalidBase) return 1;
        if (layout != GEMM_TN && (p.split_k != 1 || p.K % 64 != 0)) return 1;
        if (layout == GEMM_TN && p.split_k > 1 && p.epi != EPI_F32) return 1;
        if (p.rowsum && layout != GEMM_TN) return 1;
        // the f32-side epilogues keep 8 floats of side input per chunk in registers: one pass on 256 x 128 tiles, two on 256 x 256
        // class 0 keeps the whole (tile-padded) bias vector in LDS (32 KiB): one problem, N up to 8192
        if (c == 0 && (g.nprob != 1 || (size_t)((p.N + bn - 1) / bn) * bn * 4 > (size_t)32768)) return 1;
    }
    if (total <= 0) return 1;
#define BVC_G8(BN_, AT_, BT_, EC_) return launch_gemm8_one<256, BN_, AT_, BT_, EC_>(g, total, stream)
    if (bn == 384) {
        if (layout == GEMM_TN) return ec == 2 ? launch_gemm8_one<128, 384, true, true, 2>(g, total, stream) : 1;
        // row epilogues: one problem whose rows are exactly one tile wide, offsets of whole rows inside 32 bits
        const GemmProblem& p = g.prob[0];
        if (g.nprob != 1 || p.N != 384 || p.ldc != 384 || (double)p.M * 1536.0 >= 4294000000.0) return 1;
        if (ec == 4 && layout == GEMM_NT) return launch_gemm8_one<128, 384, false, false, 4>(g, total, stream);
        if (ec == 5 && layout == GEMM_NN) return launch_gemm8_one<128, 384, false, true, 5>(g, total, stream);
        return 1;
    }
    if (ec >= 4) return 1;
    if (layout == GEMM_NT) {
        if (ec == 0) { if (bn == 256) BVC_G8(256, false, false, 0); else BVC_G8(128, false, false, 0); }
        if (ec == 3) { if (bn == 256) BVC_G8(256, false, false, 3); else BVC_G8(128, false, false, 3); }
        if (ec == 1) { if (bn == 256) BVC_G8(256, false, false, 1); else BVC_G8(128, false, false, 1); }
    } else if (layout == GEMM_NN) {
        if (ec == 0) { if (bn == 256) BVC_G8(256, false, true, 0); else BVC_G8(128, false, true, 0); }
        if (ec == 3) { if (bn == 256) BVC_G8(256, false, true, 3); else BVC_G8(128, false, true, 3); }
        if (ec == 1) { if (bn == 256) BVC_G8(256, false, true, 1); else BVC_G8(128, false, true, 1); }
    } else {
        if (bn == 256) BVC_G8(256, true, true, 2); else BVC_G8(128, true, true, 2);
    }
#undef BVC_G8
    return 1;
}

}  // namespace bvc
