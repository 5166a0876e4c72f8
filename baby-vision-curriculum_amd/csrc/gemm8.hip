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
#include <stdlib.h>

#include <type_traits>

#include "gemm_tile.h"

namespace bvc {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr uint32_t kInvalidBase = 0x80000000u;   // beyond every operand this kernel accepts (extents < 2 GiB)

struct Unit {
    int pi, m0, n0, kt0, nkt, split, tile;
};

// unit id -> problem, tile, K range.  Uniform (kernel arguments and blockIdx only).
template <int BN>
__device__ __forceinline__ void decode_unit(const GemmGroup& g, int uid, Unit& u) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
        if (i < g.nprob && uid >= g.tile_start[i]) pi = i;
    const GemmProblem& p = g.prob[pi];
    const int lid = uid - g.tile_start[pi];
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + 255) / 256;
    const int ntiles = tiles_m * tiles_n;
    const int G = g.panel[pi];
    int split, tm, tn;
    if (G > 0) {
        split = lid / ntiles;
        tile_of(lid - split * ntiles, tiles_m, tiles_n, G, tm, tn);
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
    // integer divisions run on the vector ALU: pin the (uniform) results to scalar registers
    u.pi = pi;
    u.m0 = __builtin_amdgcn_readfirstlane(tm * 256);
    u.n0 = __builtin_amdgcn_readfirstlane(tn * BN);
    u.kt0 = __builtin_amdgcn_readfirstlane(split * per);
    u.nkt = __builtin_amdgcn_readfirstlane(per);
    u.split = __builtin_amdgcn_readfirstlane(split);
    u.tile = __builtin_amdgcn_readfirstlane(tm * tiles_n + tn);
}

}  // namespace

// EC (epilogue class): 0 = bf16 outputs without side inputs (BF16, GELU, RELU); 3 = bf16 outputs gated by a bf16 side input (DGELU,
// DRELU); 1 = f32 side inputs / outputs (F32, RESID, POS, E2D, LOSS, F32_BF16); 2 = weight gradients (TN): f32 store or split-K
// atomics, fused bias gradient.  Classes are separate instantiations because the side inputs of a whole unit sit in registers.
template <int BN, bool AT, bool BT, int EC>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(const GemmGroup g, const int total_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = 256 * 64 * 2, B_BYTES = BN * 64 * 2, TILE = A_BYTES + B_BYTES;
    constexpr int WN = BN / 4, TN = WN / 16, TM = 8;
    constexpr int NB = BN / 64, NBH = BN / 128;
    constexpr int n0c = AT ? NBH : (NB < 2 ? NB : 2), n1c = AT ? 2 : NB - n0c, n2c = AT ? NBH : 2, n3c = 2;
    constexpr int PT = n0c + n1c + n2c + n3c;          // LDS-DMA instructions per wave per K tile
    constexpr int W1 = PT, W3 = AT ? PT : n3c + n0c + n1c;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // XCD x owns a contiguous run of unit ids; its gridDim.x / 8 workgroups take them round robin
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int xq = total_units >> 3, xr = total_units & 7;
    const int x_lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int x_hi = x_lo + xq + (xcd < xr ? 1 : 0);
    int uid = x_lo + slot_id;
    if (uid >= x_hi) return;       // uniform per workgroup, before any barrier

    // ------------------------------------------------------------------ the staging cursor (runs ahead of the compute)
    Unit su;
    decode_unit<BN>(g, uid, su);
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
            s_uid += nslots;
            if (s_uid < x_hi) {
                const int old = su.pi;
                decode_unit<BN>(g, s_uid, su);
                if (su.pi != old) s_problem();
            } else {
                s_valid = false;
            }
        }
        s_ktile();
    };
    // piece j of an operand tile -> its 8 KiB of LDS at region + j * 8192, this wave's 1 KiB at + wave * 1024
    //   k-contiguous operand: rows 64 j .. 64 j + 63;   transposed operand: half h = j >> 1 (128 columns), k rows 32 (j & 1) ..
    const uint32_t lds0 = (uint32_t)(size_t)((AS3 char*)smem) + (uint32_t)wave * 1024u;   // this wave's 1 KiB of piece 0 of slot 0
    auto load_a = [&](char* region, int j) {
        const uint32_t off = s_la + s_baseA + (AT ? (uint32_t)(j & 1) * s_strideA + (uint32_t)(j >> 1) * 256u : (uint32_t)j * s_strideA);
        glds16(s_ra, off, lds0 + (uint32_t)(region - smem) + (uint32_t)j * 8192u);
    };
    auto load_b = [&](char* region, int j) {
        const uint32_t off = s_lb + s_baseB + (BT ? (uint32_t)(j & 1) * s_strideB + (uint32_t)(j >> 1) * 256u : (uint32_t)j * s_strideB);
        glds16(s_rb, off, lds0 + (uint32_t)(region - smem) + (uint32_t)j * 8192u);
    };
    // group GI of the cursor's K tile into LDS slot `buf`; after g3 the cursor moves to the next K tile of the stream
    auto stage_group = [&](auto gi_, char* buf) {
        constexpr int GI = decltype(gi_)::value;
        char* ra_ = buf;
        char* rb_ = buf + A_BYTES;
        if constexpr (!AT) {
            if constexpr (GI == 0) {
#pragma unroll
                for (int j = 0; j < n0c; ++j) load_b(rb_, j);
            } else if constexpr (GI == 1) {
#pragma unroll
                for (int j = n0c; j < NB; ++j) load_b(rb_, j);
            } else if constexpr (GI == 2) {
                load_a(ra_, 0); load_a(ra_, 2);
            } else {
                load_a(ra_, 1); load_a(ra_, 3);
            }
        } else {
            if constexpr (GI == 0) {
#pragma unroll
                for (int h = 0; h < NBH; ++h) load_b(rb_, 2 * h);
            } else if constexpr (GI == 1) {
                load_a(ra_, 0); load_a(ra_, 2);
            } else if constexpr (GI == 2) {
#pragma unroll
                for (int h = 0; h < NBH; ++h) load_b(rb_, 2 * h + 1);
            } else {
                load_a(ra_, 1); load_a(ra_, 3);
            }
        }
        if constexpr (GI == 3) s_advance();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    char* const buf0 = smem;
    char* const buf1 = smem + TILE;
    AS3 char* const wl = (AS3 char*)smem + 2 * TILE + wave * (16 * WN * 4);   // wave-private epilogue parking: 16 rows x WN f32

    // ------------------------------------------------------------------ prologue: K tile 0 and groups 0, 1 of K tile 1
    s_problem();
    s_ktile();
    stage_group(I0{}, buf0); stage_group(I1{}, buf0); stage_group(I2{}, buf0); stage_group(I3{}, buf0);
    stage_group(I0{}, buf1); stage_group(I1{}, buf1);
    wait_vmcnt<W3>();
    asm volatile("s_barrier" ::: "memory");
    if (wm == 1) asm volatile("s_barrier" ::: "memory");     // the stagger: wave row 1 runs one barrier behind wave row 0

    Unit cu;
    decode_unit<BN>(g, uid, cu);
    bool fresh = false;            // first K tile after an epilogue: its operands were drained before the stores

    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

    while (true) {
        const GemmProblem& p = g.prob[cu.pi];
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const bool do_rowsum = EC == 2 && p.rowsum != nullptr && cu.n0 == 0;   // bias gradient: wave wn takes row tiles wn and 4 + wn

        // one K tile out of LDS slot `cur`; `oth` is the other slot
        auto ktile = [&](char* cur, char* oth) {
            const char* la = cur;
            const char* lb = cur + A_BYTES;
            if constexpr (!AT) {
                bf16x8 bfr[2][TN];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (q == 0) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                bfr[ks][j] = BT ? read_frag<128, true>(lb + ((wn * WN) >> 7) * 16384, ((wn * WN) & 127) + 16 * j, ks, lane)
                                                : read_frag<BN, false>(lb, wn * WN + 16 * j, ks, lane);
                    }
                    bf16x8 af[2][2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) af[ks][ii] = read_frag<256, false>(la, wm * 128 + 16 * (2 * q + ii), ks, lane);
                    if (q == 0) stage_group(I2{}, oth);
                    else if (q == 1) stage_group(I3{}, oth);
                    else if (q == 2) stage_group(I0{}, cur);
                    else stage_group(I1{}, cur);
                    if (q == 1) { if (!fresh) wait_vmcnt<W1>(); }
                    if (q == 3) wait_vmcnt<W3>();
                    asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[2 * q + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][ii], acc[2 * q + ii][j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_barrier" ::: "memory");
                }
            } else {
                bf16x8 bfr[TN];      // the B fragments of one k half: read in the half's first phase, kept for its second
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ks = q >> 1, mh = q & 1;
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 af[4];
                    if (mh == 0) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            bfr[j] = read_frag<128, true>(lb + ((wn * WN) >> 7) * 16384, ((wn * WN) & 127) + 16 * j, ks, lane);
                    }
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) af[ii] = read_frag<128, true>(la + wm * 16384, 16 * (4 * mh + ii), ks, lane);
                    if (q == 0) stage_group(I2{}, oth);
                    else if (q == 1) stage_group(I3{}, oth);
                    else if (q == 2) stage_group(I0{}, cur);
                    else stage_group(I1{}, cur);
                    if (q == 1) { if (!fresh) wait_vmcnt<W1>(); }
                    if (q == 3) wait_vmcnt<W3>();
                    asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[4 * mh + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[ii], acc[4 * mh + ii][j], 0, 0, 0);
                    if (do_rowsum) {
                        // af[wn] is row tile 4 mh + wn of this phase: one extra MFMA against an all-ones operand
                        const bf16x8 a = wn == 0 ? af[0] : wn == 1 ? af[1] : wn == 2 ? af[2] : af[3];
                        accb[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a, accb[mh], 0, 0, 0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_barrier" ::: "memory");
                }
            }
            fresh = false;
        };

        for (int kt = 0; kt < cu.nkt; kt += 2) {
            ktile(buf0, buf1);
            ktile(buf1, buf0);
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
        constexpr int CPR = WN / 8;                   // 8-column chunks per row
        constexpr int RPU = 64 / CPR;                 // rows covered by the 64 lanes in one pass
        constexpr int U = 16 / RPU;                   // passes per 16-row round
        constexpr int NSIDE = TM * U;
        constexpr uint32_t kDrop = 0xFFFFFFF0u;       // >= every descriptor's extent: the access is dropped / reads 0
        auto park = [&](int i) {
            const int row = lane & 15;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int unit = (4 * j + (lane >> 4)) ^ (row & (UNITS - 1));
                *reinterpret_cast<AS3 f32x4*>(wl + row * (WN * 4) + unit * 16) = acc[i][j];
            }
        };
        const bool atomic = p.split_k > 1;
        if (EC == 2 && atomic) {
            // split-K: f32 atomics, one dword per lane, whole contiguous rows per wave-instruction (256 B / two 128-B rows)
            float* cbase = reinterpret_cast<float*>(p.C);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                park(i);
#pragma unroll 4
                for (int idx = lane; idx < 16 * WN; idx += 64) {
                    const int row = idx / WN, col = idx % WN;
                    const int m = m0 + wm * 128 + 16 * i + row, n = n0 + wn * WN + col;
                    const int unit = (col >> 2) ^ (row & (UNITS - 1));
                    const float v = *reinterpret_cast<const AS3 float*>(wl + row * (WN * 4) + unit * 16 + (col & 3) * 4) * alpha;
                    if (m < Mrows && n < Ncols) atomicAdd(cbase + (size_t)m * ldc + n, v);
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
            f32x4 side0[(EC == 1 || EC == 3) ? NSIDE : 1], side1[(EC == 1) ? NSIDE : 1];
            if constexpr (EC == 3) {
                const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux, kDrop);
                const int ldaux = p.ldaux;
#pragma unroll
                for (int c = 0; c < NSIDE; ++c) {
                    const int m = m0 + wm * 128 + 16 * (c / U) + (((c % U) * 64 + lane) / CPR);
                    side0[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, offs(m, ldaux, 2), 0, 0));
                }
            } else if constexpr (EC == 1) {
                const bool side_f32 = epi == EPI_RESID || epi == EPI_POS || epi == EPI_E2D || epi == EPI_LOSS;
                const bool by_tok = epi == EPI_POS || epi == EPI_E2D;
                const void* sbase = epi == EPI_RESID ? (const void*)p.resid : epi == EPI_LOSS ? (const void*)p.labels : (const void*)p.pos;
                const __amdgpu_buffer_rsrc_t rside = make_rsrc(sbase, side_f32 ? kDrop : 0u);
                const __amdgpu_buffer_rsrc_t rtok = make_rsrc(p.rowtok, by_tok ? kDrop : 0u);
#pragma unroll
                for (int c = 0; c < NSIDE; ++c) {
                    const int m = m0 + wm * 128 + 16 * (c / U) + (((c % U) * 64 + lane) / CPR);
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
                for (int c = 0; c < NSIDE; ++c) asm volatile("" : "+v"(side0[c]));
            }
            if constexpr (EC == 1) {
#pragma unroll
                for (int c = 0; c < NSIDE; ++c) asm volatile("" : "+v"(side1[c]));
            }
            float sumsq = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                park(i);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = (u * 64 + lane) / CPR;
                    const int m = m0 + wm * 128 + 16 * i + row;
                    const f32x4 lo = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc) ^ (row & (UNITS - 1))) << 4));
                    const f32x4 hi = *reinterpret_cast<const AS3 f32x4*>(wl + row * (WN * 4) + (((2 * cc + 1) ^ (row & (UNITS - 1))) << 4));
                    const int c = U * i + u;
                    const bool ok = m < Mrows && ncol_ok;
                    float v[8] = {lo[0] * alpha + bias0[0], lo[1] * alpha + bias0[1], lo[2] * alpha + bias0[2], lo[3] * alpha + bias0[3],
                                  hi[0] * alpha + bias1[0], hi[1] * alpha + bias1[1], hi[2] * alpha + bias1[2], hi[3] * alpha + bias1[3]};
                    const uint32_t o2 = offs(m, ldc, 2), o4 = offs(m, ldc, 4);
                    auto store_f32 = [&](__amdgpu_buffer_rsrc_t r, uint32_t o) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), r, o, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), r, o, 16, 0);
                    };
                    auto store_bf16 = [&](__amdgpu_buffer_rsrc_t r, uint32_t o) {
                        __builtin_amdgcn_raw_buffer_store_b128(
                            u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])}, r, o, 0, 0);
                    };
                    if constexpr (EC == 0) {
                        if (epi == EPI_RELU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        store_bf16(rc, o2);
                        if (epi == EPI_GELU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
                            store_bf16(rc2, o2);
                        }
                    } else if constexpr (EC == 3) {
                        if (epi == EPI_DGELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t w = __float_as_uint(side0[c][e]);
                                v[2 * e] *= dgelu_f(__uint_as_float(w << 16));
                                v[2 * e + 1] *= dgelu_f(__uint_as_float(w & 0xffff0000u));
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
            if (EC == 1 && epi == EPI_LOSS) {
                // deterministic per-tile partial of sum (logit - label)^2: every wave leaves its sum in the first word of its own
                // parking rows, one lane folds the eight in a fixed order.  The wave rows run one barrier apart, so the fold
                // sits behind TWO barriers: after the second one wave row 1 has passed the first, i.e. has written.
                const float w = wave_sum(sumsq);
                float* red = reinterpret_cast<float*>(smem + 2 * TILE);
                if (lane == 0) red[wave * (4 * WN)] = w;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                if (tid == 0) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) s += red[q * (4 * WN)];
                    p.partial[cu.tile] = s;
                }
            }
        }
        if (do_rowsum && (lane >> 4) == 0) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                const int m = m0 + wm * 128 + 16 * (4 * mh + wn) + lane;
                if (m < p.M) atomicAdd(p.rowsum + m, accb[mh][0] * alpha);
            }
        }
        uid += nslots;
        if (uid >= x_hi) break;
        decode_unit<BN>(g, uid, cu);
        fresh = !(EC == 2 && atomic);
    }
    // drain the out-of-range tail of the stream, then pay back the stagger barrier
    wait_vmcnt<0>();
    if (wm == 0) asm volatile("s_barrier" ::: "memory");
}

// ------------------------------------------------------------------ host side
template <int BN, bool AT, bool BT, int EC>
static int launch_gemm8_one(const GemmGroup& g, int total, hipStream_t stream) {
    constexpr size_t lds = 2 * (size_t)(256 + BN) * 64 * 2 + 8 * 16 * (BN / 4) * 4;    // 160 KiB (BN = 256) / 112 KiB (BN = 128)
    static bool attr_set = false;
    if (!attr_set) {
        BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel<BN, AT, BT, EC>),
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
    hipLaunchKernelGGL((gemm8_kernel<BN, AT, BT, EC>), dim3(grid), dim3(512), lds, stream, g, total);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

static int epi_class(int epi, GemmLayout layout) {
    if (layout == GEMM_TN) return epi == EPI_F32 ? 2 : -1;
    switch (epi) {
        case EPI_BF16: case EPI_GELU: case EPI_RELU: return 0;
        case EPI_DGELU: case EPI_DRELU: return 3;
        case EPI_F32: case EPI_RESID: case EPI_POS: case EPI_E2D: case EPI_LOSS: case EPI_F32_BF16: return 1;
        default: return -1;
    }
}

// Launcher hook used by launch_gemm (gemm.hip): bn = 256 / 128.  Returns BVC_OK after launching, 1 when the group is not eligible.
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
        // the f32-side epilogues keep 8 floats of side input per chunk in registers: 256 x 128 tiles only
        if (c == 1 && bn != 128) return 1;
    }
    if (total <= 0) return 1;
#define BVC_G8(BN_, AT_, BT_, EC_) return launch_gemm8_one<BN_, AT_, BT_, EC_>(g, total, stream)
    if (layout == GEMM_NT) {
        if (ec == 0) { if (bn == 256) BVC_G8(256, false, false, 0); else BVC_G8(128, false, false, 0); }
        if (ec == 3) { if (bn == 256) BVC_G8(256, false, false, 3); else BVC_G8(128, false, false, 3); }
        if (ec == 1) BVC_G8(128, false, false, 1);
    } else if (layout == GEMM_NN) {
        if (ec == 0) { if (bn == 256) BVC_G8(256, false, true, 0); else BVC_G8(128, false, true, 0); }
        if (ec == 3) { if (bn == 256) BVC_G8(256, false, true, 3); else BVC_G8(128, false, true, 3); }
        if (ec == 1) BVC_G8(128, false, true, 1);
    } else {
        if (bn == 256) BVC_G8(256, true, true, 2); else BVC_G8(128, true, true, 2);
    }
#undef BVC_G8
    return 1;
}

}  // namespace bvc
