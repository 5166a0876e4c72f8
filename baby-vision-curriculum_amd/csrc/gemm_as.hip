// A-stationary bf16 GEMM for the K = 384 products of the 384-wide stacks (VideoMAE decoder qkv / fc1 + GELU, JEPA predictor): gfx950.
//
// Why a third structure (round 5; profiles/r05_*): on these products the 256 x 256 kernel of gemm8.hip spends as long in a tile's
// epilogue as in its six K tiles (decoder fc1 + GELU: two M x 1536 bf16 outputs for a K = 384 contraction), one workgroup per CU has
// nothing to run under that epilogue, and every attempt to overlap the two inside gemm8's phase protocol (round 4) lost in the K loop
// what it gained: a smaller tile needs more L2 -> LDS fill per FLOP, and a second accumulator set of 128 registers does not fit.
// With K = 384 the A block of a 128-row unit is only 96 KiB:
//   * each wave keeps ITS 32 rows x 384 k as MFMA fragments in 96 registers for the whole sweep over N - no A traffic, no A fragment
//     reads inside the sweep;
//   * only B (the weights: <= 1.2 MB, L2-resident) streams, as K tiles of 128 columns x 64 k = 16 KiB: 128 FLOP per byte of fill, what a
//     256 x 256 tile needs, at a quarter of its accumulators: 8 waves as 4 (M) x 2 (N), wave tile 32 x 64, 32 accumulator registers - so
//     TWO sets fit, and the finished N tile is converted and stored (bias, GELU + GELU', bf16 packing, 16-byte stores) during the K
//     steps of the next one, a pair of 16 x 16 tiles per step;
//   * ONE ring of eight 16-KiB slots carries every tile of the stream - a unit's six A tiles (128 rows x 64 k each: an "A step" moves
//     them from the slot into the fragment registers), then its N tiles' B tiles - staged SIX steps ahead.  vmcnt retires in order, so
//     every wait for a tile also waits for whatever was issued before that tile: with two steps of distance (the first form of this
//     kernel) the epilogue's stores had ~1.5 steps to complete and a unit's A rows, which come from HBM, stalled the stream (stores
//     dropped at the descriptor: 308 vs 437 us on the decoder's qkv).  Six steps are ~3 us: HBM latency and the stores' completion both
//     fit behind them;
//   * a step is two segments, {fragment reads, the LDS-DMA of the tile six ahead, counted vmcnt for the tile one ahead, lgkmcnt(0),
//     s_barrier} and {16 MFMAs, epilogue slice, s_barrier}; waves 4 - 7 (the second wave of every SIMD) run ONE BARRIER BEHIND waves
//     0 - 3, so that on every SIMD one wave issues MFMAs while the other issues its reads and DMA (the stagger of gemm8.hip).  The
//     counted wait is exact: ten LDS-DMA operations plus the stores of the last five steps (a shift register of store counts).
// Results are bit-identical to gemm_kernel's for the same problem: same K order per output element, same epilogue arithmetic.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm_tile.h"

namespace bvc {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr uint32_t kInvalidBase = 0x80000000u;   // beyond every operand this kernel accepts (extents < 2 GiB)
constexpr uint32_t kDrop = 0xFFFFFFF0u;          // >= every output descriptor's extent: the access is dropped

constexpr int AS_NKT = 6;                        // K = 384 = 6 K tiles of 64
constexpr int AS_SLOT = 128 * 64 * 2;            // one tile of the stream: [128 rows or columns][64 k]
constexpr int AS_RING = 8;                       // slots (a power of two)
constexpr int AS_DIST = 6;                       // a tile is staged this many steps before it is read
constexpr int AS_BIAS_OFF = AS_RING * AS_SLOT;   // f32 bias copy (N <= 1536)
constexpr int AS_BIAS_MAX = 1536;
constexpr size_t AS_LDS_BYTES = (size_t)AS_BIAS_OFF + AS_BIAS_MAX * 4;

// s_waitcnt vmcnt(n) for a run-time (uniform) n out of the values the stream produces: ten LDS-DMA operations (eight when they are
// issued behind the MFMAs) plus the stores of five steps
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
#define BVC_W(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        BVC_W(8) BVC_W(9) BVC_W(10) BVC_W(11) BVC_W(12) BVC_W(13) BVC_W(14) BVC_W(15) BVC_W(16) BVC_W(17) BVC_W(18) BVC_W(19) BVC_W(20)
        BVC_W(21) BVC_W(22) BVC_W(23) BVC_W(24) BVC_W(25) BVC_W(26)
#undef BVC_W
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;     // (never less than the LDS-DMA operations in flight)
    }
}

}  // namespace

// GELU: C <- gelu'(v), C2 <- gelu(v) (EPI_GELU); otherwise C <- bf16(v) (EPI_BF16), v = alpha acc + bias.
// OVL: the epilogue of N tile n runs during the K steps of N tile n + 1 (second accumulator set); false = behind its own last K step.
// LATE: the LDS-DMA of a step is issued behind its MFMAs instead of in its read segment (A/B).
template <bool GELU, bool OVL, bool LATE>
__global__ __launch_bounds__(512, 1) void gemm_as_kernel(const GemmProblem p, const int nunits, const int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(AS_LDS_BYTES <= 160 * 1024, "LDS per CU");
    static_assert((AS_RING & (AS_RING - 1)) == 0 && AS_RING >= AS_DIST + 2, "ring: a slot is re-staged two steps after its last read");
    constexpr int NSTP = GELU ? 2 : 1;           // stores per lane and (row tile, column-tile pair)
    constexpr int NST = 4 * NSTP;                // stores per lane and N tile
    constexpr int NDMA = LATE ? 2 * (AS_DIST - 2) : 2 * (AS_DIST - 1);      // LDS-DMA operations younger than the next step's tile at the wait

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm4 = wave >> 1, wn2 = wave & 1;
    const int NT = p.N / 128;
    const int lda = p.lda, ldb = p.ldb, ldc = p.ldc, Mrows = p.M;
    const float alpha = p.alpha_dev ? p.alpha * p.alpha_dev[0] : p.alpha;

    {   // bias -> LDS (plain loads, waited for here, before any LDS-DMA is in flight)
        AS3 float* lb_ = (AS3 float*)((AS3 char*)smem + AS_BIAS_OFF);
        for (int i = tid; i < p.N; i += 512) lb_[i] = p.bias ? p.bias[i] : 0.f;
        __syncthreads();
    }
    int uid = blockIdx.x;
    const int ustep = gridDim.x;
    if (uid >= nunits) return;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes), rb = make_rsrc(p.B, p.b_bytes);
    // LDS-DMA pieces are 64 rows x 64 k (8 KiB); this wave fills rows 8 wave .. + 7 of a piece, lane l the 16-byte chunk (l & 7) of
    // row 8 wave + (l >> 3), swizzled on the SOURCE side (the LDS image is linear per wave: gemm_tile.h)
    const int r8 = 8 * wave + (lane >> 3);
    const uint32_t la = (uint32_t)((r8 * lda + (((lane & 7) ^ swz_rows(r8)) << 3)) * 2);
    const uint32_t lb = (uint32_t)((r8 * ldb + (((lane & 7) ^ swz_rows(r8)) << 3)) * 2);
    const uint32_t lds_w = (uint32_t)(size_t)((AS3 char*)smem) + (uint32_t)wave * 1024u;

    // ------------------------------------------------------------------ the staging cursor: AS_DIST tiles ahead of the compute.
    // Tile sequence of a unit: A tiles kt = 0 .. 5 (rows of the unit), then B tiles (nt, kt).  s_ph counts tiles inside the unit.
    int s_uid = uid, s_ph = 0, s_g = 0;
    const int per_unit = AS_NKT * (NT + 1);
    auto stage = [&]() {
        const bool ok = s_uid < nunits;
        const uint32_t dst = lds_w + (uint32_t)((s_g & (AS_RING - 1)) * AS_SLOT);
        if (s_ph < AS_NKT) {
            const uint32_t base = ok ? (uint32_t)(((s_uid * 128) * lda + s_ph * 64) * 2) : kInvalidBase;
            glds16(ra, la + base, dst);
            glds16(ra, la + base + (uint32_t)(64 * lda * 2), dst + 8192u);
        } else {
            const int t = s_ph - AS_NKT, nt = t / AS_NKT, kt = t - nt * AS_NKT;
            const uint32_t base = ok ? (uint32_t)((nt * 128 * ldb + kt * 64) * 2) : kInvalidBase;
            glds16(rb, lb + base, dst);
            glds16(rb, lb + base + (uint32_t)(64 * ldb * 2), dst + 8192u);
        }
        ++s_g;
        if (++s_ph == per_unit) { s_ph = 0; s_uid += ustep; }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ------------------------------------------------------------------ prologue: the first AS_DIST tiles
#pragma unroll
    for (int i = 0; i < AS_DIST; ++i) stage();
    wait_vmcnt<2 * (AS_DIST - 1)>();       // tile 0
    asm volatile("s_barrier" ::: "memory");
    const int grp = wave >> 2;             // waves w and w + 4 share a SIMD
    if (grp == 1) asm volatile("s_barrier" ::: "memory");     // the stagger: waves 4 - 7 run one barrier behind waves 0 - 3

    const AS3 float* lbias = (const AS3 float*)((AS3 char*)smem + AS_BIAS_OFF);
    // (experiments build, BVC_GEMM_DEBUG bit 1: zero-record descriptors - every store is dropped by the range check while the instruction
    //  stream, the counters and the waits stay: prices the stores)
#ifdef BVC_EXPERIMENTS
    const uint32_t cext = (dbg & 1) ? 0u : kDrop;
#else
    const uint32_t cext = kDrop;
    (void)dbg;
#endif
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, cext), rc2 = make_rsrc(GELU ? p.C2 : p.C, cext);
    const int q4 = lane >> 4, r16 = lane & 15;

    f32x4 accp[2][4];                      // the previous N tile, waiting for its epilogue (OVL)
    int prev_m0 = 0, prev_n0 = 0;
    bool have_prev = false;
    int g = 0;                             // tiles consumed so far: tile g sits in slot g & 7
    uint32_t sthist = 0;                   // stores issued in the last five steps, a nibble per step (youngest lowest)

    // Epilogue of one (row tile, column-tile pair) of an N tile held in `a`: v = alpha acc + bias in the MFMA layout (lane = row l & 15,
    // 4 consecutive columns at 4 (l >> 4) of each 16 x 16 tile), v_permlane16_swap between the two tiles -> 16 bytes per lane.
    // In two halves: the overlapped form puts the first tile's arithmetic into a step's READ segment (under the latency of the fragment
    // reads, beside the other wave group's MFMAs) and the second tile's, the packing and the stores behind the step's MFMAs - a step
    // costs 2 max(read segment, MFMA segment), the two wave groups alternate.
    auto epi_first = [&](const f32x4 (&a)[2][4], int n0, auto rt_, auto jp_, float (&va)[4], float (&ga)[4]) {
        constexpr int rt = decltype(rt_)::value, jp = decltype(jp_)::value;
        const int nb = n0 + 64 * wn2 + 32 * jp + 4 * q4;
        const f32x4 b0 = *reinterpret_cast<const AS3 f32x4*>(lbias + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) va[e] = a[rt][2 * jp][e] * alpha + b0[e];
        if constexpr (GELU) gelu_split(va, ga);
    };
    auto epi_second = [&](const f32x4 (&a)[2][4], int m0, int n0, auto rt_, auto jp_, float (&va)[4], float (&ga)[4]) {
        constexpr int rt = decltype(rt_)::value, jp = decltype(jp_)::value;
        const int nb = n0 + 64 * wn2 + 32 * jp + 4 * q4;
        const f32x4 b1 = *reinterpret_cast<const AS3 f32x4*>(lbias + nb + 16);
        float vb[4], gb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vb[e] = a[rt][2 * jp + 1][e] * alpha + b1[e];
        if constexpr (GELU) gelu_split(vb, gb);
        const int m = m0 + 32 * wm4 + 16 * rt + r16;
        const int n = n0 + 64 * wn2 + 32 * jp + 16 * (q4 & 1) + 8 * (q4 >> 1);
        const uint32_t o = m < Mrows ? (uint32_t)(((size_t)m * ldc + n) * 2) : kDrop;
        auto emit = [&](__amdgpu_buffer_rsrc_t r, const float (&x)[4], const float (&y)[4]) {
            const uint32_t a0 = pack2bf(x[0], x[1]), a1 = pack2bf(x[2], x[3]);
            const uint32_t c0 = pack2bf(y[0], y[1]), c1 = pack2bf(y[2], y[3]);
            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, c0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, c1, false, false);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, r, o, 0, 0);
        };
        emit(rc, va, vb);                         // GELU: gelu'(v); otherwise v
        if constexpr (GELU) emit(rc2, ga, gb);    // gelu(v)
    };
    auto epi_pair = [&](const f32x4 (&a)[2][4], int m0, int n0, auto rt_, auto jp_) {
        float va[4], ga[4];
        epi_first(a, n0, rt_, jp_, va, ga);
        epi_second(a, m0, n0, rt_, jp_, va, ga);
    };
    // the wait that ends a read segment: the NEXT step's tile has landed (this wave's share; the barrier does the rest)
    auto wait_next = [&]() {
        const int st5 = (int)((sthist & 15u) + ((sthist >> 4) & 15u) + ((sthist >> 8) & 15u) + ((sthist >> 12) & 15u) + ((sthist >> 16) & 15u));
        wait_vmcnt_dyn(NDMA + st5);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    bf16x8 afr[2][2 * AS_NKT];             // this wave's A fragments: rows 32 wm4 + 16 rt + (l & 15), k = 64 kt + 32 ks + 8 (l >> 4) .. + 7
    while (true) {
        const int m0 = uid * 128;
        // ---- six A steps: the unit's A tiles leave their slots for the fragment registers
        auto astep = [&](auto kt_) {
            constexpr int kt = decltype(kt_)::value;
            __builtin_amdgcn_sched_barrier(0);
            const char* slot = smem + (g & (AS_RING - 1)) * AS_SLOT;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) afr[rt][2 * kt + ks] = read_frag<128, false>(slot, 32 * wm4 + 16 * rt, ks, lane);
            if constexpr (!LATE) stage();
            wait_next();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LATE) stage();
            sthist = (sthist << 4) & 0xFFFFFu;
            ++g;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
        };
        astep(I0{}); astep(I1{}); astep(std::integral_constant<int, 2>{});
        astep(std::integral_constant<int, 3>{}); astep(std::integral_constant<int, 4>{}); astep(std::integral_constant<int, 5>{});

        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto kstep = [&](auto kt_) {
                constexpr int kt = decltype(kt_)::value;
                __builtin_amdgcn_sched_barrier(0);
                // ---- read segment
                const char* slot = smem + (g & (AS_RING - 1)) * AS_SLOT;
                bf16x8 bfr[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfr[ks][j] = read_frag<128, false>(slot, 64 * wn2 + 16 * j, ks, lane);
                if constexpr (!LATE) stage();
                float ea[4], eg[4];       // first half of the previous N tile's pair (kt >> 1, kt & 1): computed here, stored behind the MFMAs
                if constexpr (OVL && kt < 4) {
                    if (have_prev) epi_first(accp, prev_n0, std::integral_constant<int, (kt >> 1)>{}, std::integral_constant<int, (kt & 1)>{}, ea, eg);
                }
                wait_next();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA segment
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], afr[rt][2 * kt + ks], acc[rt][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                if constexpr (LATE) stage();
                uint32_t st_now = 0;
                if constexpr (OVL && kt < 4) {
                    if (have_prev) {
                        epi_second(accp, prev_m0, prev_n0, std::integral_constant<int, (kt >> 1)>{}, std::integral_constant<int, (kt & 1)>{}, ea, eg);
                        st_now = NSTP;
                    }
                }
                if constexpr (!OVL && kt == 5) {
                    epi_pair(acc, m0, nt * 128, I0{}, I0{}); epi_pair(acc, m0, nt * 128, I0{}, I1{});
                    epi_pair(acc, m0, nt * 128, I1{}, I0{}); epi_pair(acc, m0, nt * 128, I1{}, I1{});
                    st_now = NST;
                }
                sthist = ((sthist << 4) | st_now) & 0xFFFFFu;
                ++g;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
            };
            kstep(I0{}); kstep(I1{}); kstep(std::integral_constant<int, 2>{});
            kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
            if constexpr (OVL) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accp[i][j] = acc[i][j];
                prev_m0 = m0; prev_n0 = nt * 128;
                have_prev = true;
            }
        }
        uid += ustep;
        if (uid >= nunits) break;
    }
    if constexpr (OVL) {     // the last N tile of the last unit
        epi_pair(accp, prev_m0, prev_n0, I0{}, I0{}); epi_pair(accp, prev_m0, prev_n0, I0{}, I1{});
        epi_pair(accp, prev_m0, prev_n0, I1{}, I0{}); epi_pair(accp, prev_m0, prev_n0, I1{}, I1{});
    }
    wait_vmcnt<0>();         // the out-of-range tail of the stream
    if (grp == 0) asm volatile("s_barrier" ::: "memory");      // pay back the stagger barrier
}

// ------------------------------------------------------------------ host side
static int as_ncu() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

// Does the A-stationary kernel take this problem?  NT, K = 384, whole 128-column tiles, bf16 outputs without side inputs.
bool gemm_as_ok(const GemmProblem& p, GemmLayout layout) {
    if (layout != GEMM_NT || p.K != 384 || p.N % 128 != 0 || p.N > AS_BIAS_MAX || p.split_k != 1) return false;
    if (p.epi != EPI_BF16 && p.epi != EPI_GELU) return false;
    if (p.epi == EPI_GELU && p.C2 == nullptr) return false;
    if (p.a_bytes >= kInvalidBase || p.b_bytes >= kInvalidBase || (size_t)p.M * p.ldc * 2 >= 0xFFFFFFF0ull) return false;
    return p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 8 == 0 && p.M > 0;
}

template <bool GELU, bool OVL, bool LATE>
static int launch_as_one(const GemmProblem& p, hipStream_t stream) {
    if (dry_run().on) {
        snprintf(dry_run().name, sizeof(dry_run().name), "bvc::gemm_as_kernel<%s, %s, %s>", GELU ? "true" : "false", OVL ? "true" : "false",
                 LATE ? "true" : "false");
        return BVC_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_as_kernel<GELU, OVL, LATE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)AS_LDS_BYTES));
        attr_set = true;
    }
    const int nunits = (p.M + 127) / 128, ncu = as_ncu();
    const int grid = nunits < ncu ? nunits : ncu;
    const char* e = BVC_EXP_ENV("BVC_GEMM_DEBUG");
    hipLaunchKernelGGL((gemm_as_kernel<GELU, OVL, LATE>), dim3(grid), dim3(512), AS_LDS_BYTES, stream, p, nunits, e ? atoi(e) : 0);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

// variant: bit 0 = epilogue behind the tile's own last K step instead of under the next tile (A/B), bit 1 = LDS-DMA issued behind the MFMAs
// of a step instead of in its read segment (A/B).  Returns 1 when the problem is not eligible.
int launch_gemm_as(const GemmProblem& p, GemmLayout layout, int variant, hipStream_t stream) {
    if (!gemm_as_ok(p, layout)) return 1;
    const bool gelu = p.epi == EPI_GELU, ovl = !(variant & 1), late = (variant & 2) != 0;
#define BVC_AS(G_, O_, L_) return launch_as_one<G_, O_, L_>(p, stream)
    if (gelu) { if (ovl) { if (late) BVC_AS(true, true, true); else BVC_AS(true, true, false); } else { if (late) BVC_AS(true, false, true); else BVC_AS(true, false, false); } }
    if (ovl) { if (late) BVC_AS(false, true, true); else BVC_AS(false, true, false); }
    if (late) BVC_AS(false, false, true); else BVC_AS(false, false, false);
#undef BVC_AS
}

}  // namespace bvc
