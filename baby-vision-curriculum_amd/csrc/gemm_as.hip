// A-stationary bf16 GEMM for the K = 384 products of the 384-wide stacks (VideoMAE decoder qkv / fc1 + GELU, JEPA predictor): gfx950.
//
// Why a third structure (round 5; profiles/r05_*): on these products the 256 x 256 kernel of gemm8.hip spends as long in a tile's
// epilogue as in its six K tiles (decoder fc1 + GELU: two M x 1536 bf16 outputs for a K = 384 contraction), one workgroup per CU has
// nothing to run under that epilogue, and every attempt to overlap the two inside gemm8's phase protocol (round 4) lost in the K loop
// what it gained: a smaller tile needs more L2 -> LDS fill per FLOP, and a second accumulator set of 128 registers does not fit.
// With K = 384 the A block of a 128-row unit is only 96 KiB:
//   * it is staged into LDS ONCE per unit (LDS-DMA, during the previous unit) and each wave keeps ITS 32 rows x 384 k as MFMA
//     fragments in 96 registers for the whole sweep over N - no A traffic, no A fragment reads inside the sweep;
//   * only B (the weights: <= 1.2 MB, L2-resident) streams, as K tiles of 128 columns x 64 k = 16 KiB through a three-slot ring:
//     128 FLOP per byte of fill, what a 256 x 256 tile needs, at a quarter of its accumulators: 8 waves as 4 (M) x 2 (N), wave tile
//     32 x 64, 32 accumulator registers - so TWO sets fit, and the finished N tile is converted and stored (bias, GELU + GELU',
//     bf16 packing, 16-byte stores) between the MFMAs of the next one, a pair of 16 x 16 tiles per K step;
//   * a K step is two segments, {8 B fragment reads, the LDS-DMA of the step two ahead, counted vmcnt for the step one ahead, lgkmcnt(0),
//     s_barrier} and {16 MFMAs (+ epilogue slice), s_barrier}; waves 4 - 7 (the second wave of every SIMD) run ONE BARRIER BEHIND
//     waves 0 - 3, so that on every SIMD one wave issues MFMAs while the other issues its reads and DMA (the stagger of gemm8.hip);
//   * every wave issues two LDS-DMA operations per step (its share of the B K tile) in the read segment, and the twelve pieces of the
//     NEXT unit's A block in one burst behind the MFMAs of a unit's second K step; the counted waits are compile-time constants picked
//     by two uniform flags (stores issued yet? burst issued?); the epilogue's stores are counted into the wait that follows them
//     (vmcnt is one in-order counter).  A ring slot is re-staged one barrier after its last read, which is why the read segment ends in
//     lgkmcnt(0): the reads have RETURNED before any wave passes that barrier.
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
constexpr int AS_A_LDS = 128 * 384 * 2;          // the unit's A block: six [128][64] images
constexpr int AS_SLOT = 128 * 64 * 2;            // one B K tile: [128 n][64 k]
constexpr int AS_RING = 3;
constexpr int AS_SCRATCH_OFF = AS_A_LDS + AS_RING * AS_SLOT;         // 8 x 1 KiB: where dummy LDS-DMA operations land
constexpr int AS_BIAS_OFF = AS_SCRATCH_OFF + 8 * 1024;               // f32 bias copy (N <= 1536)
constexpr int AS_BIAS_MAX = 1536;
constexpr size_t AS_LDS_BYTES = (size_t)AS_BIAS_OFF + AS_BIAS_MAX * 4;

}  // namespace

// GELU: C <- gelu'(v), C2 <- gelu(v) (EPI_GELU); otherwise C <- bf16(v) (EPI_BF16), v = alpha acc + bias.
// OVL: the epilogue of N tile n runs between the MFMAs of N tile n + 1 (second accumulator set); false = after its own K steps (A/B).
template <bool GELU, bool OVL>
__global__ __launch_bounds__(512, 1) void gemm_as_kernel(const GemmProblem p, const int nunits, const int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(AS_LDS_BYTES <= 160 * 1024, "LDS per CU");
    constexpr int NSTP = GELU ? 2 : 1;           // stores per lane and (row tile, column-tile pair)
    constexpr int NST = 4 * NSTP;                // stores per lane and N tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm4 = wave >> 1, wn2 = wave & 1;
    const int NT = p.N / 128;
    const int lda = p.lda, ldb = p.ldb, ldc = p.ldc, Mrows = p.M;
    const float alpha = p.alpha_dev ? p.alpha * p.alpha_dev[0] : p.alpha;

    {   // bias -> LDS (plain loads, waited for here, before any LDS-DMA is in flight)
        AS3 float* lbias = (AS3 float*)((AS3 char*)smem + AS_BIAS_OFF);
        for (int i = tid; i < p.N; i += 512) lbias[i] = p.bias ? p.bias[i] : 0.f;
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

    // ------------------------------------------------------------------ the staging cursor: two K steps ahead of the compute
    int s_uid = uid, s_nt = 0, s_kt = 0;
    auto stage = [&](auto slot_) {
        constexpr int S = decltype(slot_)::value;
        const bool ok = s_uid < nunits;
        const uint32_t bbase = ok ? (uint32_t)((s_nt * 128 * ldb + s_kt * 64) * 2) : kInvalidBase;
        glds16(rb, lb + bbase, lds_w + (uint32_t)(AS_A_LDS + S * AS_SLOT));
        glds16(rb, lb + bbase + (uint32_t)(64 * ldb * 2), lds_w + (uint32_t)(AS_A_LDS + S * AS_SLOT + 8192));
        if (++s_kt == AS_NKT) {
            s_kt = 0;
            if (++s_nt == NT) { s_nt = 0; s_uid += ustep; }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ------------------------------------------------------------------ prologue: the first unit's A block, K steps 0 and 1
#pragma unroll
    for (int a = 0; a < 2 * AS_NKT; ++a)
        glds16(ra, la + (uint32_t)(((uid * 128 + 64 * (a & 1)) * lda + (a >> 1) * 64) * 2), lds_w + (uint32_t)((a >> 1) * 16384 + (a & 1) * 8192));
    stage(I0{});
    stage(I1{});
    wait_vmcnt<2>();                       // the A block and K step 0 (everything but step 1)
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

    // Epilogue of one (row tile, column-tile pair) of an N tile held in `a`: v = alpha acc + bias in the MFMA layout (lane = row l & 15,
    // 4 consecutive columns at 4 (l >> 4) of each 16 x 16 tile), v_permlane16_swap between the two tiles -> 16 bytes per lane.
    // In two halves, so that the overlapped form can put the first tile's arithmetic into a K step's READ segment (under the latency of
    // the fragment reads, beside the other wave group's MFMAs) and the second tile's, the packing and the stores behind the step's MFMAs:
    // a step costs 2 max(read segment, MFMA segment) - the two wave groups alternate - so the epilogue has to be split between them.
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

    while (true) {
        const int m0 = uid * 128;
        const bool a_burst = uid + ustep < nunits;
        // this wave's A fragments: rows 32 wm4 + 16 rt + (l & 15), k = 64 kt + 32 ks + 8 (l >> 4) .. + 7
        bf16x8 afr[2][2 * AS_NKT];
#pragma unroll
        for (int kt = 0; kt < AS_NKT; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) afr[rt][2 * kt + ks] = read_frag<128, false>(smem + kt * 16384, 32 * wm4 + 16 * rt, ks, lane);
        // every wave has its fragments: the block may receive the next unit's rows (pieces issued from this unit's first K step on)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto kstep = [&](auto kt_) {
                constexpr int kt = decltype(kt_)::value;
                __builtin_amdgcn_sched_barrier(0);
                // ---- read segment: this step's B fragments (its LDS-DMA was retired by the PREVIOUS step's wait, a barrier ago), the
                // LDS-DMA of the step two ahead, the counted wait for the step one ahead
                const char* slot = smem + AS_A_LDS + (kt % 3) * AS_SLOT;
                bf16x8 bfr[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfr[ks][j] = read_frag<128, false>(slot, 64 * wn2 + 16 * j, ks, lane);
                stage(std::integral_constant<int, (kt + 2) % 3>{});
                float ea[4], eg[4];       // first half of the previous N tile's pair (kt >> 1, kt & 1): computed here, stored behind the MFMAs
                if constexpr (OVL && kt < 4) {
                    if (have_prev) epi_first(accp, prev_n0, std::integral_constant<int, (kt >> 1)>{}, std::integral_constant<int, (kt & 1)>{}, ea, eg);
                }
                // younger than the next step's LDS-DMA: the two operations just issued, the stores of the previous step's MFMA segment (OVL: a
                // pair of 16 x 16 tiles per step in steps 0 .. 3; otherwise the whole tile behind step 5) and, in the third step of a unit,
                // the twelve A pieces issued behind the second step's MFMAs
                constexpr int prior = OVL ? (kt >= 1 && kt <= 4 ? NSTP : 0) : (kt == 0 ? NST : 0);
                if (kt == 2 && nt == 0 && a_burst) {
                    if (have_prev) wait_vmcnt<2 + prior + 2 * AS_NKT>(); else wait_vmcnt<2 + 2 * AS_NKT>();
                } else {
                    if (have_prev) wait_vmcnt<2 + prior>(); else wait_vmcnt<2>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
                if constexpr (kt == 1) {
                    // the NEXT unit's A block, all twelve pieces: every wave - the late group included - took this unit's fragments out of
                    // the block before the barrier at the top of the unit, and two more barriers have passed since
                    if (nt == 0 && a_burst) {
#pragma unroll
                        for (int a = 0; a < 2 * AS_NKT; ++a)
                            glds16(ra, la + (uint32_t)((((uid + ustep) * 128 + 64 * (a & 1)) * lda + (a >> 1) * 64) * 2),
                                   lds_w + (uint32_t)((a >> 1) * 16384 + (a & 1) * 8192));
                    }
                }
                if constexpr (OVL && kt < 4) {
                    if (have_prev) epi_second(accp, prev_m0, prev_n0, std::integral_constant<int, (kt >> 1)>{}, std::integral_constant<int, (kt & 1)>{}, ea, eg);
                }
                if constexpr (!OVL && kt == 5) {
                    epi_pair(acc, m0, nt * 128, I0{}, I0{}); epi_pair(acc, m0, nt * 128, I0{}, I1{});
                    epi_pair(acc, m0, nt * 128, I1{}, I0{}); epi_pair(acc, m0, nt * 128, I1{}, I1{});
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
            };
            kstep(I0{}); kstep(I1{}); kstep(I2{});
            kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
            if constexpr (OVL) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accp[i][j] = acc[i][j];
                prev_m0 = m0; prev_n0 = nt * 128;
            }
            have_prev = true;
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
    // (N >= 384: the next unit's A block is staged during K steps 1 .. 12 of a unit, which therefore has at least three N tiles)
    if (layout != GEMM_NT || p.K != 384 || p.N % 128 != 0 || p.N < 384 || p.N > AS_BIAS_MAX || p.split_k != 1) return false;
    if (p.epi != EPI_BF16 && p.epi != EPI_GELU) return false;
    if (p.epi == EPI_GELU && p.C2 == nullptr) return false;
    if (p.a_bytes >= kInvalidBase || p.b_bytes >= kInvalidBase || (size_t)p.M * p.ldc * 2 >= 0xFFFFFFF0ull) return false;
    return p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 8 == 0 && p.M > 0;
}

template <bool GELU, bool OVL>
static int launch_as_one(const GemmProblem& p, hipStream_t stream) {
    if (dry_run().on) {
        snprintf(dry_run().name, sizeof(dry_run().name), "bvc::gemm_as_kernel<%s, %s>", GELU ? "true" : "false", OVL ? "true" : "false");
        return BVC_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_as_kernel<GELU, OVL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)AS_LDS_BYTES));
        attr_set = true;
    }
    const int nunits = (p.M + 127) / 128, ncu = as_ncu();
    const int grid = nunits < ncu ? nunits : ncu;
    const char* e = BVC_EXP_ENV("BVC_GEMM_DEBUG");
    hipLaunchKernelGGL((gemm_as_kernel<GELU, OVL>), dim3(grid), dim3(512), AS_LDS_BYTES, stream, p, nunits, e ? atoi(e) : 0);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

// tile config 15 (overlapped epilogue) / 16 (epilogue after the tile: A/B only).  Returns 1 when the problem is not eligible.
int launch_gemm_as(const GemmProblem& p, GemmLayout layout, bool overlap, hipStream_t stream) {
    if (!gemm_as_ok(p, layout)) return 1;
    if (p.epi == EPI_GELU) return overlap ? launch_as_one<true, true>(p, stream) : launch_as_one<true, false>(p, stream);
    return overlap ? launch_as_one<false, true>(p, stream) : launch_as_one<false, false>(p, stream);
}

}  // namespace bvc
