// EXPERIMENT (round 1, not on the product path yet): NT GEMM with a 256x128 workgroup tile, 32-deep K steps and 128x64 wave
// tiles.  Why: with 64x64 wave tiles every MFMA flop needs 1/32 B of LDS reads, i.e. 128 B/clk/CU at MFMA peak - exactly the LDS
// peak - and tools/ab/gemm_ksweep.py measures 816 TFLOP/s marginal (33 % of peak) for the 128x128 kernel.  A 128x64 wave tile
// needs 1/42.7 B/flop (75 % of the LDS peak at MFMA peak); 32-deep K steps keep two 24 KiB stages = 48 KiB per workgroup so
// that two workgroups still share a CU.  Reached through bvc_op_gemm(tile_cfg = 3): NT layout, one problem, EPI_BF16 (+ bias).
#ifdef BVC_EXPERIMENTS     // compiled into the library only for tools/ (see gemm_tile.h)
#include "gemm.h"

namespace bvc {

#define AS3 __attribute__((address_space(3)))

namespace {

__device__ __forceinline__ int swz32(int r) { return (r >> 2) & 3; }   // 64-B rows: rows r and r+4 share banks

// rows r0.. of a k-contiguous [R][ld] array, k0..k0+31 -> image [BR][32] bf16 (64-B rows); piece = 16 rows = 1 KiB
template <int BR>
__device__ __forceinline__ void stage32(__amdgpu_buffer_rsrc_t rs, int r0, int k0, int ld, char* lds, int wave, int lane) {
    constexpr int PIECES = BR / 16;
#pragma unroll
    for (int jj = 0; jj < PIECES / 4; ++jj) {
        const int j = wave + 4 * jj;
        const int r = 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ swz32(r);
        const uint32_t off = (uint32_t)(((r0 + r) * ld + k0 + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + j * 1024), 16, off, 0, 0, 0);
    }
}

// lane l gets X[out = rbase + (l & 15)][k = 8 (l >> 4) .. + 7]
__device__ __forceinline__ bf16x8 frag32(const char* lds, int rbase, int lane) {
    const int r = rbase + (lane & 15);
    const int g = lane >> 4;
    return *reinterpret_cast<const AS3 bf16x8*>((const AS3 char*)(lds) + r * 64 + ((g ^ swz32(r)) << 4));
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace

// BM = 256: 128x64 wave tiles, 48 KiB LDS, 2 workgroups / CU.  BM = 128: 64x64 wave tiles, 32 KiB LDS, LB workgroups / CU
// (tile config 4: LB = 4, the occupancy experiment - do more resident workgroups hide the store drain of short-K products?)
template <int BM, int LB>
__global__ __launch_bounds__(256, LB) void gemm_nt_bk32_kernel(const GemmProblem p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 128, BK = 32, WM = BM / 2, WN = 64, TM = WM / 16, TN = 4;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    constexpr int DMA_PER_STAGE = (BM + BN) / 16 / 4;   // 6 LDS-DMA instructions per wave per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tiles_n = (p.N + BN - 1) / BN;
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int nt = (p.K + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage32<BM>(ra, m0, 0, p.lda, smem, wave, lane);
    stage32<BN>(rb, n0, 0, p.ldb, smem + A_BYTES, wave, lane);
    if (nt > 1) {
        stage32<BM>(ra, m0, BK, p.lda, smem + STAGE, wave, lane);
        stage32<BN>(rb, n0, BK, p.ldb, smem + STAGE + A_BYTES, wave, lane);
        wait_vm<DMA_PER_STAGE>();
    } else {
        wait_vm<0>();
    }
    asm volatile("s_barrier" ::: "memory");
    for (int it = 0; it < nt; ++it) {
        char* slot = smem + (it & 1) * STAGE;
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = frag32(slot, wm * WM + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = frag32(slot + A_BYTES, wn * WN + 16 * j, lane);
        if (it + 2 < nt) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stage32<BM>(ra, m0, (it + 2) * BK, p.lda, slot, wave, lane);
            stage32<BN>(rb, n0, (it + 2) * BK, p.ldb, slot + A_BYTES, wave, lane);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if (it + 1 < nt) {
            if (it + 2 < nt) wait_vm<DMA_PER_STAGE>(); else wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    // epilogue (experiment): straight from the MFMA layout, 8 B per lane (rows l & 15, columns 4 (l >> 4) .. + 3 of each 16 x 16)
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WM + 16 * i + (lane & 15);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + 16 * j + 4 * (lane >> 4);
            if (m < p.M && n < p.N) {
                f32x4 v = acc[i][j] * p.alpha;
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                *reinterpret_cast<uint2*>(C + (size_t)m * p.ldc + n) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// EXPERIMENT 2 (tile config 8): ONE 512-thread workgroup per CU, 256x128 tile, 64-deep K steps, THREE LDS slots of 48 KiB.
// Rationale (profiles/r01_e_gemm_decomposition_b64.txt, r01_f_pmc_gemm_persist_vs_pertile.txt): with two 128x128 workgroups
// per CU the L2 -> LDS fill alone takes as long as the MFMA + LDS work alone (~69 GB/s per CU for 64 FLOP per byte); one
// 256x128 tile per CU needs 87 FLOP per byte and keeps two K steps (96 KiB) in flight instead of 64 KiB.
namespace {

// rows r0.. of a k-contiguous [R][ld] array, k0..k0+63 -> image [BR][64] bf16 (128-B rows, swizzle as gemm_tile.h), NW waves
template <int BR, int NW>
__device__ __forceinline__ void stage64_nw(__amdgpu_buffer_rsrc_t rs, int r0, int k0, int ld, char* lds, int wave, int lane) {
    constexpr int PIECES = BR * 64 * 2 / 1024;      // 8 rows x 128 B per piece
#pragma unroll
    for (int jj = 0; jj < PIECES / NW; ++jj) {
        const int j = wave + NW * jj;
        const int r = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const uint32_t off = (uint32_t)(((r0 + r) * ld + k0 + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + j * 1024), 16, off, 0, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 frag64(const char* lds, int rbase, int ks, int lane) {
    const int r = rbase + (lane & 15);
    const int c = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const AS3 bf16x8*>((const AS3 char*)(lds) + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}

}  // namespace

__global__ __launch_bounds__(512, 1) void gemm_nt_256x128_8w_kernel(const GemmProblem p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 128, BK = 64, WM = 64, WN = 64, TM = 4, TN = 4, NS = 3;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;           // 32 + 16 = 48 KiB
    constexpr int DMA_PER_STAGE = (BM + BN) * BK * 2 / 1024 / 8;              // 6 per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                                 // 4 x 2 waves
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tiles_n = (p.N + BN - 1) / BN;
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int nt = p.K / BK;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kstep, int slot) {
        char* s = smem + slot * STAGE;
        stage64_nw<BM, 8>(ra, m0, kstep * BK, p.lda, s, wave, lane);
        stage64_nw<BN, 8>(rb, n0, kstep * BK, p.ldb, s + A_BYTES, wave, lane);
    };
    // prologue: up to three K steps in flight, the first one waited for
    const int pre = nt < NS ? nt : NS;
    for (int t = 0; t < pre; ++t) stage(t, t);
    if (pre == 3) wait_vm<2 * DMA_PER_STAGE>(); else if (pre == 2) wait_vm<DMA_PER_STAGE>(); else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    int slot = 0;
    for (int it = 0; it < nt; ++it) {
        const char* cur = smem + slot * STAGE;
        bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = frag64(cur, wm * WM + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[ks][j] = frag64(cur + A_BYTES, wn * WN + 16 * j, ks, lane);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0][j], af[0][i], acc[i][j], 0, 0, 0);
        const bool refill = it + NS < nt;
        if (refill) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stage(it + NS, slot);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[1][j], af[1][i], acc[i][j], 0, 0, 0);
        if (it + 1 < nt) {
            // K step it+1 must have landed: everything younger may stay in flight (steps it+2 and, if just issued, it+3)
            const int younger = (it + 2 < nt ? 1 : 0) + (refill ? 1 : 0);
            if (younger == 2) wait_vm<2 * DMA_PER_STAGE>(); else if (younger == 1) wait_vm<DMA_PER_STAGE>(); else wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        slot = slot == NS - 1 ? 0 : slot + 1;
    }
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WM + 16 * i + (lane & 15);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + 16 * j + 4 * (lane >> 4);
            if (m < p.M && n < p.N) {
                f32x4 v = acc[i][j] * p.alpha;
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                *reinterpret_cast<uint2*>(C + (size_t)m * p.ldc + n) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            }
        }
    }
}

int launch_gemm_big_nt(const GemmProblem& p, int cfg, hipStream_t stream) {
    BVC_REQUIRE(p.epi == EPI_BF16 && p.split_k == 1, "gemm bk32: EPI_BF16 without split-K only (experiment)");
    BVC_REQUIRE(p.K % 32 == 0 && p.N % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 8 == 0, "gemm bk32: K %% 32, N %% 8");
    if (cfg == 8) {
        BVC_REQUIRE(p.K % 64 == 0, "gemm 256x128 8-wave: K %% 64");
        const int t8 = ((p.M + 255) / 256) * ((p.N + 127) / 128);
        constexpr size_t lds8 = 3 * (256 + 128) * 64 * 2;
        static bool attr8 = false;
        if (!attr8) {
            BVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256x128_8w_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
            attr8 = true;
        }
        hipLaunchKernelGGL(gemm_nt_256x128_8w_kernel, dim3(t8), dim3(512), lds8, stream, p);
        BVC_CHECK_HIP(hipGetLastError());
        return BVC_OK;
    }
    const int bm = cfg == 3 ? 256 : 128;
    const int tiles = ((p.M + bm - 1) / bm) * ((p.N + 127) / 128);
    const size_t lds = 2 * (size_t)(bm + 128) * 32 * 2;
    if (cfg == 3) hipLaunchKernelGGL((gemm_nt_bk32_kernel<256, 2>), dim3(tiles), dim3(256), lds, stream, p);
    else if (cfg == 4) hipLaunchKernelGGL((gemm_nt_bk32_kernel<128, 4>), dim3(tiles), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((gemm_nt_bk32_kernel<128, 3>), dim3(tiles), dim3(256), lds, stream, p);
    BVC_CHECK_HIP(hipGetLastError());
    return BVC_OK;
}

}  // namespace bvc
#endif  // BVC_EXPERIMENTS
