// EXPERIMENT (round 1, not on the product path yet): NT GEMM with a 256x128 workgroup tile, 32-deep K steps and 128x64 wave
// tiles.  Why: with 64x64 wave tiles every MFMA flop needs 1/32 B of LDS reads, i.e. 128 B/clk/CU at MFMA peak - exactly the LDS
// peak - and tools/gemm_ksweep.py measures 816 TFLOP/s marginal (33 % of peak) for the 128x128 kernel.  A 128x64 wave tile
// needs 1/42.7 B/flop (75 % of the LDS peak at MFMA peak); 32-deep K steps keep two 24 KiB stages = 48 KiB per workgroup so
// that two workgroups still share a CU.  Reached through bvc_op_gemm(tile_cfg = 3): NT layout, one problem, EPI_BF16 (+ bias).
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

int launch_gemm_big_nt(const GemmProblem& p, int cfg, hipStream_t stream) {
    BVC_REQUIRE(p.epi == EPI_BF16 && p.split_k == 1, "gemm bk32: EPI_BF16 without split-K only (experiment)");
    BVC_REQUIRE(p.K % 32 == 0 && p.N % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.ldc % 8 == 0, "gemm bk32: K %% 32, N %% 8");
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
