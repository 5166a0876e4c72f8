// Micro-benchmark (round 5): does the SHAPE of a wave-instruction matter for an HBM-bound read-modify-write of f32 [M][384] rows?
//   pattern 0 "mfma":  lane l touches row (l & 15), 16 bytes at column 4 (l >> 4) + 16 j  -> 16 rows x 64 B per instruction
//                      (what an epilogue sees straight out of the MFMA accumulator layout)
//   pattern 1 "rows":  lane l of instruction t touches 16-byte chunk (t 64 + l) of a 16 x 96-float block row by row -> 2.67 rows x 384 B
//   pattern 2 "full":  one wave per 384-float row: 64 lanes x 16 B + 32 lanes x 16 B (1536 B contiguous)
// Each reads x and r, writes y = x + r (f32) and a bf16 copy: the traffic of the residual + LayerNorm epilogue without the GEMM.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/rowpattern tools/micro/rowpattern.hip ; run: /tmp/rowpattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline uint32_t pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    b2 v = __builtin_convertvector(f2{a, b}, b2);
    return *reinterpret_cast<uint32_t*>(&v);
}

template <int PAT>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, const float* __restrict__ r, float* __restrict__ y, uint16_t* __restrict__ yb, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave & 3, wm = wave >> 2;
    // a workgroup = 128 rows x 384 columns like the GEMM unit; wave (wm, wn) owns rows 32 i + 16 wm + (0..15), columns 96 wn ..
    for (int blk = blockIdx.x; blk * 128 < M; blk += gridDim.x) {
        for (int i = 0; i < 4; ++i) {
            const int row0 = blk * 128 + 32 * i + 16 * wm;
            f32x4 a[6], b[6];
            size_t off[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                int row, col;
                if (PAT == 0) { row = lane & 15; col = 4 * (lane >> 4) + 16 * j; }
                else { const int idx = j * 64 + lane; row = idx / 24; col = 4 * (idx % 24); }
                off[j] = (size_t)(row0 + row) * 384 + 96 * wn + col;
                a[j] = *reinterpret_cast<const f32x4*>(x + off[j]);
                b[j] = *reinterpret_cast<const f32x4*>(r + off[j]);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x4 v = a[j] + b[j];
                *reinterpret_cast<f32x4*>(y + off[j]) = v;
                *reinterpret_cast<uint2*>(yb + off[j]) = uint2{pack2(v[0], v[1]), pack2(v[2], v[3])};
            }
        }
    }
}

__global__ __launch_bounds__(256) void kfull(const float* __restrict__ x, const float* __restrict__ r, float* __restrict__ y, uint16_t* __restrict__ yb, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const size_t base = (size_t)row * 384;
        f32x4 a0 = *reinterpret_cast<const f32x4*>(x + base + 4 * lane), b0 = *reinterpret_cast<const f32x4*>(r + base + 4 * lane);
        f32x4 a1 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        if (lane < 32) { a1 = *reinterpret_cast<const f32x4*>(x + base + 256 + 4 * lane); b1 = *reinterpret_cast<const f32x4*>(r + base + 256 + 4 * lane); }
        const f32x4 v0 = a0 + b0, v1 = a1 + b1;
        *reinterpret_cast<f32x4*>(y + base + 4 * lane) = v0;
        *reinterpret_cast<uint2*>(yb + base + 4 * lane) = uint2{pack2(v0[0], v0[1]), pack2(v0[2], v0[3])};
        if (lane < 32) {
            *reinterpret_cast<f32x4*>(y + base + 256 + 4 * lane) = v1;
            *reinterpret_cast<uint2*>(yb + base + 256 + 4 * lane) = uint2{pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
        }
    }
}

int main() {
    const int M = 401408;
    const size_t n = (size_t)M * 384;
    float *x, *r, *y;
    uint16_t* yb;
    hipMalloc(&x, n * 4); hipMalloc(&r, n * 4); hipMalloc(&y, n * 4); hipMalloc(&yb, n * 2);
    hipMemset(x, 0, n * 4); hipMemset(r, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)n * (4 + 4 + 4 + 2);
    for (int grid : {256, 512, 1024, 2048}) {
        for (int pat = 0; pat < 3; ++pat) {
            float best = 1e30f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, x, r, y, yb, M);
                else if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, x, r, y, yb, M);
                else hipLaunchKernelGGL(kfull, dim3(grid * 4), dim3(256), 0, 0, x, r, y, yb, M);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it > 0 && ms < best) best = ms;
            }
            printf("grid %5d pattern %d (%s): %8.1f us  %6.2f TB/s\n", grid, pat, pat == 0 ? "mfma 16 rows x 64 B" : pat == 1 ? "rows 2.67 x 384 B" : "one wave per row", best * 1e3, bytes / best / 1e9);
        }
    }
    return 0;
}
