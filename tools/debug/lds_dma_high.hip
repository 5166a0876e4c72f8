// Does an LDS-DMA (buffer_load ... lds) reach LDS addresses at and above 128 KiB on gfx950?  One 64-lane wave writes 1 KiB at each
// probed LDS address (M0 = address) and reads it back with ds_read; also reports which OTHER probed address (if any) received the data.
// build: hipcc --offload-arch=gfx950 -O2 -Ibaby-vision-curriculum_amd/csrc tools/debug/lds_dma_high.hip -o /tmp/lds_dma_high
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define AS3 __attribute__((address_space(3)))
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}
__global__ void probe(const uint32_t* src, const uint32_t* addrs, int naddr, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const uint32_t base = (uint32_t)(size_t)((AS3 char*)smem);
    for (int a = 0; a < naddr; ++a) {
        // clear every probed kilobyte
        for (int b = 0; b < naddr; ++b) {
            AS3 uint32_t* p = (AS3 uint32_t*)((AS3 char*)smem + addrs[b]);
            for (int i = lane; i < 256; i += 64) p[i] = 0u;
        }
        __syncthreads();
        glds16(make_rsrc(src + 256 * a, 1024), (uint32_t)lane * 16u, __builtin_amdgcn_readfirstlane(base + addrs[a]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int b = 0; b < naddr; ++b) {
            AS3 uint32_t* p = (AS3 uint32_t*)((AS3 char*)smem + addrs[b]);
            uint32_t ok = 1;
            for (int i = 0; i < 4; ++i) ok &= (p[lane * 4 + i] == src[256 * a + lane * 4 + i]);
            const unsigned long long m = __ballot(ok);
            if (lane == 0) out[a * naddr + b] = (m == ~0ull) ? 1u : 0u;
        }
        __syncthreads();
    }
    if (lane == 0) out[naddr * naddr] = base;
}
int main() {
    std::vector<uint32_t> addrs = {0, 65536, 130048, 131072, 135168, 147456 - 1024, 163840 - 1024};
    const int n = (int)addrs.size();
    std::vector<uint32_t> src(256 * n);
    for (size_t i = 0; i < src.size(); ++i) src[i] = 0x9e3779b9u * (uint32_t)(i + 1);
    uint32_t *dsrc, *daddr, *dout;
    hipMalloc(&dsrc, src.size() * 4); hipMalloc(&daddr, n * 4); hipMalloc(&dout, (n * n + 1) * 4);
    hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(daddr, addrs.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 163840, 0, dsrc, daddr, n, dout);
    hipError_t e = hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(e));
    std::vector<uint32_t> out(n * n + 1);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    printf("LDS base of the dynamic array: %u\n", out[n * n]);
    for (int a = 0; a < n; ++a) {
        printf("DMA to LDS address %6u: landed at", addrs[a]);
        bool any = false;
        for (int b = 0; b < n; ++b) if (out[a * n + b]) { printf(" %u", addrs[b]); any = true; }
        printf("%s\n", any ? "" : " NONE of the probed addresses");
    }
    return 0;
}
