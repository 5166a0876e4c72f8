// Shared device/host helpers for libbvc_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}
// two floats -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (separate scalar casts cost 4 VALU per pair)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const bf16x2_t b = __builtin_convertvector(f32x2{lo, hi}, bf16x2_t);
    return *reinterpret_cast<const uint32_t*>(&b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU (the reference's hidden_act="gelu", HF:299-308) and its derivative.
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the result): one v_rcp, one
// v_exp and five FMAs instead of libm erff's ~40 VALU - the GELU epilogues were doubling the fc1 / dX-fc2 GEMMs.
// gelu and gelu' share the exponential: with z = x / sqrt(2),  exp(-z^2) = exp(-x^2 / 2).
// The FORWARD epilogue computes both and saves gelu'(pre) in place of the pre-activation (round 2): the backward product's epilogue is
// then one multiply per element instead of ~14 vector instructions + 2 transcendentals that nothing hid on the one-workgroup-per-CU kernel.
// Two elements at a time: the polynomial, the squares and the final combinations are v_pk_fma_f32 / v_pk_mul_f32 (one issue for
// two elements), only the reciprocal and the exponential stay scalar - 18 instructions per PAIR instead of ~16 per element.
// The GELU epilogues of the decoder's K = 384 products cost as much as their whole K loop (profiles/r02_c_gemm8_store_cost.txt).
// gelu(x) = 0.5 (x + |x| erf(|x| / sqrt 2)) (x erf(x / sqrt 2) is even), gelu'(x) = 0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2/2) / sqrt(2 pi).
struct GeluParts2 { f32x2 erf_abs, e, ax; };   // erf(|z|), exp(-x^2/2), |x|
__device__ __forceinline__ GeluParts2 gelu_parts2(f32x2 x) {
    GeluParts2 g;
    g.ax.x = __builtin_fabsf(x.x); g.ax.y = __builtin_fabsf(x.y);
    const f32x2 az = g.ax * 0.70710678118654752f;
    const f32x2 d = __builtin_elementwise_fma(az, f32x2{0.3275911f, 0.3275911f}, f32x2{1.0f, 1.0f});
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
    f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
    p = p * t;
    const f32x2 xx = x * x * -0.72134752044448170f;      // exp(-x^2/2) = 2^(-x^2/2 * log2 e)
    g.e.x = __builtin_amdgcn_exp2f(xx.x); g.e.y = __builtin_amdgcn_exp2f(xx.y);
    g.erf_abs = __builtin_elementwise_fma(-p, g.e, f32x2{1.0f, 1.0f});
    return g;
}
// gelu(x) and gelu'(x) of two elements from one set of parts (the forward epilogue stores both: the backward product then
// multiplies by the stored derivative instead of re-deriving it from the pre-activation with a reciprocal and an exponential per element)
__device__ __forceinline__ void gelu_and_grad2(f32x2 x, f32x2& act, f32x2& grad) {
    const GeluParts2 g = gelu_parts2(x);
    act = __builtin_elementwise_fma(g.ax, g.erf_abs, x) * 0.5f;
    f32x2 s;
    s.x = copysignf(g.erf_abs.x, x.x); s.y = copysignf(g.erf_abs.y, x.y);
    const f32x2 phi = __builtin_elementwise_fma(s, f32x2{0.5f, 0.5f}, f32x2{0.5f, 0.5f});
    grad = __builtin_elementwise_fma(x * 0.39894228040143268f, g.e, phi);
}
// v[0 .. 2n) -> gelu'(v) in place, gelu(v) to act[].  All pairs advance STAGE BY STAGE (round 4): written pair after pair, hipcc
// keeps each pair's chain of dependent packed operations together and pads it with an `s_nop` per link (8 per pair) - with the
// stages of several pairs side by side the links of one pair fill the gaps of the others.  Same operations per element as
// gelu_and_grad2, same results bit for bit.
template <int N2>
__device__ __forceinline__ void gelu_split(float (&v)[N2], float (&act)[N2]) {
    constexpr int P = N2 / 2;
    f32x2 x[P], ax[P], t[P], pl[P], ex[P], er[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        x[i] = f32x2{v[2 * i], v[2 * i + 1]};
        ax[i].x = __builtin_fabsf(x[i].x); ax[i].y = __builtin_fabsf(x[i].y);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const f32x2 az = ax[i] * 0.70710678118654752f;
        const f32x2 d = __builtin_elementwise_fma(az, f32x2{0.3275911f, 0.3275911f}, f32x2{1.0f, 1.0f});
        t[i].x = __builtin_amdgcn_rcpf(d.x); t[i].y = __builtin_amdgcn_rcpf(d.y);
        const f32x2 xx = x[i] * x[i] * -0.72134752044448170f;      // exp(-x^2/2) = 2^(-x^2/2 * log2 e)
        ex[i].x = __builtin_amdgcn_exp2f(xx.x); ex[i].y = __builtin_amdgcn_exp2f(xx.y);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) pl[i] = __builtin_elementwise_fma(t[i], f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
#pragma unroll
    for (int i = 0; i < P; ++i) pl[i] = __builtin_elementwise_fma(pl[i], t[i], f32x2{1.421413741f, 1.421413741f});
#pragma unroll
    for (int i = 0; i < P; ++i) pl[i] = __builtin_elementwise_fma(pl[i], t[i], f32x2{-0.284496736f, -0.284496736f});
#pragma unroll
    for (int i = 0; i < P; ++i) pl[i] = __builtin_elementwise_fma(pl[i], t[i], f32x2{0.254829592f, 0.254829592f});
#pragma unroll
    for (int i = 0; i < P; ++i) pl[i] = pl[i] * t[i];
#pragma unroll
    for (int i = 0; i < P; ++i) er[i] = __builtin_elementwise_fma(-pl[i], ex[i], f32x2{1.0f, 1.0f});       // erf(|z|)
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const f32x2 a = __builtin_elementwise_fma(ax[i], er[i], x[i]) * 0.5f;
        f32x2 sg;
        sg.x = copysignf(er[i].x, x[i].x); sg.y = copysignf(er[i].y, x[i].y);
        const f32x2 phi = __builtin_elementwise_fma(sg, f32x2{0.5f, 0.5f}, f32x2{0.5f, 0.5f});
        const f32x2 g = __builtin_elementwise_fma(x[i] * 0.39894228040143268f, ex[i], phi);
        act[2 * i] = a.x; act[2 * i + 1] = a.y; v[2 * i] = g.x; v[2 * i + 1] = g.y;
    }
}
// buffer resource over [base, base+bytes): out-of-range lanes of a buffer load return 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS at the wave-uniform byte address lds_addr) as INLINE ASM: hipcc must not
// know that these loads write LDS.  With the builtin it orders every `ds_read_b64_tr_b16` behind an `s_waitcnt vmcnt(0)` (the
// transposed-read intrinsic may alias anything), which drains the whole prefetch once per K step - measured 4x on the weight
// gradient products of gemm8.hip, and present in every NN / TN instantiation of gemm_kernel / gemm_persist_kernel and in the three attention kernels (the next
// tile's K/V prefetch was drained before the current tile's P V product) before round 2.  The data is ordered for the readers by the counted vmcnt + barrier protocol of the kernel alone.
// M0 (the DMA's LDS base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}
// the same piece with a wave-uniform byte offset in an SGPR (soffset) added to the per-lane one: one per-lane address
// register then serves several images of the same rows (the K and the V columns of a qkv row, attention.hip)
__device__ __forceinline__ void glds16s(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory");
}
// the 4-byte form (64 lanes x 4 B -> 256 B of LDS): per-row statistics next to an operand tile
__device__ __forceinline__ void glds4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}

// ---------------------------------------------------------------- host side
#include <string>
namespace bvc {
void set_error(const char* fmt, ...);
const char* last_error();
}
#include "../../include/bvc.h"
#define BVC_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            bvc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return BVC_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)
#define BVC_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            bvc::set_error(__VA_ARGS__);       \
            return BVC_ERR_INVALID;            \
        }                                      \
    } while (0)
