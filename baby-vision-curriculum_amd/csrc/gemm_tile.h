// Device helpers shared by the GEMM kernels (gemm.hip, gemm_persist.hip): LDS swizzles, HBM -> LDS staging by LDS-DMA and
// LDS -> MFMA fragment reads for 64-deep K steps.  Internal to libbvc_hip.so.
#pragma once
#include "gemm.h"

namespace bvc {

#ifndef AS3
#define AS3 __attribute__((address_space(3)))
#endif

// Experiment hooks (BVC_GEMM_DEBUG bits inside the kernels, per-launch environment switches, the 32-deep-K kernels of
// experiments/gemm_big.hip behind tile configs 3-5 / 8) exist only in a -DBVC_EXPERIMENTS build (BVC_EXTRA_HIPCC_FLAGS=-DBVC_EXPERIMENTS,
// used by tools/ab/gemm_dbg.py, tools/ab/gemm_ksweep.py and the same-process A/Bs of tools/ab/microbench.py); the product library
// compiles them out.
#ifdef BVC_EXPERIMENTS
#define BVC_DBG(g, bits) ((g).dbg & (bits))
#define BVC_EXP_ENV(name) getenv(name)
#else
#define BVC_DBG(g, bits) 0
#define BVC_EXP_ENV(name) ((const char*)nullptr)
#endif

struct GemmGroup {
    int nprob;
    int tile_start[kMaxGroup + 1];
    int panel[kMaxGroup];          // column tiles per panel of the tile walk (0 = the legacy walk, A/B only); see pick_panel
    int dbg;                       // BVC_GEMM_DEBUG experiments (tools/ab/gemm_dbg.py): 1 = drop the bf16 stores, 2 = stagger odd slots,
                                   // 8 = no B-operand refills, 16 = no MFMAs, 32 = no refills at all (results are garbage for 8/16/32)
    // Balanced weight-gradient walk of gemm8.hip (0 = off): `bal_units` full-length units (tile x K split) fill fewer workgroups than the
    // grid has, so every unit gives up the last K tiles of its tile: the splits cover `bal_lb` K tiles each, and the remainder of
    // each of the `bal_tiles` tiles becomes a short "tail" unit; the otherwise idle workgroups take a few tails each (launch_gemm8).
    int bal_units, bal_lb, bal_tiles;
    int stagger;                   // row epilogues (gemm8.hip EC 4 / 5): estimated time of one unit in 64-cycle sleep quanta, 0 = all workgroups start together
    int accum;                     // tile config 13: every unit ADDS into C (f32 atomics even when K is not split) - what lets an unsplit group be balanced
    GemmProblem prob[kMaxGroup];
};

// ------------------------------------------------------------------ swizzles (16-byte chunk index)
// k-contiguous image [rows][64] bf16, 128-B rows, read by ds_read_b128 (16 lanes = 16 rows, same chunk)
__device__ __forceinline__ int swz_rows(int r) { return (r >> 1) & 7; }
// transposed images [64 k][BR] bf16, read by ds_read_b64_tr_b16 (a 32-lane half = 8 k-rows x 32 B)
template <int BR>
__device__ __forceinline__ int swz_tr(int k) {
    if constexpr (BR == 128) return ((k & 3) | (((k >> 3) & 1) << 2)) << 1;   // 256-B rows
    else return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1;                  // 128-B rows
}

// ------------------------------------------------------------------ HBM -> LDS staging of one operand tile
// Non-transposed: rows r0..r0+BR-1 (output dim), k0..k0+63 of a [R][ld] array -> image [BR][64].
// Transposed:     k rows k0..k0+63, columns r0..r0+BR-1 of a [Kc][ld] array    -> image [64][BR].
template <int BR, bool T>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, int r0, int k0, int ld, char* lds,
                                           int wave, int lane) {
    constexpr int PIECES = BR * 64 * 2 / 1024;
#pragma unroll
    for (int jj = 0; jj < PIECES / 4; ++jj) {
        const int j = wave + 4 * jj;   // wave-uniform piece index; piece j = LDS bytes [1024 j, 1024 j + 1024)
        uint32_t off;
        if constexpr (!T) {
            const int r = 8 * j + (lane >> 3);
            const int c = (lane & 7) ^ swz_rows(r);
            off = (uint32_t)(((r0 + r) * ld + k0 + c * 8) * 2);
        } else if constexpr (BR == 128) {
            const int kr = 4 * j + (lane >> 4);
            const int c = (lane & 15) ^ swz_tr<128>(kr);
            off = (uint32_t)(((k0 + kr) * ld + r0 + c * 8) * 2);
        } else {
            const int kr = 8 * j + (lane >> 3);
            const int c = (lane & 7) ^ swz_tr<64>(kr);
            off = (uint32_t)(((k0 + kr) * ld + r0 + c * 8) * 2);
        }
        glds16(rs, off, (uint32_t)(size_t)((AS3 char*)lds) + (uint32_t)j * 1024u);
    }
}

// ------------------------------------------------------------------ LDS -> MFMA fragment
// Returns, for lane l, the 8 bf16  X[out = rbase + (l & 15)][k = 32 ks + 8 (l >> 4) + 0..7].
template <int BR, bool T>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int rbase, int ks, int lane) {
    if constexpr (!T) {
        const int r = rbase + (lane & 15);
        const int c = 4 * ks + (lane >> 4);
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(
            (const __attribute__((address_space(3))) char*)(lds) + r * 128 + ((c ^ swz_rows(r)) << 4));
    } else {
        // lane 4q+p of each 16-lane group supplies the address of k-row q, columns 4p..4p+3;
        // lane i of the group receives column i of the 4 k-rows (hardware transpose).
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int k0 = 32 * ks + 8 * g + q, k1 = k0 + 4;
        const int chunk = (rbase >> 3) + (p >> 1);
        const int within = (p & 1) * 8;
        const char* a0 = lds + k0 * (BR * 2) + ((chunk ^ swz_tr<BR>(k0)) << 4) + within;
        const char* a1 = lds + k1 * (BR * 2) + ((chunk ^ swz_tr<BR>(k1)) << 4) + within;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a1));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Tile id -> (row tile, column tile) of the panel walk: columns in panels of G tiles, rows fastest-but-one (see gemm_kernel).
__device__ __forceinline__ void tile_of(int t, int tiles_m, int tiles_n, int G, int& tm, int& tn) {
    const int full = (tiles_n / G) * G * tiles_m;
    if (t < full) {
        const int pn = t / (G * tiles_m), w = t - pn * G * tiles_m;
        tm = w / G; tn = pn * G + (w - tm * G);
    } else {
        const int r = tiles_n % G, w = t - full;
        tm = w / r; tn = (tiles_n - r) + (w - tm * r);
    }
}

}  // namespace bvc
