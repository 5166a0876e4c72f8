// Attention launchers (internal to libbvc_hip.so); head_dim 64 or 32 (D = head_dim * H).
// softmax_scale = 0 means head_dim^-1/2; a caller that zero-pads a narrower head (JEPA ViT-L predictor: 24 -> 32) passes
// the scale of the TRUE head width - the padded lanes contribute nothing to q.k, the context or any gradient.
#pragma once
#include "common.h"

namespace bvc {
// qkv bf16 [B*N][3D] -> ctx bf16 [B*N][D], lse f32 [B*H][N] (log2 units)
int launch_attn_fwd(const bf16_t* qkv, bf16_t* ctx, float* lse, int B, int N, int H, int head_dim, hipStream_t stream,
                    float softmax_scale = 0.f);
// dctx bf16 [B*N][D] (+ saved qkv, ctx, lse) -> dqkv bf16 [B*N][3D]; delta f32 [B*H][N] is scratch
int launch_attn_bwd(const bf16_t* qkv, const bf16_t* ctx, const bf16_t* dctx, const float* lse, float* delta,
                    bf16_t* dqkv, int B, int N, int H, int head_dim, hipStream_t stream, float softmax_scale = 0.f, int parts = 3);
// parts: bit 0 = the dQ kernel (also writes delta = rowsum(dO * O)), bit 1 = the dK / dV kernel (reads delta); 3 = the backward
}  // namespace bvc
