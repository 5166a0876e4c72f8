// HBM-bound row / element kernels (internal to libbvc_hip.so).
#pragma once
#include "common.h"

namespace bvc {

// logical row m -> physical row (m / rin) * rout + roff + m % rin; rin <= 0 means identity.
// Used to address "the last nmask tokens of every clip" without copying them out.
struct RowMap { int rin, rout, roff; };
static inline RowMap identity_rows() { return RowMap{0, 0, 0}; }

// clip [B][T][C][H][W]; tubes of ts frames x ps x ps pixels
struct PatchGeom { int T, C, H, W, ts, ps; };

// Where the pixels come from: f32 already normalised by the loader, or the loader's uint8 frames normalised on the fly as
// (u / 255 - mean[c]) / std[c]  - the arithmetic of ToTensor + Normalize (homeview.py:221-230), same operation order, so the
// two forms give bit-identical values while the uint8 one moves a quarter of the bytes over PCIe and out of HBM.
struct PixelSrc {
    const void* ptr;
    int is_u8;
    float mean[4], stdv[4];
};
static inline PixelSrc pixels_f32(const float* p) { return PixelSrc{p, 0, {0, 0, 0, 0}, {1, 1, 1, 1}}; }

// gamma == nullptr: no affine.  y (bf16) and/or y32 (f32) receive the result; mean / rstd may be null.
int launch_ln_fwd(const float* x, RowMap rm, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                  int M, int D, float eps, hipStream_t s, float* y32 = nullptr);
// `part` = scratch of ln_bwd_workspace_floats(M, D) floats (per-workgroup dgamma / dbeta partials)
size_t ln_bwd_workspace_floats(int M, int D);
size_t ln_bwd_workspace_floats_upto(int Mmax, int D);
int launch_ln_bwd(const bf16_t* dy, const float* x, RowMap rm, const float* mean, const float* rstd, const float* gamma,
                  float* dres, int accumulate, bf16_t* dres_bf, float* dgamma, float* dbeta, float* part, int M, int D, hipStream_t s);
// dgamma[c] += sum_b part[b][0][c], dbeta[c] += sum_b part[b][1][c] over nblk partial rows of [2][D] floats (what ln_bwd and the fused
// LayerNorm-backward epilogue of gemm8.hip leave behind)
int launch_ln_param_reduce(const float* part, int nblk, int D, float* dgamma, float* dbeta, hipStream_t s);
int launch_colsum_bf16(const bf16_t* X, int M, int N, int ld, float alpha, float* out, hipStream_t s);
int launch_colsum_bf16_scaled(const bf16_t* X, int M, int N, int ld, float alpha, const float* alpha_dev, float* out, hipStream_t s);
int launch_colsum_f32(const float* X, RowMap rm, int M, int D, float* out, hipStream_t s);
int launch_token_mean(const float* x, int B, int N, int D, float* out, hipStream_t s);
int launch_token_mean_bwd(const float* dmean, int B, int N, int D, float* dx, hipStream_t s);
int launch_cast_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
int launch_gather_rows_bf16(const float* in, RowMap rm, bf16_t* out, int M, int D, hipStream_t s);
int launch_mask_index(const uint8_t* mask, int B, int L, int nvis, int nmask, int* vis_idx, int* msk_idx, int* status, hipStream_t s);
int launch_gather_patches(PixelSrc clip, const int* vis_idx, bf16_t* A, int B, int nvis, PatchGeom pg, hipStream_t s);
int launch_labels(PixelSrc clip, const int* msk_idx, float* labels, int B, int nmask, PatchGeom pg, int norm_pix, hipStream_t s);
int launch_fill_masked(float* xfull, const float* mask_token, const float* pos, const int* msk_idx, int B, int L, int nvis,
                       int nmask, int D, hipStream_t s);
int launch_sgd_step(float* p, float* g, float* buf, size_t n, float lr, float momentum, float dampening, float wd, int nesterov,
                    int first, int maximize, const float* grad_scale, const float* found_inf, int write_grad, bf16_t* shadow,
                    hipStream_t s);
int launch_adam_prep(float* state, double lr, double beta1, double beta2, const float* found_inf, hipStream_t s);
int launch_adam_step(float* p, float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps, double wd,
                     int decoupled, int maximize, const float* state, const float* grad_scale, const float* found_inf, int write_grad,
                     bf16_t* shadow, hipStream_t s);
int launch_sgd_step_segments(float* p, float* g, float* buf, int64_t n, const int64_t* seg_start, const int* seg_group, const int* blk_seg,
                             int nseg, const bvc_sgd_groups* groups, const float* grad_scale, const float* found_inf, int write_grad,
                             bf16_t* shadow, hipStream_t s);
int launch_adam_step_segments(float* p, float* g, float* m, float* v, int64_t n, const int64_t* seg_start, const int* seg_group,
                              const int* blk_seg, int nseg, const bvc_adam_groups* groups, float* state, double* hyper_dev,
                              const float* grad_scale, const float* found_inf, int write_grad, bf16_t* shadow, hipStream_t s);
int launch_pad_heads(const bf16_t* wqkv, const float* bqkv, const bf16_t* wo, bf16_t* wqkv_p, float* bqkv_p, bf16_t* wo_p, int D, int H,
                     int hd, int hdp, hipStream_t s);
int launch_unpad_head_grads(const float* gwqkv_p, const float* gbqkv_p, const float* gwo_p, float* gwqkv, float* gbqkv, float* gwo, int D,
                            int H, int hd, int hdp, hipStream_t s);
int launch_nonfinite_check(const float* x, size_t n, float* found_inf, hipStream_t s);
int launch_row_normalize(const float* f, bf16_t* fn, float* inv, int n, int p, float eps, hipStream_t s);
int launch_row_normalize_bwd(const float* f, const float* inv, const float* dfn, float* df, int n, int p, hipStream_t s);
int launch_nce_finalize(const float* partial, int ntiles, float inv_t, double npos, float* loss, float* stats, hipStream_t s);
int launch_target_select(const float* h, const int* idx, float* out, int nsets, int B, int Np, int L, int D, float eps, hipStream_t s);
int launch_pred_assemble(const float* xe, const float* mask_token, const float* pos, const int* idx_pred, float* X, int nsets, int B,
                         int Nc, int Np, int D, hipStream_t s);
int launch_pred_ctx_grad(const float* dX, bf16_t* dxe, int nsets, int B, int Nc, int Np, int D, hipStream_t s);
int smooth_l1_blocks(size_t n);
int launch_smooth_l1_fwd(const float* z, const float* h, size_t n, float* partial, float* loss, hipStream_t s);
int launch_smooth_l1_bwd(const float* z, const float* h, const float* gout, size_t n, float* dz, hipStream_t s);
int launch_ema(float* k, const float* q, size_t n, float m, hipStream_t s);
int launch_iota_mod(int* idx, int n, int L, hipStream_t s);
int launch_loss_finalize(const float* partial, int n, double count, const int* status, float* loss, hipStream_t s);

}  // namespace bvc
