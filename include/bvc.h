/* libbvc_hip.so -- C ABI of the MI355X-native (gfx950) self-supervised video pre-training step.
 *
 * The reference (ssheybani/baby-vision-curriculum) is pure Python and has no FFI of its own: its
 * entry points call a model object.  This header is the boundary a maintainer binds (ctypes stub in
 * INTEGRATION.md) so that object's arithmetic runs as hand-written HIP.  Each entry point cites the
 * reference interface it replaces; paths are relative to the reference checkout, "HF" is
 * transformers/models/videomae/modeling_videomae.py (5.15.0), which the reference instantiates at
 * pretraining/generative/pretrain_videomae.py:61-64.
 *
 * Conventions
 *  - every function returns 0 (BVC_OK) or a negative bvc_status; bvc_last_error() returns a
 *    thread-local message.  No C++ exception crosses this boundary, HIP errors are translated.
 *  - all pointers named *_dev are device pointers borrowed from the caller (torch tensors); the
 *    library never owns parameters, gradients, inputs or outputs.  Workspaces (saved activations,
 *    bf16 weight copies) belong to the context.
 *  - kernels are enqueued on the caller's HIP stream (`stream`, a hipStream_t passed as void*); the
 *    calls do not synchronise.  One host thread drives a context at a time (one process per GPU,
 *    as the reference's mp.spawn launcher does, pretrain_videomae.py:509-513).
 */
#ifndef BVC_H
#define BVC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVC_OK 0
#define BVC_ERR_INVALID (-1) /* bad argument / unsupported shape */
#define BVC_ERR_HIP (-2)     /* a HIP runtime call failed */
#define BVC_ERR_STATE (-3)   /* call order violated (e.g. backward without a matching forward) */

const char* bvc_last_error(void);
/* Library build info: "gfx950;<git-less build tag>" */
const char* bvc_version(void);
/* Process-wide kernel-selection switches (the reference has no counterpart: ATen picks its cuBLAS kernels by itself).  They
 * exist for the parity tests and the same-process A/B tools, never change results beyond the documented tolerances, and
 * replace the environment variables earlier builds read:
 *   "gemm8"       0 (default) = the measured selection; 1 = the 256-row persistent GEMM for every product it can take,
 *                 whatever its size (runs the kernel set of the 256-clip benchmark at oracle-sized batches); -1 = never
 *   "dw_overlap"  1 = a layer's grouped weight-gradient launch runs on the context's side stream (default 0)
 *   "row_ln"      the LayerNorms of 384-wide stacks (VideoMAE decoder, JEPA predictor) inside the epilogues of the products next to
 *                 them (BVC_EPI_RESID_LN / BVC_EPI_DLN): 0 (default) = the measured selection, 1 = whenever the shapes allow,
 *                 -1 = never (separate LayerNorm passes)
 * bvc_get_option returns the value, or BVC_ERR_INVALID for an unknown name. */
int bvc_set_option(const char* name, int value);
int bvc_get_option(const char* name);

/* ------------------------------------------------------------------------------------------------
 * VideoMAE pre-training step.
 * Replaces transformers.VideoMAEConfig as built by get_config(), pretrain_videomae.py:43-58.        */
typedef struct bvc_videomae_config {
    int image_size, patch_size, num_channels, num_frames, tubelet_size;
    int hidden_size, num_hidden_layers, num_attention_heads, intermediate_size;
    int decoder_hidden_size, decoder_num_hidden_layers, decoder_num_attention_heads, decoder_intermediate_size;
    float layer_norm_eps;   /* 1e-12: every VideoMAELayer LayerNorm (HF:336-337) */
    float decoder_norm_eps; /* 1e-5 : decoder.norm (HF:484) */
    int norm_pix_loss;      /* 1 in the reference (pretrain_videomae.py:57) */
} bvc_videomae_config;

typedef struct bvc_ctx bvc_ctx;

/* Flat parameter layout.  Parameters and gradients live in ONE contiguous fp32 buffer each (forward
 * order: patch embed | encoder layers | encoder_to_decoder, mask_token | decoder layers | decoder.norm
 * | decoder.head), so casts, the optimiser and the data-parallel all-reduce are single large
 * transfers.  Entry `index` names a state-dict key of VideoMAEForPreTraining (the 264 keys that
 * model.module.state_dict() returns at pretrain_videomae.py:76) and where it sits in the buffer. */
int bvc_videomae_param_count(const bvc_videomae_config* cfg);
int64_t bvc_videomae_param_numel(const bvc_videomae_config* cfg);
int bvc_videomae_param_info(const bvc_videomae_config* cfg, int index, char* name, int name_cap, int64_t* offset,
                            int64_t* numel, int* ndim, int64_t shape[5]);

/* Allocates workspaces for up to `max_batch` clips with `num_masked` masked tokens per clip. */
int bvc_videomae_create(const bvc_videomae_config* cfg, int max_batch, int num_masked, bvc_ctx** out);
void bvc_videomae_destroy(bvc_ctx* ctx);

/* Replaces `outputs = xmodel(inputs, bool_masked_pos=m); outputs.loss` (pretrain_videomae.py:301-302,
 * i.e. VideoMAEForPreTraining.forward, HF:531-671).
 *   pixels_dev  f32 [batch][num_frames][channels][H][W]   (loader output, homeview.py:218-231)
 *   mask_dev    u8  [batch][seq_len], 1 = masked; every row must hold exactly num_masked ones
 *   params_dev  f32 flat parameter buffer (layout above)
 *   loss_dev    f32 scalar  (mean squared error over masked, per-patch-normalised pixels)
 *   logits_dev  optional f32 [batch][num_masked][patch_dim] (outputs.logits), may be NULL
 * A row with a different number of ones makes the loss NaN (flag checked on the device, no host sync). */
int bvc_videomae_forward(bvc_ctx* ctx, const float* pixels_dev, const uint8_t* mask_dev, int batch,
                         const float* params_dev, float* loss_dev, float* logits_dev, void* stream);

/* Called on the host, from inside bvc_videomae_backward, after every kernel that finalises the
 * gradient range [offset, offset+count) of the flat gradient buffer has been enqueued on `stream`.
 * The data-parallel wrapper records an event there and starts the RCCL all-reduce of that range on
 * its communication stream (what DistributedDataParallel's bucket hooks do at
 * pretrain_videomae.py:180-181,312).  Ranges arrive tail-first (decoder head ... patch embed). */
typedef void (*bvc_bucket_fn)(int64_t offset, int64_t count, void* user);

/* Pixel source of the *_px entry points.  BVC_PIXELS_F32: f32 clips already normalised by the loader (what the reference's
 * dataloader hands over, homeview.py:218-231).  BVC_PIXELS_U8: the loader's uint8 frames, normalised while they are read as
 * (u / 255 - mean[c]) / std[c] - ToTensor + Normalize with the same operation order, hence bit-identical values - so that
 * a quarter of the bytes cross PCIe and HBM (the input side of the step).  NULL format = BVC_PIXELS_F32. */
enum { BVC_PIXELS_F32 = 0, BVC_PIXELS_U8 = 1 };
typedef struct bvc_pixel_format {
    int dtype;
    float mean[4], std[4]; /* per channel; used for BVC_PIXELS_U8 only */
} bvc_pixel_format;
int bvc_videomae_forward_px(bvc_ctx* ctx, const void* pixels_dev, const bvc_pixel_format* fmt, const uint8_t* mask_dev, int batch,
                            const float* params_dev, float* loss_dev, float* logits_dev, void* stream);

/* Replaces autograd's backward of the step (scaler.scale(loss).backward(), pretrain_videomae.py:312).
 *   grad_loss_dev f32 scalar on the device: d(objective)/d(loss) (GradScaler's scale)
 *   grads_dev     f32 flat gradient buffer, OVERWRITTEN with d(objective)/d(param) */
int bvc_videomae_backward(bvc_ctx* ctx, const float* grad_loss_dev, float* grads_dev, bvc_bucket_fn on_bucket,
                          void* user, void* stream);
/* The context's bf16 copy of the flat parameters (what every product reads) and the hand-shake that lets the caller keep it
 * current instead of the forward: *shadow_bf16 / *numel (either may be NULL) describe it - element i mirrors element i of the flat
 * parameter buffer; valid = 1 vouches that it matches the parameters the NEXT forward will be given (that forward then skips its
 * cast pass over all parameters; the vouch is consumed by it), valid = 0 withdraws the vouch, valid < 0 only queries.
 * bvc_op_sgd_step / bvc_op_adam_step write the copy together with the parameters (their bf16_shadow argument).  Replaces nothing
 * in the reference: autocast re-casts every weight on every forward (pretrain_videomae.py:306-308). */
int bvc_videomae_shadow(bvc_ctx* ctx, int valid, void** shadow_bf16, int64_t* numel);

/* Copies a saved activation of the last forward as f32 into dst_dev (parity probes: "embed",
 * "enc<i>", "x_full", "dec<i>", "labels").  Returns the element count via *numel. */
int bvc_videomae_tap(bvc_ctx* ctx, const char* name, float* dst_dev, int64_t capacity, int64_t* numel, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder-only inference: the embedding extraction that runs between curriculum stages.
 * Replaces transformers.VideoMAEForVideoClassification(num_labels=0).forward as called at
 * benchmarks/compute_embeddings_videomae.py:78-96 (model assembly) and :253-264 (xmodel(pixel_values=inputs).logits):
 * every token (no mask) -> patch embedding + sinusoid -> encoder layers -> mean over tokens -> fc_norm LayerNorm.
 * `params` is a flat f32 buffer holding the "videomae.*" entries in the order of bvc_videomae_param_info (they are the
 * leading bvc_videomae_encoder_param_numel() elements of the pre-training layout, so a pre-training buffer can be passed
 * as is).  Forward only; the context owns workspaces for max_batch clips. */
typedef struct bvc_encoder_ctx bvc_encoder_ctx;
int64_t bvc_videomae_encoder_param_numel(const bvc_videomae_config* cfg);
int bvc_videomae_encoder_create(const bvc_videomae_config* cfg, int max_batch, bvc_encoder_ctx** out);
void bvc_videomae_encoder_destroy(bvc_encoder_ctx* ctx);
/*   pixels_dev   f32 [batch][T][C][H][W]
 *   fc_norm_w/b  f32 [hidden] or both NULL (then `pooled` is the plain token mean)
 *   tokens_dev   f32 [batch][L][hidden] last_hidden_state, or NULL
 *   pooled_dev   f32 [batch][hidden] = fc_norm(mean over tokens), or NULL                                     */
int bvc_videomae_encode(bvc_encoder_ctx* ctx, const float* pixels_dev, int batch, const float* params_dev,
                        const float* fc_norm_w, const float* fc_norm_b, float fc_norm_eps, float* tokens_dev,
                        float* pooled_dev, void* stream);
int bvc_videomae_encode_px(bvc_encoder_ctx* ctx, const void* pixels_dev, const bvc_pixel_format* fmt, int batch,
                           const float* params_dev, const float* fc_norm_w, const float* fc_norm_b, float fc_norm_eps,
                           float* tokens_dev, float* pooled_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * JEPA encoder and predictor (pretraining/predictive/vision_transformer.py).  Same conventions as above: flat f32
 * parameter / gradient buffers whose entries carry the reference's state-dict keys (pos_embed, patch_embed.proj.*,
 * blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*, norm.* / mask_token, predictor_pos_embed,
 * predictor_embed.*, predictor_blocks.N.*, predictor_norm.*, predictor_proj.*), borrowed device pointers, caller's stream.
 * Token index lists are int32 (the reference's int64 masks are narrowed by the host shim).                          */
typedef struct bvc_vit_config {
    int image_size, patch_size, num_channels, num_frames, tubelet_size;
    int embed_dim, depth, num_heads, mlp_hidden;
    float eps; /* 1e-6: vit_base() etc. build LayerNorm with eps=1e-6 (vision_transformer.py:546-568) */
} bvc_vit_config;
typedef struct bvc_vit_ctx bvc_vit_ctx;
int bvc_vit_param_count(const bvc_vit_config* cfg);
int64_t bvc_vit_param_numel(const bvc_vit_config* cfg);
int bvc_vit_param_info(const bvc_vit_config* cfg, int index, char* name, int name_cap, int64_t* offset, int64_t* numel,
                       int* ndim, int64_t shape[5]);
int bvc_vit_create(const bvc_vit_config* cfg, int max_batch, bvc_vit_ctx** out);
void bvc_vit_destroy(bvc_vit_ctx* ctx);
/* Replaces encoder(imgs, masks_enc) and target_encoder(imgs) (pretrain_jepa.py:386,395; VisionTransformer.forward,
 * vision_transformer.py:378-402).  imgs f32 [B][T][C][H][W]; idx int32 [B][ntok] = tokens kept per sample, NULL = all;
 * out f32 [B*ntok][embed_dim] (after the final LayerNorm). */
int bvc_vit_forward(bvc_vit_ctx* ctx, const float* imgs_dev, const int* idx_dev, int batch, int ntok, const float* params_dev,
                    float* out_dev, void* stream);
int bvc_vit_forward_px(bvc_vit_ctx* ctx, const void* imgs_dev, const bvc_pixel_format* fmt, const int* idx_dev, int batch, int ntok,
                       const float* params_dev, float* out_dev, void* stream);
/* d(out) f32 [B*ntok][embed_dim] -> flat gradients (overwritten); buckets reported tail-first as for VideoMAE. */
int bvc_vit_backward(bvc_vit_ctx* ctx, const float* dout_dev, float* grads_dev, bvc_bucket_fn on_bucket, void* user, void* stream);
/* as bvc_videomae_shadow */
int bvc_vit_shadow(bvc_vit_ctx* ctx, int valid, void** shadow_bf16, int64_t* numel);

typedef struct bvc_predictor_config {
    int seq_len;    /* tokens of the full grid (num_patches of the encoder) */
    int embed_dim;  /* encoder width */
    int pred_dim;   /* predictor_embed_dim (384) */
    int depth, num_heads, mlp_hidden;
    float eps;
} bvc_predictor_config;
typedef struct bvc_pred_ctx bvc_pred_ctx;
int bvc_predictor_param_count(const bvc_predictor_config* cfg);
int64_t bvc_predictor_param_numel(const bvc_predictor_config* cfg);
int bvc_predictor_param_info(const bvc_predictor_config* cfg, int index, char* name, int name_cap, int64_t* offset,
                             int64_t* numel, int* ndim, int64_t shape[5]);
int bvc_predictor_create(const bvc_predictor_config* cfg, int max_batch, int max_sets, int max_tokens, bvc_pred_ctx** out);
void bvc_predictor_destroy(bvc_pred_ctx* ctx);
/* Replaces predictor(z, masks_enc, masks_pred) (pretrain_jepa.py:396; VisionTransformerPredictor.forward,
 * vision_transformer.py:494-535).  z f32 [B*Nc][embed_dim]; idx_ctx int32 [B][Nc]; idx_pred int32 [nsets][B][Np];
 * out f32 [nsets*B*Np][embed_dim], rows ordered mask-set-major then sample, as apply_masks/cat produce them. */
int bvc_predictor_forward(bvc_pred_ctx* ctx, const float* z_dev, const int* idx_ctx_dev, const int* idx_pred_dev, int B, int Nc,
                          int nsets, int Np, const float* params_dev, float* out_dev, void* stream);
int bvc_predictor_backward(bvc_pred_ctx* ctx, const float* dout_dev, float* grads_dev, float* dz_dev, void* stream);
/* as bvc_videomae_shadow */
int bvc_predictor_shadow(bvc_pred_ctx* ctx, int valid, void** shadow_bf16, int64_t* numel);
/* The same, reporting gradient ranges tail-first (bvc_bucket_fn, as bvc_videomae_backward / bvc_vit_backward do) so that the
 * data-parallel wrapper can start the predictor's all-reduce per block: DDP(predictor, static_graph=True), pretrain_jepa.py:303. */
int bvc_predictor_backward_cb(bvc_pred_ctx* ctx, const float* dout_dev, float* grads_dev, float* dz_dev, bvc_bucket_fn on_bucket,
                              void* user, void* stream);

/* forward_target's post-processing (pretrain_jepa.py:387-392): F.layer_norm without affine over the feature dim, then the
 * rows the prediction masks select: out[(i*B+b)*Np + j] = LN(h[b*L + idx_pred[i][b][j]]) */
int bvc_op_target_select(const float* h, const int* idx_pred, float* out, int nsets, int B, int Np, int L, int D, float eps, void* stream);
/* F.smooth_l1_loss(z, h), beta = 1, mean (pretrain_jepa.py:400); workspace = bvc_op_smooth_l1_workspace(n) floats */
int bvc_op_smooth_l1_workspace(int64_t n);
int bvc_op_smooth_l1_fwd(const float* z, const float* h, int64_t n, float* workspace, float* loss, void* stream);
int bvc_op_smooth_l1_bwd(const float* z, const float* h, const float* grad_loss, int64_t n, float* dz, void* stream);
/* target <- momentum * target + (1 - momentum) * online over a flat range (pretrain_jepa.py:431-432) */
int bvc_op_ema(float* target, const float* online, int64_t n, float momentum, void* stream);
/* mean over the tokens of each sample, f32 [batch][ntok][dim] -> [batch][dim], and its backward (dx = dmean / ntok for every
 * token): `xmodel(inputs).mean(1)` of benchmarks/compute_embeddings_jepa.py:242 and the pooling between a ViT trunk and the
 * SimCLR head (BASELINE config 5).  Fixed summation order. */
int bvc_op_token_mean(const float* x, int batch, int ntok, int dim, float* out, void* stream);
int bvc_op_token_mean_bwd(const float* dmean, int batch, int ntok, int dim, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Operator-level entry points (the kernels the step is built from; also used by the parity tests). */
enum { BVC_GEMM_NT = 0, BVC_GEMM_NN = 1, BVC_GEMM_TN = 2 };
enum {
    BVC_EPI_F32 = 0, BVC_EPI_BF16 = 1, BVC_EPI_GELU = 2, BVC_EPI_RESID = 3, BVC_EPI_POS = 4, BVC_EPI_E2D = 5,
    BVC_EPI_LOSS = 6, BVC_EPI_DGELU = 7, BVC_EPI_F32_BF16 = 8, BVC_EPI_RELU = 9, BVC_EPI_DRELU = 10,
    BVC_EPI_NCE = 11, BVC_EPI_NCE_BWD = 12, BVC_EPI_RESID_LN = 13, BVC_EPI_DLN = 14
};
/* BVC_EPI_GELU writes TWO bf16 outputs: C2 = gelu(v + bias) and C = gelu'(v + bias), the factor the backward product needs (the
 * forward epilogue has the erf and the exponential at hand; the backward epilogue BVC_EPI_DGELU, C = v * aux with aux = that C, is
 * then one multiply per element).  BVC_EPI_RELU / BVC_EPI_DRELU: C = max(v + bias, 0) / C = v where aux > 0. */
/* C[M,N] = epilogue(alpha * alpha_dev[0] * sum_k A(m,k) B(k,n)); bf16 operands, f32 accumulation.
 * Replaces the nn.Linear forward / backward GEMMs that ATen dispatches for HF:225-237,269-275,299-322. */
typedef struct bvc_gemm_desc {
    const void* A; const void* B;   /* bf16 */
    int M, N, K, lda, ldb;
    uint32_t a_bytes, b_bytes;      /* extents of the A / B allocations (rows past them read as 0) */
    float alpha;
    const float* alpha_dev;         /* optional device scalar multiplied into alpha */
    int epi, split_k;
    void* C; int ldc; void* C2;
    const float* bias; const float* resid; const void* aux; int ldaux;
    const int* rowtok; const float* pos; const float* labels; float* partial;
    int rin, rout;
    float* rowsum;                  /* TN only: rowsum[m] += alpha * sum_k A(m,k)  (bias gradient of the same dY) */
    /* BVC_EPI_RESID_LN / BVC_EPI_DLN: the LayerNorm next to a Linear whose output rows are 384 wide (VideoMAE decoder, JEPA
     * predictor), computed in the product's epilogue on 128 x 384 tiles that hold complete rows (N == ldc == 384, one problem,
     * tile config 12 / -1).  Replaces the separate nn.LayerNorm passes of HF:339-357 / vision_transformer.py:225-231.
     *   RESID_LN (NT):  v = alpha AB + bias + resid -> C (f32);  C2 (bf16) = (v - mean) rstd ln_gamma + ln_beta, biased variance,
     *                   rstd = 1 / sqrt(var + ln_eps);  ln_mean / ln_rstd [M] written.
     *   DLN (NN):       g = alpha AB = d/d(LayerNorm output);  xhat = (ln_x - ln_mean) ln_rstd;
     *                   C (f32, in/out) += ln_rstd (g gamma - mean(g gamma) - xhat mean(g gamma xhat));  C2 (bf16) = the new C;
     *                   ln_dgamma += sum_m g xhat, ln_dbeta += sum_m g  (through ln_part: scratch of 256 x 2 x 384 floats). */
    const float* ln_gamma; const float* ln_beta;
    float* ln_mean; float* ln_rstd;
    float ln_eps;
    const float* ln_x;
    float* ln_part; float* ln_dgamma; float* ln_dbeta;
} bvc_gemm_desc;
/* tile_cfg: -1 auto; 0 = 128x128, 1 = 128x64, 2 = 64x64 (one workgroup per tile); 6 / 7 = persistent 128x128 / 128x64,
 * 9 = persistent 128x128 with deferred stores; 10 / 11 = 256x256 / 256x128, one 512-thread workgroup per CU (gemm8.hip);
 * 12 = 128x384 (TN only); 13 = 10 for TN products with plain f32 outputs that are ACCUMULATED: C += A^T B by f32 atomics even when
 * split_k == 1 (C pre-zeroed or holding the values to add to, as for split_k > 1) - bvc_op_gemm_plan_dw returns it for groups whose
 * unsplit tiles fill half to 15/16 of the chip: their K ranges are then balanced over all CUs.
 * (3-5 and 8 are experiment kernels that exist only in a -DBVC_EXPERIMENTS build.)  stages: -1 auto, 2..4 = K-loop variant. */
int bvc_op_gemm(const bvc_gemm_desc* problems, int count, int layout, int tile_cfg, int stages, void* stream);
int bvc_op_gemm_num_tiles(const bvc_gemm_desc* problem, int tile_cfg);
/* Introspection of the selection above, nothing is launched: the kernel instantiation bvc_op_gemm would run for these problems,
 * named as rocprofv3 prints it (e.g. "bvc::gemm8_kernel<256, true, true, 2>"), so that per-product timings can be attributed to the
 * rows of a kernel-stats table (bench.py's `roofline.kernels`); and the (tile_cfg, split_k) plan the step uses for a group of
 * weight-gradient products (returns the tile config, writes split_k into the descriptors). */
int bvc_op_gemm_kernel(const bvc_gemm_desc* problems, int count, int layout, int tile_cfg, int stages, char* name, int name_cap);
/* 1 when the step runs the LayerNorms of a transformer stack of this shape (tokens rows, width, MLP width, heads) inside the epilogues of
 * the neighbouring products (BVC_EPI_RESID_LN / BVC_EPI_DLN) under the current options, 0 when they are separate passes */
int bvc_op_row_ln_selected(int tokens, int width, int mlp_width, int heads);
int bvc_op_gemm_plan_dw(bvc_gemm_desc* problems, int count);

/* softmax(QK^T/sqrt(d))V for head_dim d = 64 or 32; qkv bf16 [B*N][3*d*H]; replaces HF:181-206 / SDPA (HF:239-252) and
 * Attention.forward of pretraining/predictive/vision_transformer.py:198-210 (the ViT-B predictor has d = 32) */
int bvc_op_attention_fwd(const void* qkv, void* ctx_out, float* lse, int B, int N, int H, int head_dim, void* stream);
int bvc_op_attention_bwd(const void* qkv, const void* ctx_in, const void* dctx, const float* lse, float* delta_scratch,
                         void* dqkv, int B, int N, int H, int head_dim, void* stream);
/* one of the backward's two kernels alone (timing probes): part 1 = dQ (+ delta_scratch), part 2 = dK / dV (reads delta_scratch) */
int bvc_op_attention_bwd_part(const void* qkv, const void* ctx_in, const void* dctx, const float* lse, float* delta_scratch,
                              void* dqkv, int B, int N, int H, int head_dim, int part, void* stream);
/* nn.LayerNorm forward/backward (HF:336-337,484); rows may be strided by (rin, rout, roff), rin<=0 = dense */
int bvc_op_layernorm_fwd(const float* x, int rin, int rout, int roff, const float* gamma, const float* beta, void* y_bf16,
                         float* mean, float* rstd, int M, int D, float eps, void* stream);
/* workspace: bvc_op_layernorm_bwd_workspace(M, D) floats of scratch */
int bvc_op_layernorm_bwd(const void* dy_bf16, const float* x, int rin, int rout, int roff, const float* mean,
                         const float* rstd, const float* gamma, float* dres, int accumulate, void* dres_bf16,
                         float* dgamma, float* dbeta, float* workspace, int M, int D, void* stream);
int64_t bvc_op_layernorm_bwd_workspace(int M, int D);
int bvc_op_colsum_bf16(const void* X, int M, int N, int ld, float alpha, const float* alpha_dev, float* out, void* stream);
int bvc_op_cast_bf16(const float* in, void* out, int64_t n, void* stream);
/* SimCLR loss pieces (info_nce_loss, pretraining/contrastive/pretrain_simclr.py:114-128, with its masks :284-292):
 * row_normalize = the two norms of F.cosine_similarity; the (2B x 2B) similarity is a GEMM of the normalised rows with
 * BVC_EPI_NCE (loss partials, nothing materialised) / BVC_EPI_NCE_BWD (bf16 d loss / d (cos/T)); nce_finalize folds
 * the partials: loss = logsumexp over ALL negative entries - mean over the tridiagonal positives; stats = {lse, -1/npos}. */
int bvc_op_row_normalize(const float* f, void* fn_bf16, float* inv_norm, int n, int p, float eps, void* stream);
int bvc_op_row_normalize_bwd(const float* f, const float* inv_norm, const float* dfn, float* df, int n, int p, void* stream);
int bvc_op_nce_finalize(const float* partial, int ntiles, float inv_temperature, int64_t npos, float* loss, float* stats, void* stream);
/* One-pass torch.optim.SGD(momentum, nesterov) update over a flat f32 range, replacing the optimiser step at
 * pretrain_videomae.py:187-189,313.  grad_scale / found_inf are GradScaler's device scalars (may be NULL):
 * gradients are divided by *grad_scale, and nothing is touched when *found_inf != 0.
 * bf16_shadow (may be NULL): the bf16 copy of `params` a model context keeps for its products (bvc_videomae_shadow / bvc_vit_shadow /
 * bvc_predictor_shadow, same offset as `params` inside the flat buffer): written with the updated parameters, so that the next
 * forward does not have to re-cast all of them. */
int bvc_op_sgd_step(float* params, float* grads, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                    float weight_decay, int nesterov, int first_step, int maximize, const float* grad_scale,
                    const float* found_inf, int write_unscaled_grads, void* bf16_shadow, void* stream);
/* One-pass torch.optim.AdamW / Adam update over a flat f32 range (pretrain_videomae.py:190-193, pretrain_simclr.py:238-240).
 * state3 = device {step count, lr / (1 - beta1^step), sqrt(1 - beta2^step)}: bvc_op_adam_prepare advances it once per optimiser
 * step (not at all when *found_inf != 0), bvc_op_adam_step consumes it.  Hyper-parameters are doubles (python floats) and are
 * combined in double before the cast to f32, as torch does.  decoupled = 1: AdamW (p *= 1 - lr wd); 0: Adam (g += wd p). */
int bvc_op_adam_prepare(float* state3, double lr, double beta1, double beta2, const float* found_inf, void* stream);
int bvc_op_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                     double eps, double weight_decay, int decoupled, int maximize, const float* state3, const float* grad_scale,
                     const float* found_inf, int write_unscaled_grads, void* bf16_shadow, void* stream);
/* The same updates for an optimiser with PARAMETER GROUPS over one flat buffer, as ONE launch (the reference's JEPA optimiser,
 * pretraining/predictive/helper.py:123-147: four groups - encoder / predictor weights with weight decay, their biases and 1-D
 * tensors with weight_decay 0 - over a layout that interleaves weights and biases).  The flat range [0, n) is cut into `nseg`
 * segments: seg_start (device int64 [nseg + 1], ascending element offsets, seg_start[0] = 0, seg_start[nseg] = n), seg_group
 * (device int32 [nseg]: the owning group, or -1 for elements no group owns - frozen parameters: left untouched), blk_seg (device
 * int32 [ceil(n / 1024)]: the segment holding element 1024 b).  The groups' hyper-parameters are host values passed per call
 * (schedules change them every step; the table is static).  Same arithmetic as bvc_op_sgd_step / bvc_op_adam_step per element. */
#define BVC_OPT_MAX_GROUPS 8
typedef struct bvc_sgd_groups {
    int ngroups;
    float lr[BVC_OPT_MAX_GROUPS], momentum[BVC_OPT_MAX_GROUPS], dampening[BVC_OPT_MAX_GROUPS], weight_decay[BVC_OPT_MAX_GROUPS];
    int nesterov[BVC_OPT_MAX_GROUPS], first_step[BVC_OPT_MAX_GROUPS], maximize[BVC_OPT_MAX_GROUPS];
} bvc_sgd_groups;
typedef struct bvc_adam_groups {
    int ngroups;
    double lr[BVC_OPT_MAX_GROUPS], beta1[BVC_OPT_MAX_GROUPS], beta2[BVC_OPT_MAX_GROUPS], eps[BVC_OPT_MAX_GROUPS], weight_decay[BVC_OPT_MAX_GROUPS];
    int decoupled[BVC_OPT_MAX_GROUPS], maximize[BVC_OPT_MAX_GROUPS];
} bvc_adam_groups;
int bvc_op_sgd_step_segments(float* params, float* grads, float* momentum_buf, int64_t n, const int64_t* seg_start, const int32_t* seg_group,
                             const int32_t* blk_seg, int nseg, const bvc_sgd_groups* groups, const float* grad_scale, const float* found_inf,
                             int write_unscaled_grads, void* bf16_shadow, void* stream);
/* state = device f32 [3 ngroups] ({step count, lr / (1 - beta1^step), sqrt(1 - beta2^step)} per group, advanced inside the call
 * unless *found_inf != 0); hyper_scratch = device f64 [3 ngroups] the call uploads lr / beta1 / beta2 into */
int bvc_op_adam_step_segments(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const int64_t* seg_start,
                              const int32_t* seg_group, const int32_t* blk_seg, int nseg, const bvc_adam_groups* groups, float* state,
                              double* hyper_scratch, const float* grad_scale, const float* found_inf, int write_unscaled_grads,
                              void* bf16_shadow, void* stream);
/* GradScaler's inf check (scaler.step at pretrain_videomae.py:313 -> torch.amp.GradScaler._check_inf_per_device) as one read-only
 * pass over a flat f32 range: *found_inf (device f32) is set to 1 if any element is Inf or NaN; it is never cleared here. */
int bvc_op_nonfinite_check(const float* x, int64_t n, float* found_inf, void* stream);
/* boolean mask -> ascending visible / masked token lists (the order x[~mask] / x[mask] produce, HF:121,578-579) */
int bvc_op_mask_index(const uint8_t* mask, int B, int L, int nvis, int nmask, int* vis_idx, int* msk_idx, int* status, void* stream);
/* tube patches of the visible tokens in Conv3d weight order (HF:157-177) */
int bvc_op_gather_patches(const float* clip, const int* vis_idx, void* A_bf16, int B, int nvis, int T, int C, int H, int W,
                          int ts, int ps, void* stream);
/* per-patch-normalised pixel targets of the masked tokens (HF:588-661) */
int bvc_op_pixel_labels(const float* clip, const int* msk_idx, float* labels, int B, int nmask, int T, int C, int H, int W,
                        int ts, int ps, int norm_pix, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Communication: one RCCL rank per process (one process per GPU), the communication stream and its event fences owned by the
 * library.  Replaces, for this path, what the reference reaches through torch.distributed:
 *   dist.init_process_group("nccl", rank=rank, world_size=world)      pretraining/generative/pretrain_videomae.py:87-90
 *   DistributedDataParallel's gradient reducer                         pretrain_videomae.py:180-181,312 (bucketed all-reduce, mean)
 *   AllReduce / AllGather autograd nodes                               pretraining/generative/ddputils.py:53-68,
 *                                                                      pretraining/predictive/distributed.py:49-76
 * RCCL is bound at run time (the librccl.so already mapped into the process when there is one - torch ships its own - else the
 * loader's), so the library links without it and loads on a machine without a GPU.
 * Rendezvous: rank 0 calls bvc_comm_unique_id and hands the 128 bytes to the other ranks by any side channel (the Python shim
 * broadcasts them over the process group the entry script has already initialised); every rank then calls bvc_comm_init with
 * its HIP device current.  All calls on one communicator come from the host thread that drives the step. */
#define BVC_COMM_ID_BYTES 128
typedef struct bvc_comm bvc_comm;
int bvc_comm_unique_id(void* id_out_host /* BVC_COMM_ID_BYTES */);
int bvc_comm_init(int rank, int world, const void* id_host, bvc_comm** out);
int bvc_comm_destroy(bvc_comm* comm);
/* the communication stream (hipStream_t), e.g. to record timing events around a bucket; owned by the communicator */
void* bvc_comm_stream(const bvc_comm* comm);
int bvc_comm_rank(const bvc_comm* comm);
int bvc_comm_world(const bvc_comm* comm);
/* path and version of the RCCL library in use (thread-local string; loads the library on first use) */
const char* bvc_comm_library(void);
/* One gradient bucket: in-place sum (average != 0: mean) over ranks of count f32 at buf, enqueued on the communication stream
 * behind an event recorded on producer_stream (the stream whose kernels wrote buf).  Returns immediately: this is the call a
 * bvc_bucket_fn makes while the rest of backward is still being enqueued. */
int bvc_allreduce_bucket(bvc_comm* comm, float* buf_dev, int64_t count, int average, void* producer_stream);
/* `stream` waits (event, no host block) for every bucket enqueued so far: end of backward, before the optimiser reads the gradients */
int bvc_comm_wait(bvc_comm* comm, void* stream);
/* Collectives whose result the caller's next kernels consume.  They too run on the communication stream (one communicator, one
 * stream, program order), fenced both ways: after everything enqueued on `stream` so far, and `stream` continues after them.
 *   allgather  recv[r * bytes_per_rank ...] = rank r's send buffer           (SimCLR global-batch negatives, forward)
 *   allreduce  in-place sum / mean of count f32                              (its backward; the loss scalar of AllReduce, ddputils.py:53-68)
 *   broadcast  root's buffer to every rank                                   (module-state sync at wrap time) */
int bvc_allgather(bvc_comm* comm, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream);
int bvc_allreduce(bvc_comm* comm, float* buf_dev, int64_t count, int average, void* stream);
int bvc_broadcast(bvc_comm* comm, void* buf_dev, int64_t bytes, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BVC_H */
