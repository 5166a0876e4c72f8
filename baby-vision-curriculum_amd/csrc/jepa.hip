// JEPA encoder / predictor on gfx950: contexts and forward / backward schedules (host code; kernels live in
// gemm.hip / attention.hip / rowops.hip, the transformer layer schedule in stack.hip).
//
// Follows pretraining/predictive/vision_transformer.py:
//   VisionTransformer.forward          :378-402  patch embed (Conv3d) + pos_embed, apply_masks, blocks, norm
//   VisionTransformerPredictor.forward :494-535  predictor_embed, + pos[masks_x], mask tokens + pos[masks], 4x repeat,
//                                                blocks (num_heads = encoder heads -> head_dim 32 for ViT-B), norm, slice, proj
// MI355X-first differences: only the tokens an encoder call keeps are patch-embedded (gather-GEMM; the reference convolves
// all tokens and gathers afterwards, :383-391); q,k,v are the reference's own fused qkv Linear; index lists are int32.
#include <math.h>
#include <stdio.h>

#include <algorithm>
#include <string.h>

#include "stack.h"

using namespace bvc;

namespace {

// ------------------------------------------------------------------ encoder
struct VitLayout : ParamTable {
    int64_t pe_w = 0, pe_b = 0, pos = 0, norm_w = 0, norm_b = 0;
    std::vector<LayerOff> blocks;
};

int check_vit(const bvc_vit_config& c) {
    BVC_REQUIRE(c.image_size > 0 && c.patch_size > 0 && c.image_size % c.patch_size == 0, "vit config: image_size %% patch_size != 0");
    BVC_REQUIRE(c.num_frames > 0 && c.tubelet_size > 0 && c.num_frames % c.tubelet_size == 0, "vit config: num_frames %% tubelet_size != 0");
    BVC_REQUIRE(c.patch_size % 8 == 0, "vit config: patch_size must be a multiple of 8");
    BVC_REQUIRE(c.num_heads > 0 && c.embed_dim % c.num_heads == 0, "vit config: embed_dim %% num_heads != 0");
    const int hd = c.embed_dim / c.num_heads;
    BVC_REQUIRE(hd == 64 || hd == 32, "vit config: head_dim %d unsupported (32 or 64)", hd);
    BVC_REQUIRE(c.embed_dim % 64 == 0 && c.embed_dim <= 1024, "vit config: embed_dim must be a multiple of 64, at most 1024");
    BVC_REQUIRE(c.mlp_hidden % 64 == 0 && c.depth >= 1, "vit config: mlp_hidden must be a multiple of 64");
    BVC_REQUIRE((c.num_channels * c.tubelet_size * c.patch_size * c.patch_size) % 64 == 0, "vit config: patch dim must be a multiple of 64");
    return BVC_OK;
}

int vit_seq(const bvc_vit_config& c) {
    const int g = c.image_size / c.patch_size;
    return (c.num_frames / c.tubelet_size) * g * g;
}

VitLayout make_vit_layout(const bvc_vit_config& c) {
    VitLayout L;
    const int64_t D = c.embed_dim;
    L.pos = L.add("pos_embed", {1, vit_seq(c), D});
    L.pe_w = L.add("patch_embed.proj.weight", {D, c.num_channels, c.tubelet_size, c.patch_size, c.patch_size});
    L.pe_b = L.add("patch_embed.proj.bias", {D});
    for (int i = 0; i < c.depth; ++i) L.blocks.push_back(add_layer_params(L, "blocks." + std::to_string(i) + ".", D, c.mlp_hidden, false));
    L.norm_w = L.add("norm.weight", {D});
    L.norm_b = L.add("norm.bias", {D});
    return L;
}

// ------------------------------------------------------------------ predictor
struct PredLayout : ParamTable {
    int64_t emb_w = 0, emb_b = 0, mask_token = 0, pos = 0, norm_w = 0, norm_b = 0, proj_w = 0, proj_b = 0;
    std::vector<LayerOff> blocks;
};

int check_pred(const bvc_predictor_config& c) {
    BVC_REQUIRE(c.seq_len > 0 && c.depth >= 1 && c.num_heads > 0, "predictor config: bad sizes");
    BVC_REQUIRE(c.pred_dim % c.num_heads == 0, "predictor config: pred_dim %% num_heads != 0");
    const int hd = c.pred_dim / c.num_heads;
    BVC_REQUIRE(hd % 8 == 0 && hd <= 64, "predictor config: head_dim %d unsupported (multiples of 8 up to 64; 24 runs zero-padded to 32)", hd);
    BVC_REQUIRE(c.embed_dim % 64 == 0 && c.pred_dim % 64 == 0 && c.mlp_hidden % 64 == 0, "predictor config: widths must be multiples of 64");
    BVC_REQUIRE(c.embed_dim <= 1024 && c.pred_dim <= 1024, "predictor config: widths above 1024 unsupported");
    return BVC_OK;
}

PredLayout make_pred_layout(const bvc_predictor_config& c) {
    PredLayout L;
    const int64_t D = c.embed_dim, Dp = c.pred_dim;
    L.mask_token = L.add("mask_token", {1, 1, Dp});
    L.pos = L.add("predictor_pos_embed", {1, c.seq_len, Dp});
    L.emb_w = L.add("predictor_embed.weight", {Dp, D});
    L.emb_b = L.add("predictor_embed.bias", {Dp});
    for (int i = 0; i < c.depth; ++i)
        L.blocks.push_back(add_layer_params(L, "predictor_blocks." + std::to_string(i) + ".", Dp, c.mlp_hidden, false));
    L.norm_w = L.add("predictor_norm.weight", {Dp});
    L.norm_b = L.add("predictor_norm.bias", {Dp});
    L.proj_w = L.add("predictor_proj.weight", {D, Dp});
    L.proj_b = L.add("predictor_proj.bias", {D});
    return L;
}

int param_info(const ParamTable& L, int index, char* name, int name_cap, int64_t* offset, int64_t* numel, int* ndim, int64_t shape[5]) {
    BVC_REQUIRE(index >= 0 && index < (int)L.entries.size(), "param_info: index %d out of range", index);
    const ParamEntry& e = L.entries[index];
    snprintf(name, name_cap, "%s", e.name.c_str());
    *offset = e.offset; *numel = e.numel; *ndim = e.ndim;
    for (int i = 0; i < 5; ++i) shape[i] = e.shape[i];
    return BVC_OK;
}

}  // namespace

struct bvc_vit_ctx {
    bvc_vit_config cfg;
    VitLayout lay;
    Arena arena;
    Work w;
    Stack st;
    int max_batch, L, Kp;
    int batch = 0, ntok = 0;
    bool have_forward = false;
    bf16_t* wbf;
    bool shadow_valid = false;   // set by bvc_vit_shadow for ONE forward: wbf already matches the parameters it will be given
    int* idx_all;      // identity token list [max_batch * L]
    const int* idx;    // token list of the current call
    bf16_t* Ape;       // bf16 [B*N][Kp]
    float *meanf, *rstdf;
    bf16_t* dout_bf;   // bf16 [B*N][D]
    float* dres;
};

struct bvc_pred_ctx {
    bvc_predictor_config cfg;
    PredLayout lay;
    Arena arena;
    Work w;
    Stack st;
    int max_batch, max_sets, max_tokens;
    int B = 0, Nc = 0, Np = 0, nsets = 0;
    bool have_forward = false;
    bf16_t* wbf;
    bool shadow_valid = false;   // see bvc_predictor_shadow
    bf16_t* z_bf;      // bf16 [B*Nc][D]
    float* xe;         // f32 [B*Nc][Dp] embedded context tokens (+ pos)
    const int *idx_ctx, *idx_pred;
    float *meanf, *rstdf;
    bf16_t* lnf;       // bf16 [nsets*B*Np][Dp]
    bf16_t* dout_bf;   // bf16 [nsets*B*Np][D]
    float* dres;       // f32 [nsets*B*(Nc+Np)][Dp]
    bf16_t* dxe;       // bf16 [B*Nc][Dp]
};

extern "C" {

// ============================================================================ encoder API
int bvc_vit_param_count(const bvc_vit_config* cfg) {
    if (!cfg || check_vit(*cfg) != BVC_OK) return BVC_ERR_INVALID;
    return (int)make_vit_layout(*cfg).entries.size();
}
int64_t bvc_vit_param_numel(const bvc_vit_config* cfg) {
    if (!cfg || check_vit(*cfg) != BVC_OK) return BVC_ERR_INVALID;
    return make_vit_layout(*cfg).total;
}
int bvc_vit_param_info(const bvc_vit_config* cfg, int index, char* name, int name_cap, int64_t* offset, int64_t* numel, int* ndim,
                       int64_t shape[5]) {
    BVC_REQUIRE(cfg && name && offset && numel && ndim && shape, "vit_param_info: null argument");
    TRY(check_vit(*cfg));
    return param_info(make_vit_layout(*cfg), index, name, name_cap, offset, numel, ndim, shape);
}

void bvc_vit_destroy(bvc_vit_ctx* c) {
    if (!c) return;
    free_work(c->w);
    c->arena.release();
    delete c;
}

int bvc_vit_create(const bvc_vit_config* cfg, int max_batch, bvc_vit_ctx** out) {
    BVC_REQUIRE(cfg && out && max_batch >= 1, "vit_create: bad argument");
    TRY(check_vit(*cfg));
    bvc_vit_ctx* c = new bvc_vit_ctx();
    c->cfg = *cfg;
    c->lay = make_vit_layout(*cfg);
    c->max_batch = max_batch;
    c->L = vit_seq(*cfg);
    c->Kp = cfg->num_channels * cfg->tubelet_size * cfg->patch_size * cfg->patch_size;
    const size_t M = (size_t)max_batch * c->L;
    const int D = cfg->embed_dim, I = cfg->mlp_hidden, H = cfg->num_heads;
    int rc = BVC_OK;
    auto fail = [&](int r) { bvc_vit_destroy(c); return r; };
#define A(expr) if ((rc = (expr)) != BVC_OK) return fail(rc)
    A(c->arena.alloc(&c->wbf, (size_t)c->lay.total));
    A(c->arena.alloc(&c->idx_all, M));
    A(c->arena.alloc(&c->Ape, M * c->Kp));
    A(alloc_stack(c->arena, c->st, D, I, H, cfg->depth, cfg->eps, M, (size_t)max_batch * H * c->L));
    A(c->arena.alloc(&c->meanf, M));
    A(c->arena.alloc(&c->rstdf, M));
    A(c->arena.alloc(&c->dout_bf, M * D));
    A(c->arena.alloc(&c->dres, M * D));
    A(alloc_work(c->arena, c->w, M * D, M * I, (size_t)max_batch * H * c->L, ln_bwd_workspace_floats_upto((int)M, D)));
    A(launch_iota_mod(c->idx_all, (int)M, c->L, nullptr));
    if (hipStreamSynchronize(nullptr) != hipSuccess) { set_error("vit_create: init sync failed"); return fail(BVC_ERR_HIP); }
#undef A
    *out = c;
    return BVC_OK;
}

// Replaces encoder(imgs, masks_enc) / target_encoder(imgs) (pretrain_jepa.py:386,395; VisionTransformer.forward :378-402).
//   imgs f32 [B][T][C][H][W];  idx int32 [B][ntok] token indices kept per sample (NULL = all tokens, ntok = seq_len);
//   out  f32 [B*ntok][D]
int bvc_vit_forward(bvc_vit_ctx* c, const float* imgs, const int* idx, int batch, int ntok, const float* params, float* out,
                    void* stream) {
    return bvc_vit_forward_px(c, imgs, nullptr, idx, batch, ntok, params, out, stream);
}

int bvc_vit_forward_px(bvc_vit_ctx* c, const void* imgs_any, const bvc_pixel_format* fmt, const int* idx, int batch, int ntok,
                       const float* params, float* out, void* stream) {
    BVC_REQUIRE(c && imgs_any && params && out, "vit_forward: null argument");
    PixelSrc imgs = pixels_f32((const float*)imgs_any);
    if (fmt && fmt->dtype != BVC_PIXELS_F32) {
        BVC_REQUIRE(fmt->dtype == BVC_PIXELS_U8 && c->cfg.num_channels <= 4, "vit_forward: unsupported pixel format");
        imgs.is_u8 = 1;
        for (int k = 0; k < 4; ++k) {
            BVC_REQUIRE(fmt->std[k] != 0.f || k >= c->cfg.num_channels, "vit_forward: std[%d] is zero", k);
            imgs.mean[k] = fmt->mean[k];
            imgs.stdv[k] = k < c->cfg.num_channels ? fmt->std[k] : 1.f;
        }
    }
    BVC_REQUIRE(batch >= 1 && batch <= c->max_batch, "vit_forward: batch %d outside [1, %d]", batch, c->max_batch);
    if (!idx) ntok = c->L;
    BVC_REQUIRE(ntok >= 1 && ntok <= c->L, "vit_forward: ntok %d outside [1, %d]", ntok, c->L);
    hipStream_t st = (hipStream_t)stream;
    const bvc_vit_config& cf = c->cfg;
    const VitLayout& L = c->lay;
    const int B = batch, N = ntok, M = B * N, D = cf.embed_dim;
    c->have_forward = false;
    c->batch = B; c->ntok = N;
    c->idx = idx ? idx : c->idx_all;
    c->w.params = params; c->w.wbf = c->wbf;
    const PatchGeom pg{cf.num_frames, cf.num_channels, cf.image_size, cf.image_size, cf.tubelet_size, cf.patch_size};
    if (!c->shadow_valid) TRY(launch_cast_bf16(params, c->wbf, (size_t)L.total, st));     // see bvc_vit_shadow
    c->shadow_valid = false;
    TRY(launch_gather_patches(imgs, c->idx, c->Ape, B, N, pg, st));
    {
        GemmProblem p = gemm(c->Ape, (size_t)M * c->Kp, c->Kp, c->wbf + L.pe_w, (size_t)D * c->Kp, c->Kp, M, D, c->Kp, EPI_POS, c->st.act[0].x_in, D);
        p.bias = params + L.pe_b; p.rowtok = c->idx; p.pos = params + L.pos;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    for (int i = 0; i < c->st.nlayers; ++i) {
        float* xo = i + 1 < c->st.nlayers ? c->st.act[i + 1].x_in : c->st.x_out;
        TRY(layer_forward(c->w, c->st, i, L.blocks[i], c->st.act[i].x_in, xo, B, N, st, i + 1 < c->st.nlayers ? &L.blocks[i + 1] : nullptr));
    }
    TRY(launch_ln_fwd(c->st.x_out, identity_rows(), params + L.norm_w, params + L.norm_b, nullptr, c->meanf, c->rstdf, M, D, cf.eps, st, out));
    c->have_forward = true;
    return BVC_OK;
}

// dout f32 [B*ntok][D] -> grads (flat f32, overwritten).  Pixels and pos_embed receive no gradient.
int bvc_vit_shadow(bvc_vit_ctx* c, int valid, void** shadow_bf16, int64_t* numel) {
    BVC_REQUIRE(c, "vit_shadow: null context");
    if (shadow_bf16) *shadow_bf16 = c->wbf;
    if (numel) *numel = (int64_t)c->lay.total;
    if (valid >= 0) c->shadow_valid = valid != 0;
    return BVC_OK;
}

int bvc_vit_backward(bvc_vit_ctx* c, const float* dout, float* G, bvc_bucket_fn on_bucket, void* user, void* stream) {
    BVC_REQUIRE(c && dout && G, "vit_backward: null argument");
    if (!c->have_forward) { set_error("vit_backward: no forward state"); return BVC_ERR_STATE; }
    c->have_forward = false;
    hipStream_t st = (hipStream_t)stream;
    const VitLayout& L = c->lay;
    const int B = c->batch, N = c->ntok, M = B * N, D = c->cfg.embed_dim;
    const float* params = c->w.params;
    begin_backward(c->w);
    BVC_CHECK_HIP(hipMemsetAsync(G, 0, (size_t)L.total * 4, st));
    TRY(launch_gather_rows_bf16(dout, identity_rows(), c->dout_bf, M, D, st));
    TRY(launch_ln_bwd(c->dout_bf, c->st.x_out, identity_rows(), c->meanf, c->rstdf, params + L.norm_w, c->dres, 0, c->w.dyb[0],
                      G + L.norm_w, G + L.norm_b, c->w.ln_part, M, D, st));
    if (on_bucket) on_bucket(L.norm_w, L.total - L.norm_w, user);
    for (int i = c->st.nlayers - 1; i >= 0; --i)
        TRY(layer_backward(c->w, c->st, i, L.blocks[i], c->st.act[i].x_in, c->dres, G, B, N, st, on_bucket, user));
    {
        GemmProblem p = gemm(c->w.dyb[c->w.seq % 3], (size_t)M * D, D, c->Ape, (size_t)M * c->Kp, c->Kp, D, c->Kp, M, EPI_F32, G + L.pe_w, c->Kp);
        p.rowsum = G + L.pe_b;
        const int tile = plan_dw(&p, 1);
        TRY(launch_gemm(&p, 1, GEMM_TN, tile, st));
    }
    TRY(join_side(c->w, c->w.seq & 1, st, on_bucket, user));
    TRY(join_side(c->w, (c->w.seq + 1) & 1, st, on_bucket, user));
    if (on_bucket) on_bucket(0, L.blocks.front().ln1w, user);
    return BVC_OK;
}

// ============================================================================ predictor API
int bvc_predictor_param_count(const bvc_predictor_config* cfg) {
    if (!cfg || check_pred(*cfg) != BVC_OK) return BVC_ERR_INVALID;
    return (int)make_pred_layout(*cfg).entries.size();
}
int64_t bvc_predictor_param_numel(const bvc_predictor_config* cfg) {
    if (!cfg || check_pred(*cfg) != BVC_OK) return BVC_ERR_INVALID;
    return make_pred_layout(*cfg).total;
}
int bvc_predictor_param_info(const bvc_predictor_config* cfg, int index, char* name, int name_cap, int64_t* offset, int64_t* numel,
                             int* ndim, int64_t shape[5]) {
    BVC_REQUIRE(cfg && name && offset && numel && ndim && shape, "predictor_param_info: null argument");
    TRY(check_pred(*cfg));
    return param_info(make_pred_layout(*cfg), index, name, name_cap, offset, numel, ndim, shape);
}

void bvc_predictor_destroy(bvc_pred_ctx* c) {
    if (!c) return;
    free_work(c->w);
    c->arena.release();
    delete c;
}

// max_batch samples, max_sets prediction masks per sample, sequences of at most max_tokens (context + predicted) tokens
int bvc_predictor_create(const bvc_predictor_config* cfg, int max_batch, int max_sets, int max_tokens, bvc_pred_ctx** out) {
    BVC_REQUIRE(cfg && out && max_batch >= 1 && max_sets >= 1 && max_tokens >= 2, "predictor_create: bad argument");
    TRY(check_pred(*cfg));
    bvc_pred_ctx* c = new bvc_pred_ctx();
    c->cfg = *cfg;
    c->lay = make_pred_layout(*cfg);
    c->max_batch = max_batch; c->max_sets = max_sets; c->max_tokens = max_tokens;
    const size_t S = (size_t)max_batch * max_sets, M = S * max_tokens, Mc = (size_t)max_batch * max_tokens;
    const int D = cfg->embed_dim, Dp = cfg->pred_dim, I = cfg->mlp_hidden, H = cfg->num_heads;
    int rc = BVC_OK;
    auto fail = [&](int r) { bvc_predictor_destroy(c); return r; };
#define A(expr) if ((rc = (expr)) != BVC_OK) return fail(rc)
    A(c->arena.alloc(&c->wbf, (size_t)c->lay.total));
    A(c->arena.alloc(&c->z_bf, Mc * D));
    A(c->arena.alloc(&c->xe, Mc * Dp));
    A(alloc_stack(c->arena, c->st, Dp, I, H, cfg->depth, cfg->eps, M, S * H * max_tokens));
    A(c->arena.alloc(&c->meanf, M));
    A(c->arena.alloc(&c->rstdf, M));
    A(c->arena.alloc(&c->lnf, M * Dp));
    A(c->arena.alloc(&c->dout_bf, M * D));
    A(c->arena.alloc(&c->dres, M * Dp));
    A(c->arena.alloc(&c->dxe, Mc * Dp));
    // dctx / dqkv scratch is as wide as the (possibly zero-padded) attention rows: Da = H * 32 when the heads are 24 wide
    A(alloc_work(c->arena, c->w, M * (size_t)std::max(Dp, c->st.Da), M * I, S * H * max_tokens, ln_bwd_workspace_floats_upto((int)M, Dp)));
#undef A
    *out = c;
    return BVC_OK;
}

// Replaces predictor(z, masks_enc, masks_pred) (pretrain_jepa.py:396; VisionTransformerPredictor.forward :494-535).
//   z f32 [B*Nc][D] context-encoder output; idx_ctx int32 [B][Nc]; idx_pred int32 [nsets][B][Np];
//   out f32 [nsets*B*Np][D], sequence order set-major then sample (row (i*B + b)*Np + j)
int bvc_predictor_forward(bvc_pred_ctx* c, const float* z, const int* idx_ctx, const int* idx_pred, int B, int Nc, int nsets, int Np,
                          const float* params, float* out, void* stream) {
    BVC_REQUIRE(c && z && idx_ctx && idx_pred && params && out, "predictor_forward: null argument");
    BVC_REQUIRE(B >= 1 && B <= c->max_batch && nsets >= 1 && nsets <= c->max_sets, "predictor_forward: batch / mask-set count out of range");
    BVC_REQUIRE(Nc >= 1 && Np >= 1 && Nc + Np <= c->max_tokens, "predictor_forward: %d + %d tokens exceed %d", Nc, Np, c->max_tokens);
    hipStream_t st = (hipStream_t)stream;
    const bvc_predictor_config& cf = c->cfg;
    const PredLayout& L = c->lay;
    const int D = cf.embed_dim, Dp = cf.pred_dim, T = Nc + Np, S = nsets * B, Mc = B * Nc, Mo = S * Np;
    c->have_forward = false;
    c->B = B; c->Nc = Nc; c->Np = Np; c->nsets = nsets;
    c->idx_ctx = idx_ctx; c->idx_pred = idx_pred;
    c->w.params = params; c->w.wbf = c->wbf;
    if (!c->shadow_valid) TRY(launch_cast_bf16(params, c->wbf, (size_t)L.total, st));     // see bvc_predictor_shadow
    c->shadow_valid = false;
    TRY(launch_gather_rows_bf16(z, identity_rows(), c->z_bf, Mc, D, st));
    {   // predictor_embed + bias + pos[masks_x]
        GemmProblem p = gemm(c->z_bf, (size_t)Mc * D, D, c->wbf + L.emb_w, (size_t)Dp * D, D, Mc, Dp, D, EPI_POS, c->xe, Dp);
        p.bias = params + L.emb_b; p.rowtok = idx_ctx; p.pos = params + L.pos;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    TRY(launch_pred_assemble(c->xe, params + L.mask_token, params + L.pos, idx_pred, c->st.act[0].x_in, nsets, B, Nc, Np, Dp, st));
    for (int i = 0; i < c->st.nlayers; ++i) {
        float* xo = i + 1 < c->st.nlayers ? c->st.act[i + 1].x_in : c->st.x_out;
        TRY(layer_forward(c->w, c->st, i, L.blocks[i], c->st.act[i].x_in, xo, S, T, st, i + 1 < c->st.nlayers ? &L.blocks[i + 1] : nullptr));
    }
    const RowMap tail{Np, T, Nc};
    TRY(launch_ln_fwd(c->st.x_out, tail, params + L.norm_w, params + L.norm_b, c->lnf, c->meanf, c->rstdf, Mo, Dp, cf.eps, st));
    {
        GemmProblem p = gemm(c->lnf, (size_t)Mo * Dp, Dp, c->wbf + L.proj_w, (size_t)D * Dp, Dp, Mo, D, Dp, EPI_F32, out, D);
        p.bias = params + L.proj_b;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    c->have_forward = true;
    return BVC_OK;
}

// dout f32 [nsets*B*Np][D] -> grads (flat f32, overwritten) and dz f32 [B*Nc][D] (gradient for the context encoder)
int bvc_predictor_shadow(bvc_pred_ctx* c, int valid, void** shadow_bf16, int64_t* numel) {
    BVC_REQUIRE(c, "predictor_shadow: null context");
    if (shadow_bf16) *shadow_bf16 = c->wbf;
    if (numel) *numel = (int64_t)c->lay.total;
    if (valid >= 0) c->shadow_valid = valid != 0;
    return BVC_OK;
}

int bvc_predictor_backward(bvc_pred_ctx* c, const float* dout, float* G, float* dz, void* stream) {
    return bvc_predictor_backward_cb(c, dout, G, dz, nullptr, nullptr, stream);
}

// The same with gradient ranges reported tail-first as their kernels are enqueued (bvc_bucket_fn): [norm .. end) after the final
// LayerNorm's backward, one range per block, [0 .. first block) at the end - what the data-parallel wrapper buckets on.
int bvc_predictor_backward_cb(bvc_pred_ctx* c, const float* dout, float* G, float* dz, bvc_bucket_fn on_bucket, void* user, void* stream) {
    BVC_REQUIRE(c && dout && G && dz, "predictor_backward: null argument");
    if (!c->have_forward) { set_error("predictor_backward: no forward state"); return BVC_ERR_STATE; }
    c->have_forward = false;
    hipStream_t st = (hipStream_t)stream;
    const bvc_predictor_config& cf = c->cfg;
    const PredLayout& L = c->lay;
    const int B = c->B, Nc = c->Nc, Np = c->Np, nsets = c->nsets;
    const int D = cf.embed_dim, Dp = cf.pred_dim, T = Nc + Np, S = nsets * B, Mc = B * Nc, Mo = S * Np, M = S * T;
    const float* params = c->w.params;
    const bf16_t* W = c->wbf;
    begin_backward(c->w);
    BVC_CHECK_HIP(hipMemsetAsync(G, 0, (size_t)L.total * 4, st));
    TRY(launch_gather_rows_bf16(dout, identity_rows(), c->dout_bf, Mo, D, st));
    {   // predictor_proj
        GemmProblem p = gemm(c->dout_bf, (size_t)Mo * D, D, c->lnf, (size_t)Mo * Dp, Dp, D, Dp, Mo, EPI_F32, G + L.proj_w, Dp);
        p.rowsum = G + L.proj_b;
        const int tile = plan_dw(&p, 1);
        TRY(launch_gemm(&p, 1, GEMM_TN, tile, st));
    }
    {
        GemmProblem p = gemm(c->dout_bf, (size_t)Mo * D, D, W + L.proj_w, (size_t)D * Dp, Dp, Mo, Dp, D, EPI_BF16, c->w.dln, Dp);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    BVC_CHECK_HIP(hipMemsetAsync(c->dres, 0, (size_t)M * Dp * 4, st));        // context rows get no gradient from the output slice
    BVC_CHECK_HIP(hipMemsetAsync(c->w.dyb[0], 0, (size_t)M * Dp * 2, st));
    const RowMap tail{Np, T, Nc};
    TRY(launch_ln_bwd(c->w.dln, c->st.x_out, tail, c->meanf, c->rstdf, params + L.norm_w, c->dres, 0, c->w.dyb[0],
                      G + L.norm_w, G + L.norm_b, c->w.ln_part, Mo, Dp, st));
    if (on_bucket) on_bucket(L.norm_w, L.total - L.norm_w, user);
    for (int i = c->st.nlayers - 1; i >= 0; --i)
        TRY(layer_backward(c->w, c->st, i, L.blocks[i], c->st.act[i].x_in, c->dres, G, S, T, st, on_bucket, user));
    TRY(join_side(c->w, c->w.seq & 1, st, on_bucket, user));
    TRY(join_side(c->w, (c->w.seq + 1) & 1, st, on_bucket, user));
    // sequence assembly: mask token (summed over every predicted position), context tokens (summed over the nsets copies)
    TRY(launch_colsum_f32(c->dres, tail, Mo, Dp, G + L.mask_token, st));
    TRY(launch_pred_ctx_grad(c->dres, c->dxe, nsets, B, Nc, Np, Dp, st));
    {   // predictor_embed
        GemmProblem p = gemm(c->dxe, (size_t)Mc * Dp, Dp, c->z_bf, (size_t)Mc * D, D, Dp, D, Mc, EPI_F32, G + L.emb_w, D);
        p.rowsum = G + L.emb_b;
        const int tile = plan_dw(&p, 1);
        TRY(launch_gemm(&p, 1, GEMM_TN, tile, st));
    }
    {
        GemmProblem p = gemm(c->dxe, (size_t)Mc * Dp, Dp, W + L.emb_w, (size_t)Dp * D, D, Mc, D, Dp, EPI_F32, dz, D);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    if (on_bucket) on_bucket(0, L.blocks.front().ln1w, user);      // mask token, (frozen) positions, predictor_embed
    return BVC_OK;
}

// ============================================================================ JEPA operator-level entry points
int bvc_op_target_select(const float* h, const int* idx_pred, float* out, int nsets, int B, int Np, int L, int D, float eps, void* stream) {
    BVC_REQUIRE(h && idx_pred && out, "op_target_select: null argument");
    return launch_target_select(h, idx_pred, out, nsets, B, Np, L, D, eps, (hipStream_t)stream);
}
int bvc_op_smooth_l1_workspace(int64_t n) { return smooth_l1_blocks((size_t)n); }
int bvc_op_smooth_l1_fwd(const float* z, const float* h, int64_t n, float* workspace, float* loss, void* stream) {
    BVC_REQUIRE(z && h && workspace && loss && n > 0, "op_smooth_l1_fwd: bad argument");
    return launch_smooth_l1_fwd(z, h, (size_t)n, workspace, loss, (hipStream_t)stream);
}
int bvc_op_smooth_l1_bwd(const float* z, const float* h, const float* grad_loss, int64_t n, float* dz, void* stream) {
    BVC_REQUIRE(z && h && grad_loss && dz && n > 0, "op_smooth_l1_bwd: bad argument");
    return launch_smooth_l1_bwd(z, h, grad_loss, (size_t)n, dz, (hipStream_t)stream);
}
int bvc_op_ema(float* target, const float* online, int64_t n, float momentum, void* stream) {
    BVC_REQUIRE(target && online && n >= 0, "op_ema: bad argument");
    return launch_ema(target, online, (size_t)n, momentum, (hipStream_t)stream);
}

int bvc_op_token_mean(const float* x, int batch, int ntok, int dim, float* out, void* stream) {
    BVC_REQUIRE(x && out && batch > 0 && ntok > 0 && dim > 0, "op_token_mean: bad argument");
    return launch_token_mean(x, batch, ntok, dim, out, (hipStream_t)stream);
}
int bvc_op_token_mean_bwd(const float* dmean, int batch, int ntok, int dim, float* dx, void* stream) {
    BVC_REQUIRE(dmean && dx && batch > 0 && ntok > 0 && dim > 0, "op_token_mean_bwd: bad argument");
    return launch_token_mean_bwd(dmean, batch, ntok, dim, dx, (hipStream_t)stream);
}

}  // extern "C"
