// VideoMAE pre-training step on gfx950: context, workspaces and the forward / backward schedules.
// Host code only - every kernel lives in gemm.hip / attention.hip / rowops.hip.
//
// Follows VideoMAEForPreTraining.forward (HF:531-671; instantiated by the reference at
// pretraining/generative/pretrain_videomae.py:61-64) with these MI355X-first changes:
//   * the tube patch embedding is a gather-GEMM over the VISIBLE tokens only (the reference convolves
//     all 1568 tokens and throws 90 % away, HF:119-122);
//   * sinusoid tables are built once and stay in HBM (the reference re-uploads them every step, HF:114-116,575-576);
//   * mask -> token lists are built on the device, no nonzero()/host sync;
//   * MSE and d(logits) are fused into the head GEMM's epilogue; the loss is reduced in a fixed order;
//   * parameters / gradients are one flat buffer each; weight gradients of a layer are one grouped GEMM.
// Numerics: bf16 MFMA operands, f32 accumulation, f32 residual stream, f32 LayerNorm / softmax / loss
// statistics, f32 master weights and gradients.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "stack.h"

namespace bvc {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }
}  // namespace bvc

using namespace bvc;

// ------------------------------------------------------------------ flat parameter layout
namespace {

struct Layout : ParamTable {
    int64_t pe_w = 0, pe_b = 0, e2d_w = 0, mask_token = 0, norm_w = 0, norm_b = 0, head_w = 0, head_b = 0;
    std::vector<LayerOff> enc, dec;

};

int check_config(const bvc_videomae_config& c) {
    BVC_REQUIRE(c.image_size > 0 && c.patch_size > 0 && c.image_size % c.patch_size == 0, "config: image_size %% patch_size != 0");
    BVC_REQUIRE(c.num_frames > 0 && c.tubelet_size > 0 && c.num_frames % c.tubelet_size == 0, "config: num_frames %% tubelet_size != 0");
    BVC_REQUIRE(c.patch_size % 8 == 0, "config: patch_size must be a multiple of 8");
    BVC_REQUIRE(c.hidden_size == 64 * c.num_attention_heads, "config: encoder head_dim must be 64 (hidden %d, heads %d)", c.hidden_size, c.num_attention_heads);
    BVC_REQUIRE(c.decoder_hidden_size == 64 * c.decoder_num_attention_heads, "config: decoder head_dim must be 64");
    BVC_REQUIRE(c.intermediate_size % 64 == 0 && c.decoder_intermediate_size % 64 == 0, "config: intermediate sizes must be multiples of 64");
    BVC_REQUIRE(c.hidden_size <= 1024 && c.decoder_hidden_size <= 1024, "config: hidden sizes above 1024 unsupported");
    BVC_REQUIRE((c.num_channels * c.tubelet_size * c.patch_size * c.patch_size) % 64 == 0, "config: patch dim must be a multiple of 64");
    BVC_REQUIRE(c.num_hidden_layers >= 1 && c.decoder_num_hidden_layers >= 1, "config: need at least one layer each");
    return BVC_OK;
}

Layout make_layout(const bvc_videomae_config& c) {
    Layout L;
    const int64_t D = c.hidden_size, Dd = c.decoder_hidden_size;
    const int64_t P = (int64_t)c.num_channels * c.tubelet_size * c.patch_size * c.patch_size;
    const std::string pe = "videomae.embeddings.patch_embeddings.projection.";
    L.pe_w = L.add(pe + "weight", {D, c.num_channels, c.tubelet_size, c.patch_size, c.patch_size});
    L.pe_b = L.add(pe + "bias", {D});
    for (int i = 0; i < c.num_hidden_layers; ++i)
        L.enc.push_back(add_layer_params(L, "videomae.encoder.layer." + std::to_string(i) + ".", D, c.intermediate_size, true));
    L.e2d_w = L.add("encoder_to_decoder.weight", {Dd, D});
    L.mask_token = L.add("mask_token", {1, 1, Dd});
    for (int i = 0; i < c.decoder_num_hidden_layers; ++i)
        L.dec.push_back(add_layer_params(L, "decoder.decoder_layers." + std::to_string(i) + ".", Dd, c.decoder_intermediate_size, true));
    L.norm_w = L.add("decoder.norm.weight", {Dd});
    L.norm_b = L.add("decoder.norm.bias", {Dd});
    L.head_w = L.add("decoder.head.weight", {P, Dd});
    L.head_b = L.add("decoder.head.bias", {P});
    return L;
}

}  // namespace

struct bvc_ctx {
    bvc_videomae_config cfg;
    Layout lay;
    int max_batch, nmask, nvis, L, P, Kp;
    Arena arena;
    Work w;
    // constants
    float *pos_enc, *pos_dec;
    // per-step state
    int batch = 0;
    bool have_forward = false;
    bf16_t* wbf;       // bf16 copy of the flat parameters
    bool shadow_valid = false;   // set by bvc_videomae_shadow for ONE forward: wbf already matches the parameters it will be given
    const float* params = nullptr;
    int *vis_idx, *msk_idx, *status;
    bf16_t* Ape;       // bf16 [B*nvis][Kp] gathered visible tubes
    Stack enc, dec;
    bf16_t* xe_bf;     // bf16 [B*nvis][D] encoder output
    float *meanf, *rstdf;
    bf16_t* lnf;       // bf16 [B*nmask][Dd]
    float* labels;     // f32 [B*nmask][P]
    bf16_t* diff;      // bf16 [B*nmask][P]  logits - labels
    float* partial;
    int npartial = 0;
    float *dres_enc, *dres_dec;
    bf16_t* de2d;
};

namespace {

void sinusoid(std::vector<float>& out, int n, int d) {   // HF:80-91, float64 then cast
    out.resize((size_t)n * d);
    for (int p = 0; p < n; ++p)
        for (int j = 0; j < d; ++j) {
            const double ang = (double)p / pow(10000.0, 2.0 * (j / 2) / (double)d);
            out[(size_t)p * d + j] = (float)((j & 1) ? cos(ang) : sin(ang));
        }
}

}  // namespace

namespace {
int pixel_src(const void* pixels, const bvc_pixel_format* fmt, int channels, PixelSrc* out) {
    PixelSrc px = pixels_f32((const float*)pixels);
    if (fmt && fmt->dtype != BVC_PIXELS_F32) {
        BVC_REQUIRE(fmt->dtype == BVC_PIXELS_U8, "pixel format: dtype %d unknown", fmt->dtype);
        BVC_REQUIRE(channels <= 4, "pixel format: uint8 input supports at most 4 channels");
        px.is_u8 = 1;
        for (int c = 0; c < 4; ++c) {
            BVC_REQUIRE(fmt->std[c] != 0.f || c >= channels, "pixel format: std[%d] is zero", c);
            px.mean[c] = fmt->mean[c];
            px.stdv[c] = c < channels ? fmt->std[c] : 1.f;
        }
    }
    *out = px;
    return BVC_OK;
}
}  // namespace

// ============================================================================ C ABI
extern "C" {

const char* bvc_last_error(void) { return bvc::last_error(); }
const char* bvc_version(void) { return "gfx950;bvc-hip-r5"; }

int bvc_set_option(const char* name, int value) {
    BVC_REQUIRE(name != nullptr, "set_option: null name");
    if (!strcmp(name, "gemm8")) { BVC_REQUIRE(value >= -1 && value <= 1, "set_option: gemm8 takes -1 / 0 / 1"); options().gemm8 = value; }
    else if (!strcmp(name, "dw_overlap")) options().dw_overlap = value != 0;
    else if (!strcmp(name, "row_stagger")) options().row_stagger = value != 0;
    else if (!strcmp(name, "row_ln")) { BVC_REQUIRE(value >= -1 && value <= 1, "set_option: row_ln takes -1 / 0 / 1"); options().row_ln = value; }
    else BVC_REQUIRE(false, "set_option: unknown option '%s'", name);
    return BVC_OK;
}
int bvc_get_option(const char* name) {
    if (name && !strcmp(name, "gemm8")) return options().gemm8;
    if (name && !strcmp(name, "dw_overlap")) return options().dw_overlap;
    if (name && !strcmp(name, "row_ln")) return options().row_ln;
    if (name && !strcmp(name, "row_stagger")) return options().row_stagger;
    bvc::set_error("get_option: unknown option '%s'", name ? name : "(null)");
    return BVC_ERR_INVALID;
}

int bvc_videomae_param_count(const bvc_videomae_config* cfg) {
    if (!cfg || check_config(*cfg) != BVC_OK) return BVC_ERR_INVALID;
    return (int)make_layout(*cfg).entries.size();
}

int64_t bvc_videomae_param_numel(const bvc_videomae_config* cfg) {
    if (!cfg || check_config(*cfg) != BVC_OK) return BVC_ERR_INVALID;
    return make_layout(*cfg).total;
}

int bvc_videomae_param_info(const bvc_videomae_config* cfg, int index, char* name, int name_cap, int64_t* offset,
                            int64_t* numel, int* ndim, int64_t shape[5]) {
    BVC_REQUIRE(cfg && name && offset && numel && ndim && shape, "param_info: null argument");
    TRY(check_config(*cfg));
    const Layout L = make_layout(*cfg);
    BVC_REQUIRE(index >= 0 && index < (int)L.entries.size(), "param_info: index %d out of range", index);
    const ParamEntry& e = L.entries[index];
    snprintf(name, name_cap, "%s", e.name.c_str());
    *offset = e.offset; *numel = e.numel; *ndim = e.ndim;
    for (int i = 0; i < 5; ++i) shape[i] = e.shape[i];
    return BVC_OK;
}

void bvc_videomae_destroy(bvc_ctx* c) {
    if (!c) return;
    free_work(c->w);
    c->arena.release();
    delete c;
}

int bvc_videomae_create(const bvc_videomae_config* cfg, int max_batch, int num_masked, bvc_ctx** out) {
    BVC_REQUIRE(cfg && out, "create: null argument");
    TRY(check_config(*cfg));
    BVC_REQUIRE(max_batch >= 1, "create: max_batch must be >= 1");
    bvc_ctx* c = new bvc_ctx();
    c->cfg = *cfg;
    c->lay = make_layout(*cfg);
    c->max_batch = max_batch;
    const int g = cfg->image_size / cfg->patch_size;
    c->L = (cfg->num_frames / cfg->tubelet_size) * g * g;
    c->nmask = num_masked;
    c->nvis = c->L - num_masked;
    c->P = cfg->num_channels * cfg->tubelet_size * cfg->patch_size * cfg->patch_size;
    c->Kp = c->P;
    if (!(num_masked >= 1 && c->nvis >= 1)) {
        delete c;
        set_error("create: num_masked=%d must leave at least one visible and one masked token of %d", num_masked, c->L);
        return BVC_ERR_INVALID;
    }
    const size_t B = max_batch, Mv = B * c->nvis, Md = B * c->L, Mm = B * c->nmask;
    const int D = cfg->hidden_size, Dd = cfg->decoder_hidden_size, I = cfg->intermediate_size, Id = cfg->decoder_intermediate_size;
    const int H = cfg->num_attention_heads, Hd = cfg->decoder_num_attention_heads;
    int rc = BVC_OK;
    auto fail = [&](int r) { bvc_videomae_destroy(c); return r; };
#define A(expr) if ((rc = (expr)) != BVC_OK) return fail(rc)
    A(c->arena.alloc(&c->pos_enc, (size_t)c->L * D));
    A(c->arena.alloc(&c->pos_dec, (size_t)c->L * Dd));
    A(c->arena.alloc(&c->wbf, (size_t)c->lay.total));
    A(c->arena.alloc(&c->vis_idx, Mv));
    A(c->arena.alloc(&c->msk_idx, Mm));
    // token 0 everywhere: a clip whose mask count differs from the context's leaves entries unwritten, and the gather kernels
    // must then read in-range indices (the status word flags the clip; the Python side raises on the next step)
    if (hipMemset(c->vis_idx, 0, (size_t)Mv * sizeof(int)) != hipSuccess || hipMemset(c->msk_idx, 0, (size_t)Mm * sizeof(int)) != hipSuccess) {
        set_error("videomae_create: hipMemset of the token lists failed");
        bvc_videomae_destroy(c);
        return BVC_ERR_HIP;
    }
    A(c->arena.alloc(&c->status, 4));
    A(c->arena.alloc(&c->Ape, Mv * c->Kp));
    A(alloc_stack(c->arena, c->enc, D, I, H, cfg->num_hidden_layers, cfg->layer_norm_eps, Mv, B * H * c->nvis));
    A(alloc_stack(c->arena, c->dec, Dd, Id, Hd, cfg->decoder_num_hidden_layers, cfg->layer_norm_eps, Md, B * Hd * c->L));
    A(c->arena.alloc(&c->xe_bf, Mv * D));
    A(c->arena.alloc(&c->meanf, Mm));
    A(c->arena.alloc(&c->rstdf, Mm));
    A(c->arena.alloc(&c->lnf, Mm * Dd));
    A(c->arena.alloc(&c->labels, Mm * c->P));
    A(c->arena.alloc(&c->diff, Mm * c->P));
    A(c->arena.alloc(&c->partial, (Mm / 64 + 2) * (c->P / 64 + 2)));
    A(c->arena.alloc(&c->dres_enc, Mv * D));
    A(c->arena.alloc(&c->dres_dec, Md * Dd));
    const size_t MD = std::max(Mv * D, Md * Dd), MI = std::max(Mv * I, Md * Id);
    A(c->arena.alloc(&c->de2d, Mv * Dd));
    A(alloc_work(c->arena, c->w, MD, MI, std::max(B * H * c->nvis, B * Hd * c->L),
                 std::max(ln_bwd_workspace_floats_upto((int)Mv, D), ln_bwd_workspace_floats_upto((int)Md, Dd))));
#undef A
    std::vector<float> tab;
    sinusoid(tab, c->L, D);
    if (hipMemcpy(c->pos_enc, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("create: pos upload failed"); return fail(BVC_ERR_HIP); }
    sinusoid(tab, c->L, Dd);
    if (hipMemcpy(c->pos_dec, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("create: pos upload failed"); return fail(BVC_ERR_HIP); }
    *out = c;
    return BVC_OK;
}

int bvc_videomae_forward(bvc_ctx* c, const float* pixels, const uint8_t* mask, int batch, const float* params,
                         float* loss, float* logits, void* stream) {
    return bvc_videomae_forward_px(c, pixels, nullptr, mask, batch, params, loss, logits, stream);
}

int bvc_videomae_forward_px(bvc_ctx* c, const void* pixels_any, const bvc_pixel_format* fmt, const uint8_t* mask, int batch,
                            const float* params, float* loss, float* logits, void* stream) {
    BVC_REQUIRE(c && pixels_any && mask && params && loss, "forward: null argument");
    PixelSrc pixels;
    TRY(pixel_src(pixels_any, fmt, c->cfg.num_channels, &pixels));
    BVC_REQUIRE(batch >= 1 && batch <= c->max_batch, "forward: batch %d outside [1, %d]", batch, c->max_batch);
    hipStream_t st = (hipStream_t)stream;
    const bvc_videomae_config& cf = c->cfg;
    const Layout& L = c->lay;
    const int B = batch, nvis = c->nvis, nmask = c->nmask, Lq = c->L;
    const int Mv = B * nvis, Mm = B * nmask;
    const int D = cf.hidden_size, Dd = cf.decoder_hidden_size, P = c->P;
    c->have_forward = false;
    c->batch = B;
    c->params = params;
    c->w.params = params;
    c->w.wbf = c->wbf;
    const PatchGeom pg{cf.num_frames, cf.num_channels, cf.image_size, cf.image_size, cf.tubelet_size, cf.patch_size};

    // the bf16 copy of the parameters: refreshed here unless the caller vouches that it still matches `params` (bvc_videomae_shadow:
    // the fused optimisers write it together with the parameters)
    if (!c->shadow_valid) TRY(launch_cast_bf16(params, c->wbf, (size_t)L.total, st));
    c->shadow_valid = false;
    BVC_CHECK_HIP(hipMemsetAsync(c->status, 0, 16, st));
    TRY(launch_mask_index(mask, B, Lq, nvis, nmask, c->vis_idx, c->msk_idx, c->status, st));
    TRY(launch_gather_patches(pixels, c->vis_idx, c->Ape, B, nvis, pg, st));
    {   // tube patch embedding of the visible tokens + bias + sinusoid (HF:109-124,164-177)
        GemmProblem p = gemm(c->Ape, (size_t)Mv * c->Kp, c->Kp, c->wbf + L.pe_w, (size_t)D * c->Kp, c->Kp, Mv, D, c->Kp, EPI_POS,
                             c->enc.act[0].x_in, D);
        p.bias = params + L.pe_b; p.rowtok = c->vis_idx; p.pos = c->pos_enc;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    for (int i = 0; i < c->enc.nlayers; ++i) {
        float* xo = i + 1 < c->enc.nlayers ? c->enc.act[i + 1].x_in : c->enc.x_out;
        TRY(layer_forward(c->w, c->enc, i, L.enc[i], c->enc.act[i].x_in, xo, B, nvis, st, i + 1 < c->enc.nlayers ? &L.enc[i + 1] : nullptr));
    }
    // encoder -> decoder glue (HF:566-582)
    TRY(launch_gather_rows_bf16(c->enc.x_out, identity_rows(), c->xe_bf, Mv, D, st));
    {
        GemmProblem p = gemm(c->xe_bf, (size_t)Mv * D, D, c->wbf + L.e2d_w, (size_t)Dd * D, D, Mv, Dd, D, EPI_E2D, c->dec.act[0].x_in, Dd);
        p.rowtok = c->vis_idx; p.pos = c->pos_dec; p.rin = nvis; p.rout = Lq;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    TRY(launch_fill_masked(c->dec.act[0].x_in, params + L.mask_token, c->pos_dec, c->msk_idx, B, Lq, nvis, nmask, Dd, st));
    for (int i = 0; i < c->dec.nlayers; ++i) {
        float* xo = i + 1 < c->dec.nlayers ? c->dec.act[i + 1].x_in : c->dec.x_out;
        TRY(layer_forward(c->w, c->dec, i, L.dec[i], c->dec.act[i].x_in, xo, B, Lq, st, i + 1 < c->dec.nlayers ? &L.dec[i + 1] : nullptr));
    }
    // last nmask tokens -> LayerNorm -> head, fused with the pixel-target MSE (HF:497-501,588-664)
    const RowMap tail{nmask, Lq, nvis};
    TRY(launch_ln_fwd(c->dec.x_out, tail, params + L.norm_w, params + L.norm_b, c->lnf, c->meanf, c->rstdf, Mm, Dd, cf.decoder_norm_eps, st));
    TRY(launch_labels(pixels, c->msk_idx, c->labels, B, nmask, pg, cf.norm_pix_loss, st));
    {
        GemmProblem p = gemm(c->lnf, (size_t)Mm * Dd, Dd, c->wbf + L.head_w, (size_t)P * Dd, Dd, Mm, P, Dd, EPI_LOSS, c->diff, P);
        p.bias = params + L.head_b; p.labels = c->labels; p.partial = c->partial; p.C2 = logits;
        c->npartial = gemm_num_tiles(p, -1);
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    TRY(launch_loss_finalize(c->partial, c->npartial, (double)Mm * (double)P, c->status, loss, st));
    c->have_forward = true;
    return BVC_OK;
}

int bvc_videomae_backward(bvc_ctx* c, const float* grad_loss, float* G, bvc_bucket_fn on_bucket, void* user, void* stream) {
    BVC_REQUIRE(c && grad_loss && G, "backward: null argument");
    if (!c->have_forward) { set_error("backward: no forward state (call bvc_videomae_forward first; one backward per forward)"); return BVC_ERR_STATE; }
    c->have_forward = false;
    hipStream_t st = (hipStream_t)stream;
    const bvc_videomae_config& cf = c->cfg;
    const Layout& L = c->lay;
    const int B = c->batch, nvis = c->nvis, nmask = c->nmask, Lq = c->L;
    const int Mv = B * nvis, Md = B * Lq, Mm = B * nmask;
    const int D = cf.hidden_size, Dd = cf.decoder_hidden_size, P = c->P;
    const bf16_t* W = c->wbf;
    const float* params = c->params;
    auto bucket = [&](int64_t lo, int64_t hi) { if (on_bucket) on_bucket(lo, hi - lo, user); };
    begin_backward(c->w);

    BVC_CHECK_HIP(hipMemsetAsync(G, 0, (size_t)L.total * 4, st));
    // d loss / d logits = (2 / (Mm P)) * diff * grad_loss  - folded into alpha of the three head products
    const float cmse = (float)(2.0 / ((double)Mm * (double)P));
    {
        GemmProblem p = gemm(c->diff, (size_t)Mm * P, P, c->lnf, (size_t)Mm * Dd, Dd, P, Dd, Mm, EPI_F32, G + L.head_w, Dd);
        p.alpha = cmse; p.alpha_dev = grad_loss;
        p.rowsum = G + L.head_b;
        const int tile = plan_dw(&p, 1);
        TRY(launch_gemm(&p, 1, GEMM_TN, tile, st));
    }
    {
        GemmProblem p = gemm(c->diff, (size_t)Mm * P, P, W + L.head_w, (size_t)P * Dd, Dd, Mm, Dd, P, EPI_BF16, c->w.dln, Dd);
        p.alpha = cmse; p.alpha_dev = grad_loss;
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    // visible rows of the decoder stream receive no gradient from the head
    BVC_CHECK_HIP(hipMemsetAsync(c->dres_dec, 0, (size_t)Md * Dd * 4, st));
    BVC_CHECK_HIP(hipMemsetAsync(c->w.dyb[0], 0, (size_t)Md * Dd * 2, st));
    const RowMap tail{nmask, Lq, nvis};
    TRY(launch_ln_bwd(c->w.dln, c->dec.x_out, tail, c->meanf, c->rstdf, params + L.norm_w, c->dres_dec, 0, c->w.dyb[0],
                      G + L.norm_w, G + L.norm_b, c->w.ln_part, Mm, Dd, st));
    bucket(L.norm_w, L.total);
    for (int i = c->dec.nlayers - 1; i >= 0; --i) {
        TRY(layer_backward(c->w, c->dec, i, L.dec[i], c->dec.act[i].x_in, c->dres_dec, G, B, Lq, st, on_bucket, user));
    }
    // decoder input: mask token, encoder_to_decoder
    TRY(launch_colsum_f32(c->dres_dec, tail, Mm, Dd, G + L.mask_token, st));
    const RowMap headrows{nvis, Lq, 0};
    TRY(launch_gather_rows_bf16(c->dres_dec, headrows, c->de2d, Mv, Dd, st));
    {
        GemmProblem p = gemm(c->de2d, (size_t)Mv * Dd, Dd, c->xe_bf, (size_t)Mv * D, D, Dd, D, Mv, EPI_F32, G + L.e2d_w, D);
        const int tile = plan_dw(&p, 1);
        TRY(launch_gemm(&p, 1, GEMM_TN, tile, st));
    }
    {
        GemmProblem p = gemm(c->de2d, (size_t)Mv * Dd, Dd, W + L.e2d_w, (size_t)Dd * D, D, Mv, D, Dd, EPI_F32_BF16, c->dres_enc, D);
        p.C2 = c->w.dyb[c->w.seq % 3];   // last read (as a dW operand) by backward step seq-3, fenced since
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    bucket(L.e2d_w, L.dec.front().ln1w);
    for (int i = c->enc.nlayers - 1; i >= 0; --i) {
        TRY(layer_backward(c->w, c->enc, i, L.enc[i], c->enc.act[i].x_in, c->dres_enc, G, B, nvis, st, on_bucket, user));
    }
    // patch embedding: weight and bias only (pixels need no gradient)
    {
        GemmProblem p = gemm(c->w.dyb[c->w.seq % 3], (size_t)Mv * D, D, c->Ape, (size_t)Mv * c->Kp, c->Kp, D, c->Kp, Mv, EPI_F32, G + L.pe_w, c->Kp);
        p.rowsum = G + L.pe_b;
        const int tile = plan_dw(&p, 1);
        TRY(launch_gemm(&p, 1, GEMM_TN, tile, st));
    }
    // fence the last side-stream launches (older one first so ranges keep arriving tail-first)
    TRY(join_side(c->w, c->w.seq & 1, st, on_bucket, user));
    TRY(join_side(c->w, (c->w.seq + 1) & 1, st, on_bucket, user));
    bucket(0, L.enc.front().ln1w);
    return BVC_OK;
}

int bvc_videomae_tap(bvc_ctx* c, const char* name, float* dst, int64_t capacity, int64_t* numel, void* stream) {
    BVC_REQUIRE(c && name && dst && numel, "tap: null argument");
    BVC_REQUIRE(c->batch > 0, "tap: no forward has run");
    const int B = c->batch;
    const size_t Mv = (size_t)B * c->nvis, Md = (size_t)B * c->L, Mm = (size_t)B * c->nmask;
    const int D = c->cfg.hidden_size, Dd = c->cfg.decoder_hidden_size;
    const float* src = nullptr;
    size_t n = 0;
    int idx = -1;
    if (!strcmp(name, "embed")) { src = c->enc.act[0].x_in; n = Mv * D; }
    else if (!strcmp(name, "x_full")) { src = c->dec.act[0].x_in; n = Md * Dd; }
    else if (!strcmp(name, "labels")) { src = c->labels; n = Mm * c->P; }
    else if (sscanf(name, "enc%d", &idx) == 1 && idx >= 0 && idx < c->enc.nlayers) {
        src = idx + 1 < c->enc.nlayers ? c->enc.act[idx + 1].x_in : c->enc.x_out; n = Mv * D;
    } else if (sscanf(name, "dec%d", &idx) == 1 && idx >= 0 && idx < c->dec.nlayers) {
        src = idx + 1 < c->dec.nlayers ? c->dec.act[idx + 1].x_in : c->dec.x_out; n = Md * Dd;
    }
    BVC_REQUIRE(src, "tap: unknown activation '%s'", name);
    *numel = (int64_t)n;
    BVC_REQUIRE((int64_t)n <= capacity, "tap: destination holds %lld elements, '%s' has %lld", (long long)capacity, name, (long long)n);
    BVC_CHECK_HIP(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BVC_OK;
}

// ------------------------------------------------------------------ encoder-only inference (embedding extraction)
// Replaces VideoMAEForVideoClassification(num_labels=0).forward as benchmarks/compute_embeddings_videomae.py:78-96,253-264
// uses it between curriculum stages: all tokens (no mask) -> encoder -> mean over tokens -> fc_norm (HF
// VideoMAEForVideoClassification.forward; VideoMAEModel.layernorm is None under use_mean_pooling).  Forward only: ONE set of
// layer activations is reused by every layer, so the working set is independent of depth.
struct bvc_encoder_ctx {
    bvc_videomae_config cfg;
    Layout lay;
    int max_batch, L, P;
    Arena arena;
    Work w;            // parameter views only; no backward scratch is allocated
    Stack st;          // one LayerAct
    float *pos_enc, *xa, *xb, *pooled, *mean, *rstd;
    bf16_t *wbf, *Ape;
    int* idx_all;
};

void bvc_videomae_encoder_destroy(bvc_encoder_ctx* c) {
    if (!c) return;
    c->arena.release();
    delete c;
}

int64_t bvc_videomae_encoder_param_numel(const bvc_videomae_config* cfg) {
    if (!cfg || check_config(*cfg) != BVC_OK) return BVC_ERR_INVALID;
    return make_layout(*cfg).e2d_w;    // the "videomae.*" entries are the leading part of the pre-training layout
}

int bvc_videomae_encoder_create(const bvc_videomae_config* cfg, int max_batch, bvc_encoder_ctx** out) {
    BVC_REQUIRE(cfg && out && max_batch >= 1, "encoder_create: bad argument");
    TRY(check_config(*cfg));
    bvc_encoder_ctx* c = new bvc_encoder_ctx();
    c->cfg = *cfg;
    c->lay = make_layout(*cfg);
    c->max_batch = max_batch;
    const int g = cfg->image_size / cfg->patch_size;
    c->L = (cfg->num_frames / cfg->tubelet_size) * g * g;
    c->P = cfg->num_channels * cfg->tubelet_size * cfg->patch_size * cfg->patch_size;
    const size_t M = (size_t)max_batch * c->L;
    const int D = cfg->hidden_size, I = cfg->intermediate_size, H = cfg->num_attention_heads;
    int rc = BVC_OK;
    auto fail = [&](int r) { bvc_videomae_encoder_destroy(c); return r; };
    if (M * (size_t)std::max(I, c->P) * 2 >= 0xffffffffull) { set_error("encoder_create: max_batch %d needs operands above 4 GiB", max_batch); return fail(BVC_ERR_INVALID); }
#define A(expr) if ((rc = (expr)) != BVC_OK) return fail(rc)
    A(c->arena.alloc(&c->pos_enc, (size_t)c->L * D));
    A(c->arena.alloc(&c->wbf, (size_t)c->lay.e2d_w));
    A(c->arena.alloc(&c->idx_all, M));
    A(c->arena.alloc(&c->Ape, M * c->P));
    A(alloc_stack(c->arena, c->st, D, I, H, 1, cfg->layer_norm_eps, M, (size_t)max_batch * H * c->L));
    A(c->arena.alloc(&c->xa, M * D));
    A(c->arena.alloc(&c->xb, M * D));
    A(c->arena.alloc(&c->pooled, (size_t)max_batch * D));
    A(c->arena.alloc(&c->mean, (size_t)max_batch));
    A(c->arena.alloc(&c->rstd, (size_t)max_batch));
    A(launch_iota_mod(c->idx_all, (int)M, c->L, nullptr));
#undef A
    std::vector<float> tab;
    sinusoid(tab, c->L, D);
    if (hipMemcpy(c->pos_enc, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("encoder_create: pos upload failed"); return fail(BVC_ERR_HIP); }
    *out = c;
    return BVC_OK;
}

int bvc_videomae_encode(bvc_encoder_ctx* c, const float* pixels, int batch, const float* params, const float* fc_norm_w,
                        const float* fc_norm_b, float fc_norm_eps, float* tokens, float* pooled, void* stream) {
    return bvc_videomae_encode_px(c, pixels, nullptr, batch, params, fc_norm_w, fc_norm_b, fc_norm_eps, tokens, pooled, stream);
}

int bvc_videomae_encode_px(bvc_encoder_ctx* c, const void* pixels_any, const bvc_pixel_format* fmt, int batch, const float* params,
                           const float* fc_norm_w, const float* fc_norm_b, float fc_norm_eps, float* tokens, float* pooled,
                           void* stream) {
    BVC_REQUIRE(c && pixels_any && params && (tokens || pooled), "encode: null argument");
    PixelSrc pixels;
    TRY(pixel_src(pixels_any, fmt, c->cfg.num_channels, &pixels));
    BVC_REQUIRE(batch >= 1 && batch <= c->max_batch, "encode: batch %d outside [1, %d]", batch, c->max_batch);
    BVC_REQUIRE((fc_norm_w == nullptr) == (fc_norm_b == nullptr), "encode: fc_norm weight and bias go together");
    hipStream_t st = (hipStream_t)stream;
    const bvc_videomae_config& cf = c->cfg;
    const Layout& L = c->lay;
    const int B = batch, N = c->L, M = B * N, D = cf.hidden_size, P = c->P;
    c->w.params = params;
    c->w.wbf = c->wbf;
    const PatchGeom pg{cf.num_frames, cf.num_channels, cf.image_size, cf.image_size, cf.tubelet_size, cf.patch_size};
    TRY(launch_cast_bf16(params, c->wbf, (size_t)L.e2d_w, st));
    TRY(launch_gather_patches(pixels, c->idx_all, c->Ape, B, N, pg, st));
    float* x = c->xa;
    float* y = c->xb;
    {
        GemmProblem p = gemm(c->Ape, (size_t)M * P, P, c->wbf + L.pe_w, (size_t)D * P, P, M, D, P, EPI_POS, x, D);
        p.bias = params + L.pe_b; p.rowtok = c->idx_all; p.pos = c->pos_enc;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    const int nl = (int)L.enc.size();
    for (int i = 0; i < nl; ++i) {
        float* dst = (i + 1 == nl && tokens) ? tokens : y;     // the last layer writes straight into the caller's buffer
        TRY(layer_forward(c->w, c->st, 0, L.enc[i], x, dst, B, N, st));
        if (dst == y) std::swap(x, y); else x = dst;
    }
    if (pooled) {
        float* mp = fc_norm_w ? c->pooled : pooled;
        TRY(launch_token_mean(x, B, N, D, mp, st));
        if (fc_norm_w)
            TRY(launch_ln_fwd(mp, identity_rows(), fc_norm_w, fc_norm_b, nullptr, c->mean, c->rstd, B, D, fc_norm_eps, st, pooled));
    }
    return BVC_OK;
}

// ------------------------------------------------------------------ operator-level entry points
int bvc_op_gemm(const bvc_gemm_desc* problems, int count, int layout, int tile_cfg, int stages, void* stream) {
    BVC_REQUIRE(problems, "op_gemm: null problems");
    BVC_REQUIRE(layout >= 0 && layout <= 2, "op_gemm: bad layout %d", layout);
    return launch_gemm(problems, count, (GemmLayout)layout, tile_cfg, (hipStream_t)stream, stages);
}
int bvc_op_gemm_kernel(const bvc_gemm_desc* problems, int count, int layout, int tile_cfg, int stages, char* name, int name_cap) {
    BVC_REQUIRE(problems && name && name_cap > 0, "op_gemm_kernel: null argument");
    BVC_REQUIRE(layout >= 0 && layout <= 2, "op_gemm_kernel: bad layout %d", layout);
    DryRun& d = dry_run();
    d.on = true;
    d.name[0] = 0;
    const int rc = launch_gemm(problems, count, (GemmLayout)layout, tile_cfg, nullptr, stages);
    d.on = false;
    if (rc == BVC_OK) snprintf(name, name_cap, "%s", d.name);
    return rc;
}
int bvc_op_gemm_plan_dw(bvc_gemm_desc* problems, int count) {
    BVC_REQUIRE(problems && count >= 1 && count <= 4, "op_gemm_plan_dw: bad argument");
    return plan_dw(problems, count);
}
int bvc_op_gemm_num_tiles(const bvc_gemm_desc* problem, int tile_cfg) {
    if (!problem) return BVC_ERR_INVALID;
    return gemm_num_tiles(*problem, tile_cfg);
}
int bvc_op_attention_fwd(const void* qkv, void* ctx_out, float* lse, int B, int N, int H, int head_dim, void* stream) {
    BVC_REQUIRE(qkv && ctx_out && lse, "op_attention_fwd: null argument");
    return launch_attn_fwd((const bf16_t*)qkv, (bf16_t*)ctx_out, lse, B, N, H, head_dim, (hipStream_t)stream);
}
int bvc_op_attention_bwd(const void* qkv, const void* ctx_in, const void* dctx, const float* lse, float* delta, void* dqkv,
                         int B, int N, int H, int head_dim, void* stream) {
    BVC_REQUIRE(qkv && ctx_in && dctx && lse && delta && dqkv, "op_attention_bwd: null argument");
    return launch_attn_bwd((const bf16_t*)qkv, (const bf16_t*)ctx_in, (const bf16_t*)dctx, lse, delta, (bf16_t*)dqkv, B, N, H, head_dim,
                           (hipStream_t)stream);
}
int bvc_op_attention_bwd_part(const void* qkv, const void* ctx_in, const void* dctx, const float* lse, float* delta, void* dqkv,
                              int B, int N, int H, int head_dim, int part, void* stream) {
    BVC_REQUIRE(qkv && ctx_in && dctx && lse && delta && dqkv, "op_attention_bwd_part: null argument");
    BVC_REQUIRE(part == 1 || part == 2, "op_attention_bwd_part: part is 1 (dQ + delta) or 2 (dK, dV)");
    return launch_attn_bwd((const bf16_t*)qkv, (const bf16_t*)ctx_in, (const bf16_t*)dctx, lse, delta, (bf16_t*)dqkv, B, N, H, head_dim,
                           (hipStream_t)stream, 0.f, part);
}
int bvc_op_layernorm_fwd(const float* x, int rin, int rout, int roff, const float* gamma, const float* beta, void* y,
                         float* mean, float* rstd, int M, int D, float eps, void* stream) {
    BVC_REQUIRE(x && gamma && beta && y && mean && rstd, "op_layernorm_fwd: null argument");
    return launch_ln_fwd(x, RowMap{rin, rout, roff}, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps, (hipStream_t)stream);
}
int bvc_op_layernorm_bwd(const void* dy, const float* x, int rin, int rout, int roff, const float* mean, const float* rstd,
                         const float* gamma, float* dres, int accumulate, void* dres_bf16, float* dgamma, float* dbeta,
                         float* workspace, int M, int D, void* stream) {
    BVC_REQUIRE(dy && x && mean && rstd && gamma && dres && dgamma && dbeta && workspace, "op_layernorm_bwd: null argument");
    return launch_ln_bwd((const bf16_t*)dy, x, RowMap{rin, rout, roff}, mean, rstd, gamma, dres, accumulate, (bf16_t*)dres_bf16,
                         dgamma, dbeta, workspace, M, D, (hipStream_t)stream);
}
int64_t bvc_op_layernorm_bwd_workspace(int M, int D) { return (int64_t)ln_bwd_workspace_floats(M, D); }
int bvc_op_colsum_bf16(const void* X, int M, int N, int ld, float alpha, const float* alpha_dev, float* out, void* stream) {
    BVC_REQUIRE(X && out, "op_colsum_bf16: null argument");
    return launch_colsum_bf16_scaled((const bf16_t*)X, M, N, ld, alpha, alpha_dev, out, (hipStream_t)stream);
}
int bvc_op_sgd_step(float* params, float* grads, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                    float weight_decay, int nesterov, int first_step, int maximize, const float* grad_scale,
                    const float* found_inf, int write_unscaled_grads, void* bf16_shadow, void* stream) {
    BVC_REQUIRE(params && grads && n >= 0, "op_sgd_step: bad argument");
    return launch_sgd_step(params, grads, momentum_buf, (size_t)n, lr, momentum, dampening, weight_decay, nesterov, first_step,
                           maximize, grad_scale, found_inf, write_unscaled_grads, (bf16_t*)bf16_shadow, (hipStream_t)stream);
}
int bvc_op_adam_prepare(float* state3, double lr, double beta1, double beta2, const float* found_inf, void* stream) {
    BVC_REQUIRE(state3, "op_adam_prepare: null state");
    return launch_adam_prep(state3, lr, beta1, beta2, found_inf, (hipStream_t)stream);
}
int bvc_op_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                     double eps, double weight_decay, int decoupled, int maximize, const float* state3, const float* grad_scale,
                     const float* found_inf, int write_unscaled_grads, void* bf16_shadow, void* stream) {
    BVC_REQUIRE(params && grads && exp_avg && exp_avg_sq && state3 && n >= 0, "op_adam_step: bad argument");
    return launch_adam_step(params, grads, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, decoupled, maximize,
                            state3, grad_scale, found_inf, write_unscaled_grads, (bf16_t*)bf16_shadow, (hipStream_t)stream);
}
int bvc_op_row_ln_selected(int tokens, int width, int mlp_width, int heads) {
    if (tokens <= 0 || width <= 0 || heads <= 0 || width % heads != 0) return 0;
    Stack s;
    s.D = width; s.I = mlp_width; s.H = heads; s.nlayers = 0; s.eps = 0.f; s.x_out = nullptr;
    s.hd = width / heads; s.hdp = s.hd <= 32 ? 32 : 64; s.Da = heads * s.hdp;
    return fuse_row_ln(s, tokens) ? 1 : 0;
}
int bvc_op_sgd_step_segments(float* params, float* grads, float* momentum_buf, int64_t n, const int64_t* seg_start, const int32_t* seg_group,
                             const int32_t* blk_seg, int nseg, const bvc_sgd_groups* groups, const float* grad_scale, const float* found_inf,
                             int write_unscaled_grads, void* bf16_shadow, void* stream) {
    BVC_REQUIRE(params && grads && n >= 0, "op_sgd_step_segments: bad argument");
    return launch_sgd_step_segments(params, grads, momentum_buf, n, seg_start, seg_group, blk_seg, nseg, groups, grad_scale, found_inf,
                                    write_unscaled_grads, (bf16_t*)bf16_shadow, (hipStream_t)stream);
}
int bvc_op_adam_step_segments(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const int64_t* seg_start,
                              const int32_t* seg_group, const int32_t* blk_seg, int nseg, const bvc_adam_groups* groups, float* state,
                              double* hyper_scratch, const float* grad_scale, const float* found_inf, int write_unscaled_grads,
                              void* bf16_shadow, void* stream) {
    BVC_REQUIRE(params && grads && exp_avg && exp_avg_sq && n >= 0, "op_adam_step_segments: bad argument");
    return launch_adam_step_segments(params, grads, exp_avg, exp_avg_sq, n, seg_start, seg_group, blk_seg, nseg, groups, state, hyper_scratch,
                                     grad_scale, found_inf, write_unscaled_grads, (bf16_t*)bf16_shadow, (hipStream_t)stream);
}
int bvc_videomae_shadow(bvc_ctx* c, int valid, void** shadow_bf16, int64_t* numel) {
    BVC_REQUIRE(c, "videomae_shadow: null context");
    if (shadow_bf16) *shadow_bf16 = c->wbf;
    if (numel) *numel = (int64_t)c->lay.total;
    if (valid >= 0) c->shadow_valid = valid != 0;
    return BVC_OK;
}
int bvc_op_nonfinite_check(const float* x, int64_t n, float* found_inf, void* stream) {
    BVC_REQUIRE(x && found_inf && n >= 0, "op_nonfinite_check: bad argument");
    return launch_nonfinite_check(x, (size_t)n, found_inf, (hipStream_t)stream);
}
int bvc_op_row_normalize(const float* f, void* fn_bf16, float* inv_norm, int n, int p, float eps, void* stream) {
    BVC_REQUIRE(f && fn_bf16 && inv_norm, "op_row_normalize: null argument");
    return launch_row_normalize(f, (bf16_t*)fn_bf16, inv_norm, n, p, eps, (hipStream_t)stream);
}
int bvc_op_row_normalize_bwd(const float* f, const float* inv_norm, const float* dfn, float* df, int n, int p, void* stream) {
    BVC_REQUIRE(f && inv_norm && dfn && df, "op_row_normalize_bwd: null argument");
    return launch_row_normalize_bwd(f, inv_norm, dfn, df, n, p, (hipStream_t)stream);
}
int bvc_op_nce_finalize(const float* partial, int ntiles, float inv_temperature, int64_t npos, float* loss, float* stats, void* stream) {
    BVC_REQUIRE(partial && loss && stats && ntiles > 0 && npos > 0, "op_nce_finalize: bad argument");
    return launch_nce_finalize(partial, ntiles, inv_temperature, (double)npos, loss, stats, (hipStream_t)stream);
}
int bvc_op_cast_bf16(const float* in, void* out, int64_t n, void* stream) {
    BVC_REQUIRE(in && out && n >= 0, "op_cast_bf16: bad argument");
    return launch_cast_bf16(in, (bf16_t*)out, (size_t)n, (hipStream_t)stream);
}
int bvc_op_mask_index(const uint8_t* mask, int B, int L, int nvis, int nmask, int* vis_idx, int* msk_idx, int* status, void* stream) {
    BVC_REQUIRE(mask && vis_idx && msk_idx && status, "op_mask_index: null argument");
    return launch_mask_index(mask, B, L, nvis, nmask, vis_idx, msk_idx, status, (hipStream_t)stream);
}
int bvc_op_gather_patches(const float* clip, const int* vis_idx, void* A, int B, int nvis, int T, int C, int H, int W, int ts,
                          int ps, void* stream) {
    BVC_REQUIRE(clip && vis_idx && A, "op_gather_patches: null argument");
    return launch_gather_patches(pixels_f32(clip), vis_idx, (bf16_t*)A, B, nvis, PatchGeom{T, C, H, W, ts, ps}, (hipStream_t)stream);
}
int bvc_op_pixel_labels(const float* clip, const int* msk_idx, float* labels, int B, int nmask, int T, int C, int H, int W, int ts,
                        int ps, int norm_pix, void* stream) {
    BVC_REQUIRE(clip && msk_idx && labels, "op_pixel_labels: null argument");
    return launch_labels(pixels_f32(clip), msk_idx, labels, B, nmask, PatchGeom{T, C, H, W, ts, ps}, norm_pix, (hipStream_t)stream);
}

}  // extern "C"
