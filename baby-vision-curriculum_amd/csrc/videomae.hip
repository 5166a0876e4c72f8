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

#include <string>
#include <vector>

#include "attention.h"
#include "gemm.h"
#include "rowops.h"

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

#define TRY(expr)                     \
    do {                              \
        int _rc = (expr);             \
        if (_rc != BVC_OK) return _rc; \
    } while (0)

// ------------------------------------------------------------------ flat parameter layout
namespace {

struct LayerOff {
    int64_t ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, w1, b1, w2, b2, end;
};

struct ParamEntry {
    std::string name;
    int64_t offset, numel;
    int ndim;
    int64_t shape[5];
};

struct Layout {
    std::vector<ParamEntry> entries;
    int64_t total = 0;
    int64_t pe_w = 0, pe_b = 0, e2d_w = 0, mask_token = 0, norm_w = 0, norm_b = 0, head_w = 0, head_b = 0;
    std::vector<LayerOff> enc, dec;

    int64_t add(const std::string& name, std::initializer_list<int64_t> shp) {
        ParamEntry e;
        e.name = name;
        e.offset = total;
        e.numel = 1;
        e.ndim = (int)shp.size();
        int i = 0;
        for (auto s : shp) { e.shape[i++] = s; e.numel *= s; }
        for (; i < 5; ++i) e.shape[i] = 1;
        entries.push_back(e);
        total += e.numel;
        return e.offset;
    }
    LayerOff add_layer(const std::string& p, int64_t d, int64_t inter) {
        LayerOff o;
        o.ln1w = add(p + "layernorm_before.weight", {d});
        o.ln1b = add(p + "layernorm_before.bias", {d});
        o.wqkv = add(p + "attention.attention.query.weight", {d, d});   // q | k | v contiguous = one [3d][d] matrix
        add(p + "attention.attention.key.weight", {d, d});
        add(p + "attention.attention.value.weight", {d, d});
        o.bqkv = add(p + "attention.attention.query.bias", {d});
        add(p + "attention.attention.key.bias", {d});
        add(p + "attention.attention.value.bias", {d});
        o.wo = add(p + "attention.output.dense.weight", {d, d});
        o.bo = add(p + "attention.output.dense.bias", {d});
        o.ln2w = add(p + "layernorm_after.weight", {d});
        o.ln2b = add(p + "layernorm_after.bias", {d});
        o.w1 = add(p + "intermediate.dense.weight", {inter, d});
        o.b1 = add(p + "intermediate.dense.bias", {inter});
        o.w2 = add(p + "output.dense.weight", {d, inter});
        o.b2 = add(p + "output.dense.bias", {d});
        o.end = total;
        return o;
    }
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
        L.enc.push_back(L.add_layer("videomae.encoder.layer." + std::to_string(i) + ".", D, c.intermediate_size));
    L.e2d_w = L.add("encoder_to_decoder.weight", {Dd, D});
    L.mask_token = L.add("mask_token", {1, 1, Dd});
    for (int i = 0; i < c.decoder_num_hidden_layers; ++i)
        L.dec.push_back(L.add_layer("decoder.decoder_layers." + std::to_string(i) + ".", Dd, c.decoder_intermediate_size));
    L.norm_w = L.add("decoder.norm.weight", {Dd});
    L.norm_b = L.add("decoder.norm.bias", {Dd});
    L.head_w = L.add("decoder.head.weight", {P, Dd});
    L.head_b = L.add("decoder.head.bias", {P});
    return L;
}

// saved activations of one transformer layer
struct LayerAct {
    float* x_in;      // f32 [M][D]  layer input (residual stream)
    float* h;         // f32 [M][D]  after attention residual
    bf16_t* ln1o;     // bf16 [M][D]
    bf16_t* qkv;      // bf16 [M][3D]
    bf16_t* ctx;      // bf16 [M][D]
    float* lse;       // f32 [B*H][N]
    bf16_t* ln2o;     // bf16 [M][D]
    bf16_t* pre;      // bf16 [M][I]
    bf16_t* act;      // bf16 [M][I]
    float *mean1, *rstd1, *mean2, *rstd2;
};

struct Stack {   // encoder or decoder
    int D, I, H, nlayers;
    std::vector<LayerAct> act;
    float* x_out;     // f32 [M][D] output of the last layer
};

}  // namespace

struct bvc_ctx {
    bvc_videomae_config cfg;
    Layout lay;
    int max_batch, nmask, nvis, L, P, Kp;
    std::vector<void*> allocs;
    // constants
    float *pos_enc, *pos_dec;
    // per-step state
    int batch = 0;
    bool have_forward = false;
    bf16_t* wbf;       // bf16 copy of the flat parameters
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
    // backward scratch (sized for the larger of encoder / decoder)
    float *dres_enc, *dres_dec;
    // dY operands of the weight-gradient products are multi-buffered: the grouped dW launch of backward step s runs
    // on the side stream while the main stream already works on step s+1 (buffers are reused at step s+2 / s+3)
    bf16_t *dyb[3], *dhb[2], *dqkv[2], *dh[2];
    bf16_t *dln, *dctx, *de2d;
    float* delta;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    bool overlap = true;
    int seq = 0;                       // backward step counter of the current call
    bool join_pending[2] = {false, false};
    int64_t pend_lo[2], pend_hi[2];    // gradient range reported once the side stream's launch is fenced
    float* ln_part;    // per-workgroup LayerNorm parameter-gradient partials
};

namespace {

template <typename T>
int dev_alloc(bvc_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    BVC_CHECK_HIP(hipMalloc(&q, count * sizeof(T) + 256));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return BVC_OK;
}

int alloc_stack(bvc_ctx* c, Stack& s, int D, int I, int H, int nlayers, size_t M, size_t BHN) {
    s.D = D; s.I = I; s.H = H; s.nlayers = nlayers;
    s.act.resize(nlayers);
    for (auto& a : s.act) {
        TRY(dev_alloc(c, &a.x_in, M * D));
        TRY(dev_alloc(c, &a.h, M * D));
        TRY(dev_alloc(c, &a.ln1o, M * D));
        TRY(dev_alloc(c, &a.qkv, M * 3 * D));
        TRY(dev_alloc(c, &a.ctx, M * D));
        TRY(dev_alloc(c, &a.lse, BHN));
        TRY(dev_alloc(c, &a.ln2o, M * D));
        TRY(dev_alloc(c, &a.pre, M * I));
        TRY(dev_alloc(c, &a.act, M * I));
        TRY(dev_alloc(c, &a.mean1, M));
        TRY(dev_alloc(c, &a.rstd1, M));
        TRY(dev_alloc(c, &a.mean2, M));
        TRY(dev_alloc(c, &a.rstd2, M));
    }
    TRY(dev_alloc(c, &s.x_out, M * D));
    return BVC_OK;
}

void sinusoid(std::vector<float>& out, int n, int d) {   // HF:80-91, float64 then cast
    out.resize((size_t)n * d);
    for (int p = 0; p < n; ++p)
        for (int j = 0; j < d; ++j) {
            const double ang = (double)p / pow(10000.0, 2.0 * (j / 2) / (double)d);
            out[(size_t)p * d + j] = (float)((j & 1) ? cos(ang) : sin(ang));
        }
}

GemmProblem gemm(const bf16_t* A, size_t a_elems, int lda, const bf16_t* B, size_t b_elems, int ldb, int M, int N, int K,
                 int epi, void* C, int ldc) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.a_bytes = (uint32_t)(a_elems * 2); p.b_bytes = (uint32_t)(b_elems * 2);
    p.alpha = 1.f; p.epi = epi; p.split_k = 1; p.C = C; p.ldc = ldc;
    return p;
}

// Tile and split-K choice for a group of weight-gradient products (contraction over all tokens).
// Measured (profiles/r01_b_microbench.json): when the 128x128 tiles alone cover the chip (encoder layer: 432)
// use them unsplit; otherwise 64x64 tiles with just enough K-splits for ~850 workgroups (decoder layer: 432 x 2).
int plan_dw(GemmProblem* g, int n) {
    int t128 = 0, t64 = 0;
    for (int i = 0; i < n; ++i) {
        t128 += ((g[i].M + 127) / 128) * ((g[i].N + 127) / 128);
        t64 += ((g[i].M + 63) / 64) * ((g[i].N + 63) / 64);
    }
    if (t128 >= 400) return 0;
    for (int i = 0; i < n; ++i) {
        const int ksteps = (g[i].K + 63) / 64;
        int s = (864 + t64 / 2) / t64;
        s = std::min(s, std::max(1, ksteps / 16));
        g[i].split_k = std::max(1, s);
    }
    return 2;
}

int layer_forward(bvc_ctx* c, Stack& s, int li, const LayerOff& o, float* x_in_external, float* x_out, int B, int N, hipStream_t st) {
    LayerAct& a = s.act[li];
    const int D = s.D, I = s.I, M = B * N;
    const float* P = c->params;
    const bf16_t* W = c->wbf;
    const float* x_in = x_in_external;
    const float eps = c->cfg.layer_norm_eps;
    TRY(launch_ln_fwd(x_in, identity_rows(), P + o.ln1w, P + o.ln1b, a.ln1o, a.mean1, a.rstd1, M, D, eps, st));
    {
        GemmProblem p = gemm(a.ln1o, (size_t)M * D, D, W + o.wqkv, (size_t)3 * D * D, D, M, 3 * D, D, EPI_BF16, a.qkv, 3 * D);
        p.bias = P + o.bqkv;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    TRY(launch_attn_fwd(a.qkv, a.ctx, a.lse, B, N, s.H, st));
    {
        GemmProblem p = gemm(a.ctx, (size_t)M * D, D, W + o.wo, (size_t)D * D, D, M, D, D, EPI_RESID, a.h, D);
        p.bias = P + o.bo; p.resid = x_in;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    TRY(launch_ln_fwd(a.h, identity_rows(), P + o.ln2w, P + o.ln2b, a.ln2o, a.mean2, a.rstd2, M, D, eps, st));
    {
        GemmProblem p = gemm(a.ln2o, (size_t)M * D, D, W + o.w1, (size_t)I * D, D, M, I, D, EPI_GELU, a.pre, I);
        p.bias = P + o.b1; p.C2 = a.act;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    {
        GemmProblem p = gemm(a.act, (size_t)M * I, I, W + o.w2, (size_t)D * I, I, M, D, I, EPI_RESID, x_out, D);
        p.bias = P + o.b2; p.resid = a.h;
        TRY(launch_gemm(&p, 1, GEMM_NT, -1, st));
    }
    return BVC_OK;
}

// Fence a side-stream weight-gradient launch into the main stream and report its gradient range.
int join_side(bvc_ctx* c, int parity, hipStream_t st, bvc_bucket_fn on_bucket, void* user) {
    if (!c->join_pending[parity]) return BVC_OK;
    BVC_CHECK_HIP(hipStreamWaitEvent(st, c->ev_join[parity], 0));
    c->join_pending[parity] = false;
    if (on_bucket) on_bucket(c->pend_lo[parity], c->pend_hi[parity] - c->pend_lo[parity], user);
    return BVC_OK;
}

// dres (f32 [M][D]) holds d/d(layer output) on entry and d/d(layer input) on exit; dyb[seq % 3] is its bf16 copy.
// The four weight gradients (+ bias gradients) of the layer are one grouped launch on the side stream, overlapping the
// next layer's dX chain; its gradient range [o.ln1w, o.end) is reported when that launch has been fenced (two steps later).
int layer_backward(bvc_ctx* c, Stack& s, int li, const LayerOff& o, const float* x_in, float* dres, float* G, int B, int N,
                   hipStream_t st, bvc_bucket_fn on_bucket, void* user) {
    LayerAct& a = s.act[li];
    const int D = s.D, I = s.I, M = B * N;
    const float* P = c->params;
    const bf16_t* W = c->wbf;
    const int q = c->seq, par = q & 1;
    bf16_t* dyb = c->dyb[q % 3];
    bf16_t* dyb_next = c->dyb[(q + 1) % 3];
    bf16_t *dh = c->dh[par], *dhb = c->dhb[par], *dqkv = c->dqkv[par];
    // the buffers of this parity were last read by the side launch of step q-2
    TRY(join_side(c, par, st, on_bucket, user));
    // MLP
    {
        GemmProblem p = gemm(dyb, (size_t)M * D, D, W + o.w2, (size_t)D * I, I, M, I, D, EPI_DGELU, dh, I);
        p.aux = a.pre; p.ldaux = I;
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    {
        GemmProblem p = gemm(dh, (size_t)M * I, I, W + o.w1, (size_t)I * D, D, M, D, I, EPI_BF16, c->dln, D);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    TRY(launch_ln_bwd(c->dln, a.h, identity_rows(), a.mean2, a.rstd2, P + o.ln2w, dres, 1, dhb, G + o.ln2w, G + o.ln2b, c->ln_part, M, D, st));
    // attention
    {
        GemmProblem p = gemm(dhb, (size_t)M * D, D, W + o.wo, (size_t)D * D, D, M, D, D, EPI_BF16, c->dctx, D);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    TRY(launch_attn_bwd(a.qkv, a.ctx, c->dctx, a.lse, c->delta, dqkv, B, N, s.H, st));
    {
        GemmProblem p = gemm(dqkv, (size_t)M * 3 * D, 3 * D, W + o.wqkv, (size_t)3 * D * D, D, M, D, 3 * D, EPI_BF16, c->dln, D);
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    // the four weight gradients of the layer as one grouped launch:  dW = dY^T X,  db = column sums of dY
    hipStream_t ws = st;
    if (c->overlap) {
        BVC_CHECK_HIP(hipEventRecord(c->ev_fork, st));
        BVC_CHECK_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        ws = c->side;
    }
    {
        GemmProblem g[4];
        g[0] = gemm(dyb, (size_t)M * D, D, a.act, (size_t)M * I, I, D, I, M, EPI_F32, G + o.w2, I);
        g[1] = gemm(dh, (size_t)M * I, I, a.ln2o, (size_t)M * D, D, I, D, M, EPI_F32, G + o.w1, D);
        g[2] = gemm(dhb, (size_t)M * D, D, a.ctx, (size_t)M * D, D, D, D, M, EPI_F32, G + o.wo, D);
        g[3] = gemm(dqkv, (size_t)M * 3 * D, 3 * D, a.ln1o, (size_t)M * D, D, 3 * D, D, M, EPI_F32, G + o.wqkv, D);
        g[0].rowsum = G + o.b2;     // bias gradients ride along as one extra MFMA column each
        g[1].rowsum = G + o.b1;
        g[2].rowsum = G + o.bo;
        g[3].rowsum = G + o.bqkv;
        const int tile = plan_dw(g, 4);
        TRY(launch_gemm(g, 4, GEMM_TN, tile, ws));
    }
    TRY(launch_ln_bwd(c->dln, x_in, identity_rows(), a.mean1, a.rstd1, P + o.ln1w, dres, 1, dyb_next, G + o.ln1w, G + o.ln1b, c->ln_part, M, D, st));
    if (c->overlap) {
        BVC_CHECK_HIP(hipEventRecord(c->ev_join[par], c->side));
        c->join_pending[par] = true;
        c->pend_lo[par] = o.ln1w;
        c->pend_hi[par] = o.end;
    } else if (on_bucket) {
        on_bucket(o.ln1w, o.end - o.ln1w, user);
    }
    c->seq = q + 1;
    return BVC_OK;
}

}  // namespace

// ============================================================================ C ABI
extern "C" {

const char* bvc_last_error(void) { return bvc::last_error(); }
const char* bvc_version(void) { return "gfx950;bvc-hip-r1"; }

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
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int i = 0; i < 2; ++i) if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    for (void* p : c->allocs) (void)hipFree(p);
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
    A(dev_alloc(c, &c->pos_enc, (size_t)c->L * D));
    A(dev_alloc(c, &c->pos_dec, (size_t)c->L * Dd));
    A(dev_alloc(c, &c->wbf, (size_t)c->lay.total));
    A(dev_alloc(c, &c->vis_idx, Mv));
    A(dev_alloc(c, &c->msk_idx, Mm));
    A(dev_alloc(c, &c->status, 4));
    A(dev_alloc(c, &c->Ape, Mv * c->Kp));
    A(alloc_stack(c, c->enc, D, I, H, cfg->num_hidden_layers, Mv, B * H * c->nvis));
    A(alloc_stack(c, c->dec, Dd, Id, Hd, cfg->decoder_num_hidden_layers, Md, B * Hd * c->L));
    A(dev_alloc(c, &c->xe_bf, Mv * D));
    A(dev_alloc(c, &c->meanf, Mm));
    A(dev_alloc(c, &c->rstdf, Mm));
    A(dev_alloc(c, &c->lnf, Mm * Dd));
    A(dev_alloc(c, &c->labels, Mm * c->P));
    A(dev_alloc(c, &c->diff, Mm * c->P));
    A(dev_alloc(c, &c->partial, (Mm / 64 + 2) * (c->P / 64 + 2)));
    A(dev_alloc(c, &c->dres_enc, Mv * D));
    A(dev_alloc(c, &c->dres_dec, Md * Dd));
    const size_t MD = std::max(Mv * D, Md * Dd), MI = std::max(Mv * I, Md * Id);
    for (int i = 0; i < 3; ++i) A(dev_alloc(c, &c->dyb[i], MD));
    for (int i = 0; i < 2; ++i) {
        A(dev_alloc(c, &c->dhb[i], MD));
        A(dev_alloc(c, &c->dqkv[i], 3 * MD));
        A(dev_alloc(c, &c->dh[i], MI));
    }
    A(dev_alloc(c, &c->dln, MD));
    A(dev_alloc(c, &c->dctx, MD));
    A(dev_alloc(c, &c->de2d, Mv * Dd));
    A(dev_alloc(c, &c->delta, std::max(B * H * c->nvis, B * Hd * c->L)));
    A(dev_alloc(c, &c->ln_part, std::max(ln_bwd_workspace_floats_upto((int)Mv, D), ln_bwd_workspace_floats_upto((int)Md, Dd))));
#undef A
    // Measured on MI355X (B=16): running the grouped dW launch on a side stream next to the dX chain gains nothing
    // (1398 vs 1419 clips/s) - each GEMM already holds all of a CU's LDS - so it is opt-in for experiments.
    c->overlap = getenv("BVC_DW_OVERLAP") != nullptr;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join[1], hipEventDisableTiming) != hipSuccess) {
        set_error("create: side stream / events");
        return fail(BVC_ERR_HIP);
    }
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
    BVC_REQUIRE(c && pixels && mask && params && loss, "forward: null argument");
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
    const PatchGeom pg{cf.num_frames, cf.num_channels, cf.image_size, cf.image_size, cf.tubelet_size, cf.patch_size};

    TRY(launch_cast_bf16(params, c->wbf, (size_t)L.total, st));
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
        TRY(layer_forward(c, c->enc, i, L.enc[i], c->enc.act[i].x_in, xo, B, nvis, st));
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
        TRY(layer_forward(c, c->dec, i, L.dec[i], c->dec.act[i].x_in, xo, B, Lq, st));
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
    c->seq = 0;
    c->join_pending[0] = c->join_pending[1] = false;

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
        GemmProblem p = gemm(c->diff, (size_t)Mm * P, P, W + L.head_w, (size_t)P * Dd, Dd, Mm, Dd, P, EPI_BF16, c->dln, Dd);
        p.alpha = cmse; p.alpha_dev = grad_loss;
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    // visible rows of the decoder stream receive no gradient from the head
    BVC_CHECK_HIP(hipMemsetAsync(c->dres_dec, 0, (size_t)Md * Dd * 4, st));
    BVC_CHECK_HIP(hipMemsetAsync(c->dyb[0], 0, (size_t)Md * Dd * 2, st));
    const RowMap tail{nmask, Lq, nvis};
    TRY(launch_ln_bwd(c->dln, c->dec.x_out, tail, c->meanf, c->rstdf, params + L.norm_w, c->dres_dec, 0, c->dyb[0],
                      G + L.norm_w, G + L.norm_b, c->ln_part, Mm, Dd, st));
    bucket(L.norm_w, L.total);
    for (int i = c->dec.nlayers - 1; i >= 0; --i) {
        TRY(layer_backward(c, c->dec, i, L.dec[i], c->dec.act[i].x_in, c->dres_dec, G, B, Lq, st, on_bucket, user));
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
        p.C2 = c->dyb[c->seq % 3];   // last read (as a dW operand) by backward step seq-3, fenced since
        TRY(launch_gemm(&p, 1, GEMM_NN, -1, st));
    }
    bucket(L.e2d_w, L.dec.front().ln1w);
    for (int i = c->enc.nlayers - 1; i >= 0; --i) {
        TRY(layer_backward(c, c->enc, i, L.enc[i], c->enc.act[i].x_in, c->dres_enc, G, B, nvis, st, on_bucket, user));
    }
    // patch embedding: weight and bias only (pixels need no gradient)
    {
        GemmProblem p = gemm(c->dyb[c->seq % 3], (size_t)Mv * D, D, c->Ape, (size_t)Mv * c->Kp, c->Kp, D, c->Kp, Mv, EPI_F32, G + L.pe_w, c->Kp);
        p.rowsum = G + L.pe_b;
        const int tile = plan_dw(&p, 1);
        TRY(launch_gemm(&p, 1, GEMM_TN, tile, st));
    }
    // fence the last side-stream launches (older one first so ranges keep arriving tail-first)
    TRY(join_side(c, c->seq & 1, st, on_bucket, user));
    TRY(join_side(c, (c->seq + 1) & 1, st, on_bucket, user));
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

// ------------------------------------------------------------------ operator-level entry points
int bvc_op_gemm(const bvc_gemm_desc* problems, int count, int layout, int tile_cfg, int stages, void* stream) {
    BVC_REQUIRE(problems, "op_gemm: null problems");
    BVC_REQUIRE(layout >= 0 && layout <= 2, "op_gemm: bad layout %d", layout);
    return launch_gemm(problems, count, (GemmLayout)layout, tile_cfg, (hipStream_t)stream, stages);
}
int bvc_op_gemm_num_tiles(const bvc_gemm_desc* problem, int tile_cfg) {
    if (!problem) return BVC_ERR_INVALID;
    return gemm_num_tiles(*problem, tile_cfg);
}
int bvc_op_attention_fwd(const void* qkv, void* ctx_out, float* lse, int B, int N, int H, void* stream) {
    BVC_REQUIRE(qkv && ctx_out && lse, "op_attention_fwd: null argument");
    return launch_attn_fwd((const bf16_t*)qkv, (bf16_t*)ctx_out, lse, B, N, H, (hipStream_t)stream);
}
int bvc_op_attention_bwd(const void* qkv, const void* ctx_in, const void* dctx, const float* lse, float* delta, void* dqkv,
                         int B, int N, int H, void* stream) {
    BVC_REQUIRE(qkv && ctx_in && dctx && lse && delta && dqkv, "op_attention_bwd: null argument");
    return launch_attn_bwd((const bf16_t*)qkv, (const bf16_t*)ctx_in, (const bf16_t*)dctx, lse, delta, (bf16_t*)dqkv, B, N, H, (hipStream_t)stream);
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
                    const float* found_inf, int write_unscaled_grads, void* stream) {
    BVC_REQUIRE(params && grads && n >= 0, "op_sgd_step: bad argument");
    return launch_sgd_step(params, grads, momentum_buf, (size_t)n, lr, momentum, dampening, weight_decay, nesterov, first_step,
                           maximize, grad_scale, found_inf, write_unscaled_grads, (hipStream_t)stream);
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
    return launch_gather_patches(clip, vis_idx, (bf16_t*)A, B, nvis, PatchGeom{T, C, H, W, ts, ps}, (hipStream_t)stream);
}
int bvc_op_pixel_labels(const float* clip, const int* msk_idx, float* labels, int B, int nmask, int T, int C, int H, int W, int ts,
                        int ps, int norm_pix, void* stream) {
    BVC_REQUIRE(clip && msk_idx && labels, "op_pixel_labels: null argument");
    return launch_labels(clip, msk_idx, labels, B, nmask, PatchGeom{T, C, H, W, ts, ps}, norm_pix, (hipStream_t)stream);
}

}  // extern "C"
