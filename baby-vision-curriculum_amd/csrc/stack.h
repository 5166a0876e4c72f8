// Pre-LN transformer stack shared by the VideoMAE encoder/decoder and the JEPA encoder/predictor (host code).
// One layer = LN -> fused qkv Linear -> attention -> proj (+residual) -> LN -> fc1 + GELU -> fc2 (+residual), which is
// VideoMAELayer (HF:339-357) and Block (pretraining/predictive/vision_transformer.py:213-231) alike.
#pragma once
#include <string>
#include <vector>

#include "attention.h"
#include "gemm.h"
#include "rowops.h"

namespace bvc {

#define TRY(expr)                      \
    do {                               \
        int _rc = (expr);              \
        if (_rc != BVC_OK) return _rc; \
    } while (0)

// offsets (elements) of one layer's parameters in a flat f32 buffer; q | k | v weights are one [3d][d] matrix
struct LayerOff {
    int64_t ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, w1, b1, w2, b2, end;
};

struct ParamEntry {
    std::string name;
    int64_t offset, numel;
    int ndim;
    int64_t shape[5];
};

struct ParamTable {
    std::vector<ParamEntry> entries;
    int64_t total = 0;
    int64_t add(const std::string& name, std::initializer_list<int64_t> shp);
};

// saved activations of one transformer layer
struct LayerAct {
    float* x_in;      // f32 [M][D]  layer input (residual stream)
    float* h;         // f32 [M][D]  after attention residual
    bf16_t* ln1o;     // bf16 [M][D]
    bf16_t* qkv;      // bf16 [M][3D]
    bf16_t* ctx;      // bf16 [M][D]
    float* lse;       // f32 [B*H][N]
    bf16_t* ln2o;     // bf16 [M][D]
    bf16_t* pre;      // bf16 [M][I]: gelu'(fc1 pre-activation), written by the forward fc1 epilogue for the backward dX-fc2 product
    bf16_t* act;      // bf16 [M][I]
    float *mean1, *rstd1, *mean2, *rstd2;
};

struct Stack {
    int D, I, H, nlayers;
    float eps;
    std::vector<LayerAct> act;
    float* x_out;     // f32 [M][D] output of the last layer
    // attention width: heads of hd = D / H dims run at hdp dims (hdp == hd unless hd is not 32 / 64, e.g. 24 -> 32, zero padded);
    // Da = H * hdp is the row width of qkv thirds and of ctx.  Scratch below exists only when hdp != hd.
    int hd, hdp, Da;
    // set by a layer whose fc2 epilogue already produced the NEXT layer's first LayerNorm (ln1o / mean1 / rstd1), cleared by the consumer
    bool ln1_ready = false;
    bf16_t *wqkv_pad = nullptr, *wo_pad = nullptr;             // bf16 [3 Da][D], [D][Da]
    float *bqkv_pad = nullptr;                                 // f32 [3 Da]
    float *gwqkv_pad = nullptr, *gbqkv_pad = nullptr, *gwo_pad = nullptr;   // f32 gradients in the padded layout
};

// device allocations owned by a context
struct Arena {
    std::vector<void*> ptrs;
    template <typename T>
    int alloc(T** p, size_t count) {
        void* q = nullptr;
        BVC_CHECK_HIP(hipMalloc(&q, count * sizeof(T) + 256));
        ptrs.push_back(q);
        *p = reinterpret_cast<T*>(q);
        return BVC_OK;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

// per-call parameter views + the backward scratch shared by every layer of a context
struct Work {
    const float* params = nullptr;   // f32 flat parameters of the current call
    const bf16_t* wbf = nullptr;     // their bf16 shadow
    // dY operands of the weight-gradient products are multi-buffered so that the grouped dW launch of backward step s may
    // run on the side stream while the main stream works on step s+1 (opt-in, BVC_DW_OVERLAP=1; no gain measured)
    bf16_t *dyb[3] = {}, *dhb[2] = {}, *dqkv[2] = {}, *dh[2] = {};
    bf16_t *dln = nullptr, *dctx = nullptr;
    float *delta = nullptr, *ln_part = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    bool overlap = false;
    int seq = 0;
    bool join_pending[2] = {false, false};
    int64_t pend_lo[2], pend_hi[2];
};

LayerOff add_layer_params(ParamTable& t, const std::string& prefix, int64_t d, int64_t inter, bool hf_names);
int alloc_stack(Arena& a, Stack& s, int D, int I, int H, int nlayers, float eps, size_t M, size_t BHN);
// MD = max tokens x width, MI = max tokens x intermediate, over every stack that will use this scratch
int alloc_work(Arena& a, Work& w, size_t MD, size_t MI, size_t delta_elems, size_t lnpart_elems);
void free_work(Work& w);

GemmProblem gemm(const bf16_t* A, size_t a_elems, int lda, const bf16_t* B, size_t b_elems, int ldb, int M, int N, int K,
                 int epi, void* C, int ldc);
int plan_dw(GemmProblem* g, int n);
int join_side(Work& w, int parity, hipStream_t st, bvc_bucket_fn on_bucket, void* user);
void begin_backward(Work& w);
// next: the parameters of layer li + 1 when its activations are s.act[li + 1] (its first LayerNorm may then be produced by this
// layer's fc2 epilogue), nullptr for the last layer or when the caller reuses one set of activations
int layer_forward(Work& w, Stack& s, int li, const LayerOff& o, const float* x_in, float* x_out, int B, int N, hipStream_t st,
                  const LayerOff* next = nullptr);
// Do the LayerNorms of this stack run inside the epilogues of the 384-wide products next to them (gemm8.hip, EPI_RESID_LN / EPI_DLN)?
bool fuse_row_ln(const Stack& s, int M);
int layer_backward(Work& w, Stack& s, int li, const LayerOff& o, const float* x_in, float* dres, float* G, int B, int N,
                   hipStream_t st, bvc_bucket_fn on_bucket, void* user);

}  // namespace bvc
