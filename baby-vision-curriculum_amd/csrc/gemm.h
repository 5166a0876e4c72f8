// bf16 MFMA GEMM for gfx950: problem descriptors and launcher (internal to libbvc_hip.so).
#pragma once
#include "common.h"
#include "../../include/bvc.h"

namespace bvc {

// C[M,N] = epilogue(alpha * sum_k A(m,k) * B(k,n))
// Operand storage ("layout"):
//   GEMM_NT : A stored [M][lda] (k contiguous), B stored [N][ldb] (k contiguous)   y = x W^T   (forward Linear)
//   GEMM_NN : A stored [M][lda] (k contiguous), B stored [K][ldb] (n contiguous)   dx = dy W   (input gradient)
//   GEMM_TN : A stored [K][lda] (m contiguous), B stored [K][ldb] (n contiguous)   dW = dy^T x (weight gradient)
enum GemmLayout { GEMM_NT = BVC_GEMM_NT, GEMM_NN = BVC_GEMM_NN, GEMM_TN = BVC_GEMM_TN };

enum GemmEpilogue {
    EPI_F32 = BVC_EPI_F32,          // C f32 = v (+bias)            ; split_k>1: atomicAdd into C (C pre-zeroed / accumulating)
    EPI_BF16 = BVC_EPI_BF16,        // C bf16 = v (+bias)
    EPI_GELU = BVC_EPI_GELU,        // C bf16 = gelu'(v+bias) (what the backward product multiplies by), C2 bf16 = gelu(v+bias)
    EPI_RESID = BVC_EPI_RESID,      // C f32 = resid + v + bias      (resid may alias C); split_k>1: atomicAdd, resid must alias C
    EPI_POS = BVC_EPI_POS,          // C f32 = v + bias + pos[rowtok[m]][n]
    EPI_E2D = BVC_EPI_E2D,          // C f32[(m/rin)*rout + m%rin][n] = v + pos[rowtok[m]][n]
    EPI_LOSS = BVC_EPI_LOSS,        // d = v + bias - labels[m][n]; C bf16 = d; C2 f32 = v+bias (optional); partial[tile] = sum d^2
    EPI_DGELU = BVC_EPI_DGELU,      // C bf16 = v * aux[m][n], aux = the gelu' saved by EPI_GELU
    EPI_F32_BF16 = BVC_EPI_F32_BF16, // C f32 = v (+bias), C2 bf16 = same value
    EPI_RELU = BVC_EPI_RELU,         // C bf16 = relu(v + bias)
    EPI_DRELU = BVC_EPI_DRELU,       // C bf16 = v * (aux > 0)
    EPI_NCE = BVC_EPI_NCE,           // SimCLR loss partials (no C): see gemm.hip
    EPI_NCE_BWD = BVC_EPI_NCE_BWD,   // C bf16 = d loss / d (cos/T)
    EPI_RESID_LN = BVC_EPI_RESID_LN, // C f32 = resid + v + bias, C2 bf16 = LayerNorm(C) (N = 384: full rows in one tile; gemm8.hip only)
    EPI_DLN = BVC_EPI_DLN            // LayerNorm backward fused into the dX product that feeds it (N = 384; gemm8.hip only)
};

// the public descriptor IS the internal problem record (include/bvc.h)
using GemmProblem = bvc_gemm_desc;

constexpr int kMaxGroup = 4;

// Process-wide switches behind bvc_set_option (include/bvc.h): what the parity tests and the same-process A/B tools flip.
//   gemm8:      0 = the measured selection (pick_gemm8 / plan_dw), 1 = the 256-row persistent kernel for EVERY product it can take
//               (whatever its size: how the tests run the bench's kernel set at oracle-sized batches), -1 = never
//   dw_overlap: 1 = the grouped weight-gradient launch of a layer runs on the context's side stream (no gain measured; A/B only)
//   row_ln:     LayerNorm fused into the neighbouring 384-wide product (gemm8.hip EC 4 / 5): 0 = the measured selection, 1 = whenever
//               the shapes allow, -1 = never (the separate ln_fwd / ln_bwd passes)
//   row_stagger: 1 (default) = the workgroups of a row-epilogue launch (gemm8.hip EC 4 / 5) start spread over one unit time, so that their
//               HBM-heavy epilogues do not all fall together; 0 = all start together (A/B)
struct Options { int gemm8 = 0; int dw_overlap = 0; int row_ln = 0; int row_stagger = 1; };
Options& options();

// bvc_op_gemm_kernel: while `on`, the launchers write the name of the kernel instantiation they would launch (as rocprofv3
// prints it) and launch nothing - what bench.py uses to attribute its per-product timings to the rows of a kernel-stats table.
struct DryRun { bool on = false; char name[160] = ""; };
DryRun& dry_run();

// Launch 1..4 independent problems of the same layout as ONE grid (grouped GEMM) on `stream`.
// tile_cfg: -1 = pick from the tile count; 0 = 128x128, 1 = 128x64, 2 = 64x64.
// stages: -1 = pick from the grid size; 2..4 = LDS ring depth (K-steps of LDS-DMA in flight + 1).
int launch_gemm(const GemmProblem* probs, int nprob, GemmLayout layout, int tile_cfg, hipStream_t stream, int stages = -1);

// number of loss partials an EPI_LOSS problem writes with the tile config launch_gemm would pick
int gemm_num_tiles(const GemmProblem& p, int tile_cfg);
int gemm_pick_tile(const GemmProblem* probs, int nprob, int tile_cfg);
bool gemm_row_ln_ok(int M, int N, int K);
// scratch floats an EPI_DLN problem needs in ln_part (one [2][384] row per workgroup of the one-workgroup-per-CU grid)
constexpr size_t kRowLnPartFloats = 512 * 2 * 384;

}  // namespace bvc
