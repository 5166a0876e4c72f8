"""Kernel selection of the GEMM launcher, pinned WITHOUT a GPU: bvc_op_gemm_kernel names the kernel instantiation a problem
would run on (nothing is launched), bvc_op_gemm_plan_dw returns the (tile, K split) plan of a weight-gradient group, and
bvc_set_option switches the 256-row persistent kernel never / measured / forced.  What is pinned is what the profiles under
profiles/r03_* were measured with: a change of the table has to come with new measurements."""
import ctypes

import pytest


def _desc(L, M, N, K, epi, layout):
    d = L.GemmDesc()
    d.A, d.B, d.C = 4096, 8192, 4096          # never dereferenced: nothing is launched
    d.M, d.N, d.K = M, N, K
    d.alpha, d.epi, d.split_k, d.ldc = 1.0, epi, 1, N
    d.a_bytes, d.b_bytes = M * K * 2, N * K * 2
    if layout == 0:
        d.lda, d.ldb = K, K
    elif layout == 1:
        d.lda, d.ldb = K, N
    else:
        d.lda, d.ldb = M, N
    return d


def _name(L, descs, layout, tile=-1):
    arr = (L.GemmDesc * len(descs))(*descs)
    buf = ctypes.create_string_buffer(160)
    L.check(L.lib().bvc_op_gemm_kernel(arr, len(descs), layout, tile, -1, buf, 160), "bvc_op_gemm_kernel")
    return buf.value.decode()


def _dw_group(L, M, D, I):
    return [_desc(L, D, I, M, 0, 2), _desc(L, I, D, M, 0, 2), _desc(L, D, D, M, 0, 2), _desc(L, 3 * D, D, M, 0, 2)]


def _plan(L, descs):
    arr = (L.GemmDesc * len(descs))(*descs)
    tile = L.lib().bvc_op_gemm_plan_dw(arr, len(descs))
    return tile, arr[0].split_k, _name(L, list(arr), 2, tile)


@pytest.fixture
def L(bvc):
    lib = bvc._lib
    old = lib.set_option("gemm8", 0)
    yield lib
    lib.set_option("gemm8", old)


NT, NN = 0, 1
BF16, GELU, RESID, LOSS, DGELU = 1, 2, 3, 6, 7


def test_forward_and_input_gradient_products_by_batch(L):
    # VideoMAE-base at 256 clips: everything long enough runs on the one-workgroup-per-CU kernel
    Md, Me, Mm = 256 * 1568, 256 * 160, 256 * 1408
    assert _name(L, [_desc(L, Md, 1536, 384, GELU, NT)], NT) == "bvc::gemm8_kernel<256, 256, false, false, 0>"     # decoder fc1 + GELU
    assert _name(L, [_desc(L, Me, 2304, 768, BF16, NT)], NT) == "bvc::gemm8_kernel<256, 256, false, false, 0>"     # encoder qkv
    assert _name(L, [_desc(L, Md, 384, 1536, RESID, NT)], NT) == "bvc::gemm8_kernel<256, 128, false, false, 1>"    # decoder fc2 (N = 384)
    assert _name(L, [_desc(L, Md, 1536, 384, DGELU, NN)], NN) == "bvc::gemm8_kernel<256, 256, false, true, 3>"     # decoder dX fc2
    assert _name(L, [_desc(L, Md, 384, 1536, BF16, NN)], NN) == "bvc::gemm8_kernel<256, 128, false, true, 0>"      # decoder dX fc1
    assert _name(L, [_desc(L, Mm, 1536, 384, LOSS, NT)], NT) == "bvc::gemm8_kernel<256, 256, false, false, 1>"     # head + MSE (round 3)
    assert _name(L, [_desc(L, Md, 384, 384, RESID, NT)], NT) == "bvc::gemm_persist_kernel<128, false, 0>"          # decoder proj: K = 384
    # 16 clips (the reference's per-GPU batch): every launch is below the 45-GFLOP gate of gemm8
    Md, Me = 16 * 1568, 16 * 160
    assert _name(L, [_desc(L, Md, 1536, 384, GELU, NT)], NT) == "bvc::gemm_persist_kernel<128, false, 0>"
    assert _name(L, [_desc(L, Me, 2304, 768, BF16, NT)], NT) == "bvc::gemm_kernel<128, 64, false, false, 2, 2, true>"
    assert "gemm8" not in _name(L, [_desc(L, 16 * 1408, 1536, 384, LOSS, NT)], NT)


def test_weight_gradient_plans(L):
    # decoder widths are multiples of 384, not of 256: 128 x 384 tiles (tile config 12), seven K splits = 252 units on 256 CUs
    for clips in (256, 64, 16):
        assert _plan(L, _dw_group(L, clips * 1568, 384, 1536)) == (12, 7, "bvc::gemm8_kernel<128, 384, true, true, 2>"), clips
    # encoder widths fill 256 x 256 tiles: tile config 10, two K splits; short launches stay on the 128 x 128 kernel
    assert _plan(L, _dw_group(L, 256 * 160, 768, 3072)) == (10, 2, "bvc::gemm8_kernel<256, 256, true, true, 2>")
    assert _plan(L, _dw_group(L, 64 * 160, 768, 3072)) == (10, 2, "bvc::gemm8_kernel<256, 256, true, true, 2>")
    tile, split, name = _plan(L, _dw_group(L, 16 * 160, 768, 3072))
    assert tile == 0 and name == "bvc::gemm_kernel<128, 128, true, true, 2, 2, true>"
    tile, split, name = _plan(L, _dw_group(L, 2 * 1568, 384, 1536))          # 2 clips: far below every gate
    assert tile in (0, 1, 2) and "gemm8" not in name
    # ViT-L layers (JEPA context encoder, 256 samples x 100 tokens): 192 tiles of 256 x 256 fill 3/4 of the chip unsplit and overflow it when
    # split - tile config 13 (accumulated outputs) so that gemm8's balanced walk can use the idle quarter (round 4)
    assert _plan(L, _dw_group(L, 256 * 100, 1024, 4096)) == (13, 1, "bvc::gemm8_kernel<256, 256, true, true, 2>")
    # the head's weight gradient (1536 x 384 over the masked tokens) follows the decoder rule
    assert _plan(L, [_desc(L, 1536, 384, 256 * 1408, 0, 2)])[0] == 12


def test_wide_outputs_never_reach_the_register_epilogue(L):
    # class 0 (bf16 outputs in registers) keeps the tile-padded bias vector in 32 KiB of LDS: N > 8192 must not be auto-selected
    # for it (it used to turn an auto-dispatched bvc_op_gemm into a hard error)
    assert "gemm8" not in _name(L, [_desc(L, 65536, 8448, 1024, BF16, NT)], NT)
    assert _name(L, [_desc(L, 65536, 8192, 1024, BF16, NT)], NT) == "bvc::gemm8_kernel<256, 256, false, false, 0>"
    with pytest.raises(L.BvcError):
        _name(L, [_desc(L, 65536, 8448, 1024, BF16, NT)], NT, tile=10)       # asked for explicitly: an error, not a silent fallback


def test_options(L):
    small = [_desc(L, 320, 2304, 768, BF16, NT)]
    assert "gemm8" not in _name(L, small, NT)
    assert L.set_option("gemm8", 1) == 0
    assert _name(L, small, NT) == "bvc::gemm8_kernel<256, 256, false, false, 0>"           # forced: whatever the size
    tile, split, name = _plan(L, _dw_group(L, 2 * 1568, 384, 1536))
    assert (tile, name) == (12, "bvc::gemm8_kernel<128, 384, true, true, 2>") and split >= 1
    assert _plan(L, _dw_group(L, 2 * 160, 768, 3072))[0] == 10
    L.set_option("gemm8", -1)
    assert "gemm8" not in _name(L, [_desc(L, 256 * 1568, 1536, 384, GELU, NT)], NT)
    assert "gemm8" not in _plan(L, _dw_group(L, 256 * 1568, 384, 1536))[2]
    L.set_option("gemm8", 0)
    with pytest.raises(L.BvcError):
        L.set_option("gemm8", 2)
    with pytest.raises(L.BvcError):
        L.set_option("no_such_option", 1)
    assert L.lib().bvc_get_option(b"no_such_option") < 0


def test_a_stationary_and_row_layernorm_selection(L):
    # round 5: K = 384 bf16 products (decoder qkv) move to the A-stationary kernel once gemm8 would have been chosen or M >= 16384 ...
    Md = 256 * 1568
    assert _name(L, [_desc(L, Md, 1152, 384, BF16, NT)], NT) == "bvc::gemm_as_kernel<false, true, false>"
    assert _name(L, [_desc(L, 64 * 1568, 1152, 384, BF16, NT)], NT) == "bvc::gemm_as_kernel<false, true, false>"
    assert "gemm_as" not in _name(L, [_desc(L, 2 * 1568, 1152, 384, BF16, NT)], NT)
    # ... but not with the GELU epilogue (measured slower: profiles/r05_n_*), not for K != 384, not for N beyond 1536
    gelu = _desc(L, Md, 1536, 384, GELU, NT)
    gelu.C2 = 12288                                             # the activation output next to the pre-activation
    assert _name(L, [gelu], NT) == "bvc::gemm8_kernel<256, 256, false, false, 0>"
    assert "gemm_as" not in _name(L, [_desc(L, 256 * 160, 2304, 768, BF16, NT)], NT)
    assert "gemm_as" not in _name(L, [_desc(L, Md, 1664, 384, BF16, NT)], NT)
    assert _name(L, [gelu], NT, tile=15) == "bvc::gemm_as_kernel<true, true, false>"                                 # asked for explicitly
    with pytest.raises(L.BvcError):
        _name(L, [_desc(L, Md, 1152, 768, BF16, NT)], NT, tile=15)
    L.set_option("gemm8", -1)
    assert "gemm_as" not in _name(L, [_desc(L, Md, 1152, 384, BF16, NT)], NT)
    L.set_option("gemm8", 0)
    # LayerNorm fused into the 384-wide products: a stack-level choice (bvc_op_row_ln_selected mirrors stack.hip's fuse_row_ln)
    sel = L.lib().bvc_op_row_ln_selected
    assert sel(256 * 1568, 384, 1536, 6) == 1 and sel(64 * 1568, 384, 1536, 6) == 1 and sel(16 * 1568, 384, 1536, 6) == 1
    assert sel(8 * 1568, 384, 1536, 6) == 0                     # below 128 row tiles of 128: the separate kernels win
    assert sel(256 * 160, 768, 3072, 12) == 0                   # a 768-wide row does not fit one tile
    assert sel(256 * 100, 384, 1536, 12) == 1                   # JEPA ViT-B predictor (head_dim 32) qualifies by width alone
    old = L.set_option("row_ln", -1)
    assert sel(256 * 1568, 384, 1536, 6) == 0
    L.set_option("row_ln", 1)
    assert sel(2 * 1568, 384, 1536, 6) == 1                     # forced: whatever the size
    L.set_option("row_ln", old)
