"""Thin Python wrappers over the operator-level C ABI (include/bvc.h) used by the host-side modules."""
import ctypes

import torch

from . import _lib

NT, NN, TN = 0, 1, 2
EPI = dict(F32=0, BF16=1, GELU=2, RESID=3, POS=4, E2D=5, LOSS=6, DGELU=7, F32_BF16=8, RELU=9, DRELU=10, NCE=11, NCE_BWD=12)


def _p(t):
    return t.data_ptr() if t is not None else None


def gemm_desc(A, B, M, N, K, epi, C, ldc=None, lda=None, ldb=None, alpha=1.0, alpha_dev=None, split_k=1, C2=None, bias=None,
              resid=None, aux=None, labels=None, partial=None, rowsum=None):
    d = _lib.GemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else A.shape[-1]
    d.ldb = ldb if ldb is not None else B.shape[-1]
    d.a_bytes, d.b_bytes = A.numel() * 2, B.numel() * 2
    d.alpha, d.alpha_dev = alpha, _p(alpha_dev)
    d.epi, d.split_k = epi, split_k
    d.C = _p(C)
    d.ldc = ldc if ldc is not None else (C.shape[-1] if C is not None else N)
    d.C2, d.bias, d.resid, d.aux = _p(C2), _p(bias), _p(resid), _p(aux)
    d.ldaux = aux.shape[-1] if aux is not None else 0
    d.labels, d.partial, d.rowsum = _p(labels), _p(partial), _p(rowsum)
    return d


def gemm(desc, layout, tile=-1, stages=-1):
    arr = (_lib.GemmDesc * 1)(desc)
    _lib.check(_lib.lib().bvc_op_gemm(arr, 1, layout, tile, stages, _lib.current_stream_ptr()), "bvc_op_gemm")


def num_tiles(desc, tile=-1):
    return _lib.lib().bvc_op_gemm_num_tiles(ctypes.byref(desc), tile)


def cast_bf16(x):
    """f32 -> bf16 copy (round to nearest even) by the library's cast kernel."""
    x = x.contiguous()
    if x.dtype == torch.bfloat16:
        return x
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().bvc_op_cast_bf16(x.float().data_ptr() if x.dtype != torch.float32 else x.data_ptr(), out.data_ptr(),
                                           x.numel(), _lib.current_stream_ptr()), "bvc_op_cast_bf16")
    return out
