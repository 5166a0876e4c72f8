"""Thin Python wrappers over the operator-level C ABI (include/bvc.h) used by the host-side modules."""
import ctypes

import torch

from . import _lib

NT, NN, TN = 0, 1, 2
EPI = dict(F32=0, BF16=1, GELU=2, RESID=3, POS=4, E2D=5, LOSS=6, DGELU=7, F32_BF16=8, RELU=9, DRELU=10, NCE=11, NCE_BWD=12,
           RESID_LN=13, DLN=14)


def _p(t):
    return t.data_ptr() if t is not None else None


def gemm_desc(A, B, M, N, K, epi, C, ldc=None, lda=None, ldb=None, alpha=1.0, alpha_dev=None, split_k=1, C2=None, bias=None,
              resid=None, aux=None, labels=None, partial=None, rowsum=None, rowtok=None, pos=None, rin=0, rout=0, ln_gamma=None,
              ln_beta=None, ln_mean=None, ln_rstd=None, ln_eps=0.0, ln_x=None, ln_part=None, ln_dgamma=None, ln_dbeta=None):
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
    d.rowtok, d.pos, d.rin, d.rout = _p(rowtok), _p(pos), rin, rout
    # BVC_EPI_RESID_LN / BVC_EPI_DLN: the LayerNorm fused into a 384-wide product (include/bvc.h)
    d.ln_gamma, d.ln_beta, d.ln_mean, d.ln_rstd, d.ln_eps = _p(ln_gamma), _p(ln_beta), _p(ln_mean), _p(ln_rstd), ln_eps
    d.ln_x, d.ln_part, d.ln_dgamma, d.ln_dbeta = _p(ln_x), _p(ln_part), _p(ln_dgamma), _p(ln_dbeta)
    return d


def gemm(desc, layout, tile=-1, stages=-1):
    """One problem (a GemmDesc) or a group of up to four (a list) as ONE launch."""
    descs = list(desc) if isinstance(desc, (list, tuple)) else [desc]
    arr = (_lib.GemmDesc * len(descs))(*descs)
    _lib.check(_lib.lib().bvc_op_gemm(arr, len(descs), layout, tile, stages, _lib.current_stream_ptr()), "bvc_op_gemm")


def gemm_kernel_name(desc, layout, tile=-1, stages=-1):
    """The kernel instantiation `gemm` would launch for these problems, as rocprofv3 names it (bvc_op_gemm_kernel; launches nothing)."""
    descs = list(desc) if isinstance(desc, (list, tuple)) else [desc]
    arr = (_lib.GemmDesc * len(descs))(*descs)
    buf = ctypes.create_string_buffer(160)
    _lib.check(_lib.lib().bvc_op_gemm_kernel(arr, len(descs), layout, tile, stages, buf, 160), "bvc_op_gemm_kernel")
    return buf.value.decode()


def plan_dw(descs):
    """(tile config, split_k) the step uses for a group of weight-gradient products; split_k is written into the descriptors."""
    arr = (_lib.GemmDesc * len(descs))(*descs)
    tile = _lib.lib().bvc_op_gemm_plan_dw(arr, len(descs))
    for d, a in zip(descs, arr):
        d.split_k = a.split_k
    return tile, arr[0].split_k


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
