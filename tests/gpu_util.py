"""Helpers for the -m gpu parity tests: thin Python over the C ABI of libbvc_hip.so (include/bvc.h)."""
import ctypes
import os

import torch

import __graft_entry__ as ge

bvc = ge.load_package()
L = bvc._lib
NT, NN, TN = 0, 1, 2
EPI = dict(F32=0, BF16=1, GELU=2, RESID=3, POS=4, E2D=5, LOSS=6, DGELU=7, F32_BF16=8, RESID_LN=13, DLN=14)


def stream():
    return L.current_stream_ptr()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def gemm_desc(A, B, M, N, K, epi, C, ldc=None, lda=None, ldb=None, alpha=1.0, alpha_dev=None, split_k=1, C2=None,
              bias=None, resid=None, aux=None, rowtok=None, pos=None, labels=None, partial=None, rin=0, rout=0, rowsum=None,
              ln_gamma=None, ln_beta=None, ln_mean=None, ln_rstd=None, ln_eps=0.0, ln_x=None, ln_part=None, ln_dgamma=None, ln_dbeta=None):
    d = L.GemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else A.shape[-1]
    d.ldb = ldb if ldb is not None else B.shape[-1]
    d.a_bytes, d.b_bytes = A.numel() * 2, B.numel() * 2
    d.alpha = alpha
    d.alpha_dev = alpha_dev.data_ptr() if alpha_dev is not None else None
    d.epi, d.split_k = epi, split_k
    d.C = C.data_ptr()
    d.ldc = ldc if ldc is not None else C.shape[-1]
    d.C2 = C2.data_ptr() if C2 is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.aux = aux.data_ptr() if aux is not None else None
    d.ldaux = aux.shape[-1] if aux is not None else 0
    d.rowtok = rowtok.data_ptr() if rowtok is not None else None
    d.pos = pos.data_ptr() if pos is not None else None
    d.labels = labels.data_ptr() if labels is not None else None
    d.partial = partial.data_ptr() if partial is not None else None
    d.rin, d.rout = rin, rout
    d.rowsum = rowsum.data_ptr() if rowsum is not None else None
    for name, t in (("ln_gamma", ln_gamma), ("ln_beta", ln_beta), ("ln_mean", ln_mean), ("ln_rstd", ln_rstd), ("ln_x", ln_x),
                    ("ln_part", ln_part), ("ln_dgamma", ln_dgamma), ("ln_dbeta", ln_dbeta)):
        setattr(d, name, t.data_ptr() if t is not None else None)
    d.ln_eps = ln_eps
    return d


def run_gemm(descs, layout, tile_cfg=-1, stages=-1):
    arr = (L.GemmDesc * len(descs))(*descs)
    L.check(L.lib().bvc_op_gemm(arr, len(descs), layout, tile_cfg, stages, stream()), "bvc_op_gemm")


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf16_randn(*shape, scale=1.0, seed=0, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(device)


PARITY_REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def log_parity(msg):
    """One line of the parity report (gpurun_out/parity_report.txt; copies are committed under profiles/)."""
    os.makedirs(os.path.dirname(PARITY_REPORT), exist_ok=True)
    with open(PARITY_REPORT, "a") as f:
        f.write(msg + "\n")
    print(msg)
