"""A handful of launches of the weight-gradient group of a decoder / encoder layer on ONE tile config (BVC_DW_CASE = dec10 / dec12 /
enc10) for rocprofv3 --pmc passes (tools/gpu_check.sh pmc_dw): executed MFMA instructions, MFMA-pipe busy and wave-cycle shares of the
256 x 256 and the 128 x 384 tile on the same problem."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"
case = os.environ.get("BVC_DW_CASE", "dec12")
Bc = int(os.environ.get("BVC_BATCH", "256"))
tag, tile, split = {"dec10": ("dec", 10, 6), "dec12": ("dec", 12, 7), "enc10": ("enc", 10, 2)}[case]
M, D, I = (Bc * 1568, 384, 1536) if tag == "dec" else (Bc * 160, 768, 3072)
dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
dqkv = G.bf16_randn(M, 3 * D, seed=11)
outs = [torch.zeros(D, I, device=dev), torch.zeros(I, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
bs = [torch.zeros(D, device=dev), torch.zeros(I, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
ds = [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),
      G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),
      G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),
      G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]
for _ in range(4):
    G.run_gemm(ds, G.TN, tile)
torch.cuda.synchronize()
print(case, "done", flush=True)
