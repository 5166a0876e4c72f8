"""Experiments build only (BVC_EXTRA_HIPCC_FLAGS=-DBVC_EXPERIMENTS): what bounds a step of the ping-pong GEMM (csrc/gemm_pp.hip)?
Times tile config 14 on a few products with BVC_GEMM_DEBUG ablations (results are wrong under them, only the time counts):
64 = the epilogue row does not wait for the next K tile, 128 = no MFMAs, 256 = no LDS-DMA."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402
from tools.g8_tiles_ab import build, time_once, cases_for  # noqa: E402


def main():
    Bc = int(os.environ.get("BVC_BATCH", "256"))
    want = os.environ.get("BVC_ONLY", "dec qkv,dec fc1,enc fc1,square 4096").split(",")
    for name, lay, M, N, K, epi in cases_for(Bc):
        if name not in want:
            continue
        d, C, C2 = build(name, lay, M, N, K, epi)
        row = []
        for tile, dbg in ((10, 0), (14, 0), (14, 64), (14, 128), (14, 256), (14, 64 + 128), (14, 64 + 256), (14, 64 + 128 + 256)):
            os.environ["BVC_GEMM_DEBUG"] = str(dbg)
            for _ in range(2):
                G.run_gemm([d], lay, tile)
            torch.cuda.synchronize()
            ts = [time_once(lambda: G.run_gemm([d], lay, tile), 5) for _ in range(5)]
            row.append(f"tile{tile}/dbg{dbg}: {statistics.median(ts):7.1f}")
        os.environ["BVC_GEMM_DEBUG"] = "0"
        print(f"{name:12s} {epi:5s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
