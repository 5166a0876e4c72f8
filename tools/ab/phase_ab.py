"""Experiments build: every other workgroup of the persistent GEMM (csrc/gemm8.hip) starts n x 3.4 us late (BVC_GEMM_DEBUG = 1024 + (n << 12)).
The workgroups of a launch walk equal units in step, so their epilogues (stores, side inputs) reach the memory system in bursts; a start
offset persists and interleaves one half's epilogues with the other half's K loops.  Same-process, interleaved rounds, median [min-max] us."""
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
    want = os.environ.get("BVC_ONLY", "dec fc1,dec dX fc2,dec qkv,dec fc2,dec dX fc1,enc fc1,enc qkv,enc dX fc2,enc fc2").split(",")
    shifts = [int(x) for x in os.environ.get("BVC_SHIFTS", "0,1,2,3,4,6").split(",")]
    rounds = int(os.environ.get("BVC_ROUNDS", "5"))
    print(f"tools/ab/phase_ab.py at BVC_BATCH={Bc}, {rounds} interleaved rounds; columns: start offset of odd workgroups in units of 3.4 us")
    for name, lay, M, N, K, epi in cases_for(Bc):
        if name not in want:
            continue
        tile = 11 if N == 384 else 10
        d, C, C2 = build(name, lay, M, N, K, epi)
        ts = {n: [] for n in shifts}
        for _ in range(rounds):
            for n in shifts:
                os.environ["BVC_GEMM_DEBUG"] = str(1024 + (n << 12)) if n else "0"
                G.run_gemm([d], lay, tile)
                ts[n].append(time_once(lambda: G.run_gemm([d], lay, tile), 5))
        os.environ["BVC_GEMM_DEBUG"] = "0"
        print(f"{name:12s} {epi:5s} tile{tile} " + " | ".join(f"{n}: {statistics.median(ts[n]):7.1f} [{min(ts[n]):6.1f}-{max(ts[n]):6.1f}]" for n in shifts), flush=True)


if __name__ == "__main__":
    main()
