"""Experiments build: the panel width G of the persistent GEMM's tile walk (csrc/gemm.hip pick_panel) on the step's long-K products and
on squares - BVC_GEMM_PANEL=G overrides the model per launch.  Interleaved rounds, median [min-max] us per G; 'model' = pick_panel."""
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
    want = os.environ.get("BVC_ONLY", "enc qkv,enc fc1,enc fc2,enc dX fc2,enc dX fc1,enc dX qkv,dec qkv,dec fc1,patch,square 4096,square 8192").split(",")
    gs = [0] + [int(x) for x in os.environ.get("BVC_PANELS", "1,2,3,4,6,8,12").split(",")]
    rounds = int(os.environ.get("BVC_ROUNDS", "5"))
    for name, lay, M, N, K, epi in cases_for(Bc):
        if name not in want:
            continue
        d, C, C2 = build(name, lay, M, N, K, epi)
        times = {g: [] for g in gs}
        for _ in range(rounds):
            for g in gs:
                if g:
                    os.environ["BVC_GEMM_PANEL"] = str(g)
                else:
                    os.environ.pop("BVC_GEMM_PANEL", None)
                G.run_gemm([d], lay, 10)
                times[g].append(time_once(lambda: G.run_gemm([d], lay, 10), 5))
        os.environ.pop("BVC_GEMM_PANEL", None)
        tn = (N + 255) // 256
        parts = []
        for g in gs:
            if g > tn and g != gs[1]:
                continue
            m = statistics.median(times[g])
            parts.append(f"{'model' if g == 0 else 'G=%d' % g} {m:7.1f} [{min(times[g]):6.1f}-{max(times[g]):6.1f}]")
        print(f"{name:12s} K={K:5d} tiles_n={tn:3d} " + " | ".join(parts), flush=True)


if __name__ == "__main__":
    main()
