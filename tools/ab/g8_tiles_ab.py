"""Same-process A/B of tile configurations of the persistent GEMM (csrc/gemm8.hip) on the NT / NN products of the VideoMAE-base step
at BVC_BATCH clips.  BVC_TILES="10,11,13" names the columns; the first one is listed twice ("A" and "A'", interleaved like the
others): the spread between two columns of the SAME kernel is the noise floor every A-B difference has to clear.
Per column: median and min-max over BVC_ROUNDS interleaved rounds (HIP events on the launch stream); a row whose best A-B
effect is smaller than its A-A' spread is flagged "~".  BVC_CHECK=1 also bit-compares every column's output with the first's."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"


def time_once(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def cases_for(Bc):
    Me, Md, Mm = Bc * 160, Bc * 1568, Bc * 1408
    return [
        ("enc qkv", G.NT, Me, 2304, 768, "BF16"), ("enc proj", G.NT, Me, 768, 768, "RESID"), ("enc fc1", G.NT, Me, 3072, 768, "GELU"),
        ("enc fc2", G.NT, Me, 768, 3072, "RESID"),
        ("dec qkv", G.NT, Md, 1152, 384, "BF16"), ("dec proj", G.NT, Md, 384, 384, "RESID"), ("dec fc1", G.NT, Md, 1536, 384, "GELU"),
        ("dec fc2", G.NT, Md, 384, 1536, "RESID"), ("head", G.NT, Mm, 1536, 384, "BF16"), ("patch", G.NT, Me, 768, 1536, "F32"),
        ("enc dX fc2", G.NN, Me, 3072, 768, "DGELU"), ("enc dX fc1", G.NN, Me, 768, 3072, "BF16"), ("enc dX qkv", G.NN, Me, 768, 2304, "BF16"),
        ("enc dX proj", G.NN, Me, 768, 768, "BF16"),
        ("dec dX fc2", G.NN, Md, 1536, 384, "DGELU"), ("dec dX fc1", G.NN, Md, 384, 1536, "BF16"), ("dec dX qkv", G.NN, Md, 384, 1152, "BF16"),
        ("dec dX proj", G.NN, Md, 384, 384, "BF16"),
        ("square 4096", G.NT, 4096, 4096, 4096, "BF16"), ("square 8192", G.NT, 8192, 8192, 8192, "BF16"),
    ]


def build(name, lay, M, N, K, epi):
    if lay == G.NT:
        A, B = G.bf16_randn(M, K), G.bf16_randn(N, K, seed=1)
    else:
        A, B = G.bf16_randn(M, K), G.bf16_randn(K, N, seed=1)
    f32 = epi in ("RESID", "F32")
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi == "GELU":
        kw["C2"] = torch.zeros_like(C)
    if epi == "RESID":
        kw["resid"] = torch.randn(M, N, device=dev)
    if epi == "DGELU":
        kw["aux"] = G.bf16_randn(M, N, seed=5)
    if epi != "DGELU":
        kw["bias"] = torch.randn(N, device=dev)
    d = G.gemm_desc(A, B, M, N, K, G.EPI[epi], C, **kw)
    d._keep = (A, B, kw.get("bias", torch.zeros(N, device=dev)))
    return d, C, kw.get("C2")


def main():
    Bc = int(os.environ.get("BVC_BATCH", "256"))
    tiles = [int(t) for t in os.environ.get("BVC_TILES", "10,11").split(",")]
    rounds = int(os.environ.get("BVC_ROUNDS", "7"))
    only = os.environ.get("BVC_ONLY")
    check = os.environ.get("BVC_CHECK") == "1"
    cols = [tiles[0]] + tiles              # column 0 and 1: the same kernel (A, A')
    names = ["A tile%d" % tiles[0], "A' tile%d" % tiles[0]] + ["tile%d" % t for t in tiles[1:]]
    print(f"tools/ab/g8_tiles_ab.py at BVC_BATCH={Bc}, {rounds} interleaved rounds; per column median [min-max] us; '~': best A-B effect < A-A' spread")
    for name, lay, M, N, K, epi in cases_for(Bc):
        if only and only not in name:
            continue
        d, C, C2 = build(name, lay, M, N, K, epi)
        ok = {}
        ref = None
        for t in sorted(set(cols), key=cols.index):
            try:
                C.zero_()
                G.run_gemm([d], lay, t)
                G.run_gemm([d], lay, t)
                torch.cuda.synchronize()
                ok[t] = True
                if check:
                    got = (C.clone(), C2.clone() if C2 is not None else None)
                    if epi in ("BF16", "F32"):     # against torch's f32 product of the same bf16 operands
                        Af, Bf = d._keep[0].float(), d._keep[1].float()
                        want = (Af @ (Bf.t() if lay == G.NT else Bf)) + d._keep[2].float()
                        err = float((got[0].float() - want).norm() / want.norm())
                        bad = int(((got[0].float() - want).abs() > 0.02 * want.abs() + 1.0).sum())
                        print(f"    {name}: tile{t} vs torch f32: rel {err:.2e}, elements off by more than 2 % + 1: {bad}")
                    if t == cols[0]:
                        ref = got
                    elif ref is not None:
                        same = torch.equal(got[0], ref[0]) and (got[1] is None or torch.equal(got[1], ref[1]))
                        if not same:
                            print(f"    {name}: tile{t} output differs from tile{cols[0]} (max abs {float((got[0].float() - ref[0].float()).abs().max()):.3e})")
            except Exception as e:      # a tile configuration that does not take the problem
                ok[t] = False
        times = [[] for _ in cols]
        for _ in range(rounds):
            for i, t in enumerate(cols):
                if ok[t]:
                    times[i].append(time_once(lambda: G.run_gemm([d], lay, t), 5))
        fl = 2.0 * M * N * K
        med = [statistics.median(x) if x else float("nan") for x in times]
        parts = []
        for i, nm in enumerate(names):
            if not times[i]:
                parts.append(f"{nm} -")
                continue
            parts.append(f"{nm} {med[i]:7.1f} [{min(times[i]):6.1f}-{max(times[i]):6.1f}] {fl / med[i] / 1e6:6.1f}TF")
        aa = abs(med[0] - med[1])
        base = min(med[0], med[1])
        best = min([m for m in med[2:] if m == m], default=float("nan"))
        flag = "~" if best == best and abs(base - best) < aa else " "
        print(f"{flag} {name:12s} {epi:6s} " + " | ".join(parts), flush=True)


if __name__ == "__main__":
    main()
