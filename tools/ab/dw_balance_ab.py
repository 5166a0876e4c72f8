"""Experiments build: the balanced walk of a split-K weight-gradient group that does not fill the chip (csrc/gemm8.hip plan_balance) against
the plain split walk (BVC_G8_NO_BALANCE=1), same process, interleaved rounds: the encoder layer's four gradients on 256 x 256 tiles,
split 2 (108 tiles x 2 = 216 units on 256 CUs), at BVC_BATCH clips.  Outputs are compared first (f32 atomics: three partial sums per
element instead of two, so equal up to the order of additions).  BVC_G8_BALANCE_EPI / _TAIL are the two constants of the plan."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_util as G  # noqa: E402

dev = "cuda"
rounds = int(os.environ.get("BVC_ROUNDS", "7"))


def time_once(fn, iters=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def setenv(cfg):
    for k in ("BVC_G8_NO_BALANCE", "BVC_G8_BALANCE_EPI", "BVC_G8_BALANCE_TAIL"):
        os.environ.pop(k, None)
    for k, v in cfg.items():
        os.environ[k] = v


def main():
    print(f"tools/ab/dw_balance_ab.py, {rounds} interleaved rounds, median [min-max] us")
    variants = [("plain", {"BVC_G8_NO_BALANCE": "1"}), ("balanced", {})]
    for e in os.environ.get("BVC_EPIS", "0,12").split(","):
        variants.append((f"e={e}", {"BVC_G8_BALANCE_EPI": e}))
    for c in os.environ.get("BVC_TAILS", "1.15,1.3").split(","):
        variants.append((f"c={c}", {"BVC_G8_BALANCE_TAIL": c}))
    for Bc in [int(x) for x in os.environ.get("BVC_BATCHES", "256,64").split(",")]:
        for tag, M, D, I, tile, split in (("enc ViT-B", Bc * 160, 768, 3072, 10, 2), ("enc ViT-L 196 tok", Bc * 196, 1024, 4096, 10, 1),
                                          ("enc ViT-B tile11", Bc * 160, 768, 3072, 11, 1)):
            if split == 1:
                continue    # (listed for the record: unsplit groups are not balanced)
            dy, act = G.bf16_randn(M, D, seed=7), G.bf16_randn(M, I, seed=8)
            dh, ln2 = G.bf16_randn(M, I, seed=9), G.bf16_randn(M, D, seed=10)
            dqkv = G.bf16_randn(M, 3 * D, seed=11)
            shapes = [(D, I), (I, D), (D, D), (3 * D, D)]

            def fresh():
                return [torch.zeros(s, device=dev) for s in shapes], [torch.zeros(s[0], device=dev) for s in shapes]

            def mk(outs, bs):
                return [G.gemm_desc(dy, act, D, I, M, G.EPI["F32"], outs[0], rowsum=bs[0], split_k=split),
                        G.gemm_desc(dh, ln2, I, D, M, G.EPI["F32"], outs[1], rowsum=bs[1], split_k=split),
                        G.gemm_desc(dy, ln2, D, D, M, G.EPI["F32"], outs[2], rowsum=bs[2], split_k=split),
                        G.gemm_desc(dqkv, ln2, 3 * D, D, M, G.EPI["F32"], outs[3], rowsum=bs[3], split_k=split)]
            # outputs: plain vs balanced (both accumulate into zeroed buffers)
            ref = None
            for name, cfg in variants[:2]:
                setenv(cfg)
                outs, bs = fresh()
                G.run_gemm(mk(outs, bs), G.TN, tile)
                torch.cuda.synchronize()
                if ref is None:
                    ref = (outs, bs)
                else:
                    for i in range(4):
                        dw = (outs[i] - ref[0][i]).abs().max().item() / ref[0][i].abs().max().item()
                        db = (bs[i] - ref[1][i]).abs().max().item() / ref[1][i].abs().max().item()
                        print(f"  {tag} at {Bc} clips, product {i}: balanced vs plain  max |dW diff| / max |dW| = {dw:.2e}, bias gradient {db:.2e}")
            flops = 2.0 * M * (D * I * 2 + D * D * 4)
            outs, bs = fresh()
            ds = mk(outs, bs)
            res = {n: [] for n, _ in variants}
            for _ in range(rounds):
                for n, cfg in variants:
                    setenv(cfg)
                    G.run_gemm(ds, G.TN, tile)
                    res[n].append(time_once(lambda: G.run_gemm(ds, G.TN, tile)))
            setenv({})
            for n, _ in variants:
                us = statistics.median(res[n])
                print(f"{tag} at {Bc} clips dW group tile{tile} split {split}  {n:9s}: {us:8.1f} [{min(res[n]):7.1f}-{max(res[n]):7.1f}] us  {flops / us / 1e6:7.1f} TF",
                      flush=True)


def head():
    # the decoder head's weight gradient (1536 x 384 outputs = 12 tiles of 128 x 384, K = masked tokens): K splits 16 (192 units) vs 21 (252)
    for Bc in [int(x) for x in os.environ.get("BVC_BATCHES", "256,64").split(",")]:
        M = Bc * 1408
        dy, x = G.bf16_randn(M, 1536, seed=3), G.bf16_randn(M, 384, seed=4)
        out, bias = torch.zeros(1536, 384, device=dev), torch.zeros(1536, device=dev)
        flops = 2.0 * M * 1536 * 384
        combos = [(16, {"BVC_G8_NO_BALANCE": "1"}), (16, {}), (18, {}), (20, {}), (21, {}), (21, {"BVC_G8_NO_BALANCE": "1"}), (24, {})]
        res = {i: [] for i in range(len(combos))}
        for _ in range(rounds):
            for i, (sp, cfg) in enumerate(combos):
                setenv(cfg)
                d = [G.gemm_desc(dy, x, 1536, 384, M, G.EPI["F32"], out, rowsum=bias, split_k=sp)]
                G.run_gemm(d, G.TN, 12)
                res[i].append(time_once(lambda: G.run_gemm(d, G.TN, 12)))
        setenv({})
        for i, (sp, cfg) in enumerate(combos):
            us = statistics.median(res[i])
            print(f"head dW at {Bc} clips tile12 split {sp:2d} {'plain   ' if cfg else 'balanced'}: {us:8.1f} [{min(res[i]):7.1f}-{max(res[i]):7.1f}] us  {flops / us / 1e6:7.1f} TF",
                  flush=True)


if __name__ == "__main__":
    if os.environ.get("BVC_HEAD"):
        head()
        sys.exit(0)
    main()
