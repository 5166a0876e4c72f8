"""Same-process A/B of the weight-gradient side stream (bvc_set_option("dw_overlap", 1 / 0)): whole fwd+bwd steps of VideoMAE-base at a
given batch, interleaved rounds, HIP-event times.  Usage: python tools/ab/dw_overlap_ab.py [batch] [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge   # noqa: E402

bvc = ge.load_package()
from oracle import videomae_oracle as vo   # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = vo.BASE
kw = {k: v for k, v in cfg.__dict__.items() if k != "decoder_norm_eps"}
model = bvc.VideoMAEForPreTraining(bvc.VideoMAEConfig(**kw)).to(dev).train()
g = torch.Generator().manual_seed(1)
px = torch.randint(0, 256, (B, 16, 3, 224, 224), generator=g, dtype=torch.uint8).to(dev)
gen = bvc.mask.TubeMaskingGenerator((8, 14, 14), 0.9)
np.random.seed(0)
mk = torch.from_numpy(np.stack([gen() for _ in range(B)])).bool().to(dev)


def step():
    for p in model.parameters():
        p.grad = None
    out = model(px, bool_masked_pos=mk)
    out.loss.backward()
    return out.loss


def timed(n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, float(loss)


res = {1: [], 0: []}
for mode in (1, 0):
    bvc._lib.set_option("dw_overlap", mode)
    timed(2)
for r in range(ROUNDS):
    for mode in (1, 0, 0, 1):
        bvc._lib.set_option("dw_overlap", mode)
        t, loss = timed()
        res[mode].append(t)
        print(f"round {r} dw_overlap={mode:+d}: {t:8.3f} ms/step (fwd+bwd, no optimiser) loss {loss:.6f}", flush=True)
bvc._lib.set_option("dw_overlap", 0)
for mode in (1, 0):
    a = np.array(res[mode])
    print(f"dw_overlap={mode:+d}: median {np.median(a):8.3f} ms  min {a.min():8.3f}  max {a.max():8.3f}  ({B} clips)")
print(f"side stream / one stream = {np.median(res[1]) / np.median(res[0]):.4f}")
