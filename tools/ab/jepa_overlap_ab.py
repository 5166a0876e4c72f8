"""A/B in one process: the I-JEPA step with the target encoder's forward on a second stream (beside the context encoder and the predictor)
against the sequential step of pretrain_jepa.py:383-433.  BVC_BATCH (default 16), BVC_MODEL (vit_large), BVC_ROUNDS (3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge   # noqa: E402

bvc = ge.load_package()
from tools import bench_legs   # noqa: E402

dev = torch.device("cuda:0")
batch = int(os.environ.get("BVC_BATCH", "16"))
model = os.environ.get("BVC_MODEL", "vit_large")
steps = 10 if batch <= 64 else 5
for r in range(int(os.environ.get("BVC_ROUNDS", "3"))):
    a = bench_legs.jepa_leg(bvc, dev, model=model, batch=batch, steps=steps, warmup=int(os.environ.get('BVC_WARMUP', '6')), overlap_target=False)
    b = bench_legs.jepa_leg(bvc, dev, model=model, batch=batch, steps=steps, warmup=int(os.environ.get('BVC_WARMUP', '6')), overlap_target=True)
    print(f"{model} b{batch} round {r}: sequential {a['ms_per_step']:.3f} ms (loss {a['final_loss']}) | target on a second stream "
          f"{b['ms_per_step']:.3f} ms (loss {b['final_loss']})  {100 * (b['ms_per_step'] / a['ms_per_step'] - 1):+.1f} %", flush=True)
