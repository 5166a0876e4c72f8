"""Every product / attention call / LayerNorm of one VideoMAE-base step, launched alone at BVC_BATCH clips (the probe behind
bench.py's `roofline.kernels`, baby-vision-curriculum_amd/probe.py), one line per product: kernel instantiation, launches per step,
us per launch, algorithmic TFLOP/s and GB/s."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

bvc = ge.load_package()
B = int(os.environ.get("BVC_BATCH", "256"))
rows, total = bvc.probe.step_kernels(B, torch.device("cuda:0"))
print(f"tools/step_probe.py at BVC_BATCH={B}: probed sum {total / 1e3:.2f} ms per step")
print(f"{'kernel / product':86s} {'n':>4s} {'us':>9s} {'TF/s':>7s} {'GB/s':>7s} {'bound':>5s} {'frac':>6s}")
for r in rows:
    print(f"{r['kernel']:86s} {r['launches_per_step']:4d} {r['us_per_step']:9.1f} {r['tflops']:7.1f} {r['gb_per_s']:7.1f} {r['bound']:>5s} {r['frac']:6.3f}")
    for p in r["products"]:
        print(f"    {p['name']:82s} {p['launches']:4d} {p['launch_us']:9.1f} {p['tflops']:7.1f} {p['gb_per_s']:7.1f}")
