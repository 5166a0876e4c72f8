"""The bf16-OPERAND mode of the VideoMAE oracle (oracle/videomae_oracle_bf16.py), on CPU.

  * every switch off == the fp32 oracle (same loss, same gradients to f32 round-off): the mode changes roundings, nothing else;
  * one linear layer in "autocast" mode == `torch.autocast("cpu", dtype=torch.bfloat16)` around F.linear - forward bits, dX and dW:
    the op-level pin of the operand policy, where the oracle's policy and an executable autocast coincide;
  * the committed fixture tests/golden/videomae_bf16_policy.json (written by oracle/make_golden.py --bf16-policy, which also runs
    transformers' VideoMAEForPreTraining under CPU autocast) is reproduced at the tiny size, and its base-size entries say what
    the parity report relies on: under bf16 operands the three grad_logger norms deviate from the fp32 step SYSTEMATICALLY
    (grad-EFL high, grad-ELL low, grad-DLL high), in transformers' own autocast run as in the oracle's modes.
"""
import json
import os

import torch
import torch.nn.functional as F

from oracle import videomae_oracle as vo
from oracle import videomae_oracle_bf16 as vb


def test_all_switches_off_is_the_fp32_step():
    cfg = vo.TINY
    params = vo.make_params(cfg, seed=3)
    pixels, mask = vo.synthetic_batch(cfg, 2, 4, 0.75)
    l0, g0 = vo.step(cfg, params, pixels, mask, grad_scale=8.0)
    l1, g1 = vb.step(cfg, params, pixels, mask, vb.F32, grad_scale=8.0)
    assert abs(float(l0) - float(l1)) < 1e-6 * abs(float(l0))
    gmax = max(float(g.norm()) for g in g0.values())
    for k in g0:
        assert float((g0[k] - g1[k]).norm()) < 2e-5 * (float(g0[k].norm()) + 1e-3 * gmax), k


def test_linear_matches_cpu_autocast():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(48, 96, generator=g)
    w = torch.randn(64, 96, generator=g) * 0.1
    b = torch.randn(64, generator=g) * 0.1
    dy = torch.randn(48, 64, generator=g)
    xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, b))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = F.linear(xa, wa, ba)
    assert ya.dtype == torch.bfloat16
    ya.backward(dy.to(torch.bfloat16))
    pol = vb.Policy(True, True, True, True, True)
    # (autocast also casts the BIAS to bf16, which neither the oracle's modes nor the build do: hand the oracle the rounded bias)
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, vb._r(b)))
    yo = vb.linear(xo, wo, bo, pol)
    yo.backward(dy)
    assert torch.equal(ya.float(), yo)                      # forward: bit for bit
    for a, o in ((xa.grad, xo.grad), (wa.grad, wo.grad), (ba.grad, bo.grad)):
        assert float((a.float() - o).norm() / o.norm()) < 4e-3          # bf16 outputs on both sides


def test_fixture_is_reproduced_and_shows_the_systematic_pattern(golden_dir):
    with open(os.path.join(golden_dir, "videomae_bf16_policy.json")) as f:
        fx = json.load(f)
    assert fx["probe_keys"] == list(vo.GRAD_PROBES)
    c = fx["cases"]["tiny_s0"]
    cfg = vo.TINY
    params = vo.make_params(cfg, seed=c["weight_seed"])
    pixels, mask = vo.synthetic_batch(cfg, c["batch"], c["seed"], c["mask_ratio"])
    loss, grads = vb.step(cfg, params, pixels, mask, vb.BUILD)
    assert abs(float(loss) - c["build"]["loss"]) < 1e-5 * c["build"]["loss"]
    for a, b in zip(vb.probe_norms(grads), c["build"]["probes"]):
        assert abs(a - b) < 2e-4 * b          # bf16 roundings may flip with the host's thread count; the norms move by ~1e-5
    for name in ("base_b2_s0", "base_b2_s1", "base_b16_s0"):
        e = fx["cases"][name]
        ref = e["fp32"]["probes"]
        for mode in ("build", "autocast") + (("transformers_cpu_autocast",) if "transformers_cpu_autocast" in e else ()):
            dev = [(a - b) / b for a, b in zip(e[mode]["probes"], ref)]
            assert dev[0] > 5e-5 and dev[1] < -2e-4 and dev[2] > 2e-5, (name, mode, dev)       # EFL high, ELL low, DLL high
            assert max(abs(d) for d in dev) < 1e-3, (name, mode, dev)
            assert abs(e[mode]["loss"] - e["fp32"]["loss"]) < 2e-3 * e["fp32"]["loss"]
