"""Where do the systematic deviations of the three grad_logger norms (grad-EFL / ELL / DLL) from the fp32 step come from?

CPU only.  VideoMAE-base at the fixture's inputs (batch 2, seed 0 by default): the fp32 oracle step, the bf16-operand oracle step
under the build's full policy (oracle/videomae_oracle_bf16.py BUILD) and one run per ablation - a single operand family switched
ON, and the full policy with a single family switched OFF.  Prints the signed relative deviation of every probe norm from the fp32
step: what ANY implementation with bf16 operands shows (the reference under CUDA autocast included), and which family owns it.
`--gpu-report` appends the build's own measured deviations (lines of gpurun_out/parity_report.txt) for the same case."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import videomae_oracle as vo  # noqa: E402
from oracle import videomae_oracle_bf16 as vb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--wseed", type=int, default=0)
    ap.add_argument("--ratio", type=float, default=0.9)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    cfg = vo.TINY if a.tiny else vo.BASE
    params = vo.make_params(cfg, seed=a.wseed)
    pixels, mask = vo.synthetic_batch(cfg, a.batch, a.seed, 0.75 if a.tiny else a.ratio)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    t0 = time.time()
    ref_loss, ref_grads = vo.step(cfg, params, pixels, mask)
    ref = vb.probe_norms(ref_grads)
    say(f"tools/efl_decomposition.py: VideoMAE-{'tiny' if a.tiny else 'base'}, batch {a.batch}, input seed {a.seed}, weight seed {a.wseed} "
        f"(fp32 step {time.time() - t0:.0f} s on {a.threads} threads)")
    say(f"fp32 oracle: loss {float(ref_loss):.7f}  grad-EFL {ref[0]:.6e}  grad-ELL {ref[1]:.6e}  grad-DLL {ref[2]:.6e}")
    say(f"{'operand policy':44s} {'loss rel':>10s} {'EFL':>10s} {'ELL':>10s} {'DLL':>10s}   worst per-tensor gradient rel L2")
    P = vb.Policy
    runs = [("all off (must reproduce fp32)", vb.F32),
            ("BUILD: weights+acts+grads+gelu_grad", vb.BUILD),
            ("only weights", P(True, False, False, False)),
            ("only acts", P(False, True, False, False)),
            ("only grads", P(False, False, True, False)),
            ("only gelu_grad", P(False, False, False, True)),
            ("BUILD without weights (f32 weight operands)", P(False, True, True, True)),
            ("BUILD without acts", P(True, False, True, True)),
            ("BUILD without grads", P(True, True, False, True)),
            ("BUILD without gelu_grad (f32 saved gelu')", P(True, True, True, False)),
            ("autocast proper (+ linear outputs, dW in bf16)", P(True, True, True, True, True))]
    gmax = max(float(g.norm()) for g in ref_grads.values())
    for name, pol in runs:
        loss, grads = vb.step(cfg, params, pixels, mask, pol)
        n = vb.probe_norms(grads)
        worst = max(float((grads[k] - ref_grads[k]).norm() / (ref_grads[k].norm() + 1e-3 * gmax)) for k in ref_grads)
        say(f"{name:44s} {(float(loss) - float(ref_loss)) / float(ref_loss):+10.2e} " +
            " ".join(f"{(x - r) / r:+10.2e}" for x, r in zip(n, ref)) + f"   {worst:.2e}")
    say("(signed: (norm - fp32 norm) / fp32 norm.  The bar of tests/test_gpu_videomae.py is |.| < 1e-3 against the fp32 step.)")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
