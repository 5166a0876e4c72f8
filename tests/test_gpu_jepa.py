"""HIP JEPA path (encoder with index masks, target encoder, predictor with head_dim 32, target selection, smooth-L1,
EMA) against the oracle and the fixture from the reference's own modules.  Tolerances as for VideoMAE: loss 1e-3,
activations 2e-2, per-tensor gradients 5e-2 relative L2 (bf16 MFMA operands, f32 accumulation)."""
import copy
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from tests import gpu_util as G   # noqa: E402
from oracle import jepa_oracle as jo   # noqa: E402
from oracle import videomae_oracle_bf16 as vb   # noqa: E402

bvc = G.bvc
dev = torch.device("cuda:0")
OWN_BAR = 5e-4      # |probe norm - bf16-operand oracle's| / norm at ViT-B / ViT-L size: the build's own share of a deviation


def _modules(cfg, enc_p, pred_p, tgt_p):
    kw = dict(img_size=[cfg.image_size], patch_size=cfg.patch_size, num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size,
              embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio)
    enc = bvc.jepa.VisionTransformer(**kw)
    enc.load_state_dict(enc_p)
    tgt = copy.deepcopy(enc)            # pretrain_jepa.py:258
    tgt.load_state_dict(tgt_p)
    pred = bvc.jepa.vit_predictor(sequence_shape=enc.sequence_shape, embed_dim=cfg.embed_dim, predictor_embed_dim=cfg.pred_dim,
                                  depth=cfg.pred_depth, num_heads=enc.num_heads)
    pred.load_state_dict(pred_p)
    for p in tgt.parameters():
        p.requires_grad = False
    return enc.to(dev), pred.to(dev), tgt.to(dev)


# 3 = predictor heads of 24 dims (ViT-L's shape) at toy size, run zero-padded to 32; 4 = BASELINE config 4 itself: ViT-L/16
# (1024 wide, 24 layers, 16 heads, predictor heads of 24 dims), B=2, N_ctx 100, N_pred 25 (tests/golden/jepa_vit_l.json)
@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4])
def test_train_step_matches_oracle_and_fixture(golden_dir, idx):
    _train_step_case(golden_dir, idx)


@pytest.mark.parametrize("idx", [2, 4])
def test_train_step_with_layernorm_inside_the_predictor_products(golden_dir, idx):
    """Round 5: the predictor is 384 wide, so from 128 row units upward its LayerNorms run inside the epilogues of proj / fc2 (forward)
    and of the dX products of fc1 / qkv (backward) - gemm8.hip EC 4 / 5, Block.forward of vision_transformer.py:225-231.  The fixture
    batches are far below that size, so the schedule is forced (bvc_set_option("row_ln", 1)) and the same oracle / reference-fixture
    comparison runs on it: ViT-B (heads of 32 dims) and ViT-L (16 heads of 24 dims, run zero-padded to 32: K = 1536 / 512 products)."""
    L = bvc._lib
    old = L.set_option("row_ln", 1)
    try:
        assert L.lib().bvc_op_row_ln_selected(1000, 384, 1536, 12) == 1 and L.lib().bvc_op_row_ln_selected(1000, 768, 3072, 12) == 0
        _train_step_case(golden_dir, idx, tag_suffix=" row_ln")
    finally:
        L.set_option("row_ln", old)


def _train_step_case(golden_dir, idx, tag_suffix=""):
    cases = json.load(open(os.path.join(golden_dir, "jepa.json")))["cases"] + \
        json.load(open(os.path.join(golden_dir, "jepa_vit_l.json")))["cases"]
    c = cases[idx]
    cfg = jo.JepaConfig(**c["config"])
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, c["seed"])
    pred_p = jo.make_params(jo.predictor_shapes(cfg), cfg, c["seed"] + 50)
    tgt_p = jo.make_params(jo.encoder_shapes(cfg), cfg, c["seed"] + 100)
    imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, c["B"], c["seed"], c["n_ctx"], c["n_pred"])
    scale = 1024.0
    rloss, rge, rgp, rz, rh = jo.step(cfg, enc_p, pred_p, tgt_p, imgs, m_enc, m_pred, grad_scale=scale)
    # the same step with bf16 OPERANDS (oracle/videomae_oracle_bf16.py's policy of the build): what any bf16-operand run shows
    bloss, bge, _bgp, _bz, _bh = jo.step(cfg, enc_p, pred_p, tgt_p, imgs, m_enc, m_pred, grad_scale=scale, pol=vb.BUILD)
    enc, pred, tgt = _modules(cfg, enc_p, pred_p, tgt_p)
    x = imgs.to(dev)
    me, mp = [m.to(dev) for m in m_enc], [m.to(dev) for m in m_pred]
    # train_step of pretrain_jepa.py:383-418 with the fused target selection / loss
    with torch.no_grad():
        h = bvc.jepa.select_targets(tgt(x), mp)
    zc = enc(x, me)
    z = pred(zc, me, mp)
    loss = bvc.jepa.smooth_l1_loss(z, h)
    loss = bvc.AllReduce.apply(loss)
    (loss * scale).backward()
    torch.cuda.synchronize()
    tag = f"jepa {c['case']}{tag_suffix}"
    eh, ez = G.rel_err(h.cpu(), rh), G.rel_err(z.detach().cpu(), rz)
    rel = abs(float(loss) - float(rloss)) / float(rloss)
    G.log_parity(f"[{tag}] loss hip {float(loss):.7f} oracle {float(rloss):.7f} rel {rel:.2e}; vs the reference modules' fixture rel "
                 f"{abs(float(loss) - c['loss']) / c['loss']:.2e}; targets h rel {eh:.2e}, predictions z rel {ez:.2e}")
    assert eh < 2e-2 and ez < 2e-2
    assert rel < 1e-3, (float(loss), float(rloss))
    assert abs(float(loss) - c["loss"]) / c["loss"] < 1e-3          # the number the reference's own modules produced
    gmax = max(float(g.norm()) for g in list(rge.values()) + list(rgp.values()))
    worst = ("", 0.0)
    for mod, ref in ((enc, rge), (pred, rgp)):
        for k, p in mod.named_parameters():
            if not p.requires_grad:
                assert p.grad is None
                continue
            e = float((p.grad.float().cpu() - ref[k]).norm() / (ref[k].norm() + 1e-3 * gmax))
            if e > worst[1]:
                worst = (k, e)
            assert e < 5e-2, (k, e)
    G.log_parity(f"[{tag}] worst per-tensor gradient rel L2 {worst[1]:.2e} ({worst[0]})")
    # the predictive entry point's grad_logger probes (pretraining/predictive/loggingtools.py:98-112): first / last qkv weight norms.
    # Against the fp32 step they sit 1.6e-3 ... 3.2e-3 LOW in every case - and so does the oracle's own bf16-operand step (the
    # reference under CUDA autocast, pretrain_jepa.py:404-405, computes with bf16 operands too): the deviation belongs to the
    # operand type, not to this build.  Held here: |hip - bf16-operand oracle| < OWN_BAR (the build's own share; 1e-3 on the toy
    # configurations, whose norms average the rounding of ~100x fewer elements), and |hip - fp32| < 5e-3 with the measured value,
    # the bf16-operand oracle's and the margin printed.
    wide = cfg.embed_dim >= 768
    own_bar = OWN_BAR if wide else 1e-3
    for k in ("blocks.0.attn.qkv.weight", f"blocks.{cfg.depth - 1}.attn.qkv.weight"):
        gn, rn, bn = float(dict(enc.named_parameters())[k].grad.norm()), float(rge[k].norm()), float(bge[k].norm())
        fx = c["grad_first_qkv" if k.startswith("blocks.0.") else "grad_last_qkv"] * scale
        e, eb, own = (gn - rn) / rn, (bn - rn) / rn, (gn - bn) / bn
        G.log_parity(f"[{tag}] grad-norm {k}: hip vs fp32 {e:+.2e} (bar 5e-3) | bf16-operand oracle vs fp32 {eb:+.2e} | hip vs bf16-operand oracle "
                     f"{own:+.2e} (own bar {own_bar:.0e}, margin {own_bar / max(abs(own), 1e-12):.1f}x); hip vs the reference modules' fixture {(gn - fx) / fx:+.2e}")
        assert abs(e) < 5e-3, (k, e)
        assert abs(own) < own_bar, (k, own)
    G.log_parity(f"[{tag}] bf16-operand oracle: loss rel {(float(bloss) - float(rloss)) / float(rloss):+.2e} vs fp32; hip vs it {(float(loss) - float(bloss)) / float(bloss):+.2e}")


def test_reference_formulation_of_targets_and_loss():
    """The reference's own Python for the target path (F.layer_norm + apply_masks + repeat_interleave_batch) and
    F.smooth_l1_loss give the same numbers as the fused kernels."""
    cfg = jo.TINY
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, 3)
    imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, 4, 3, 7, 5)
    enc, _, _ = _modules(cfg, enc_p, jo.make_params(jo.predictor_shapes(cfg), cfg, 4), enc_p)
    x, mp = imgs.to(dev), [m.to(dev) for m in m_pred]
    with torch.no_grad():
        full = enc(x)                                       # no masks: every token
        a = bvc.jepa.select_targets(full, mp)
        b = torch.nn.functional.layer_norm(full, (full.size(-1),))
        b = bvc.jepa.repeat_interleave_batch(bvc.jepa.apply_masks(b, mp), 4, repeat=1)
    assert G.rel_err(a, b) < 1e-5
    ref = jo.encoder_forward(cfg, enc_p, imgs)
    assert G.rel_err(full.cpu(), ref) < 2e-2
    z = torch.randn(64, 5, 128, device=dev, requires_grad=True)
    t = torch.randn(64, 5, 128, device=dev) * 2
    l1 = bvc.jepa.smooth_l1_loss(z, t)
    (l1 * 7.0).backward()
    z2 = z.detach().clone().requires_grad_(True)
    l2 = torch.nn.functional.smooth_l1_loss(z2, t)
    (l2 * 7.0).backward()
    assert abs(float(l1) - float(l2)) / float(l2) < 1e-6 and G.rel_err(z.grad, z2.grad) < 1e-6


def test_ema_update_and_training_loop():
    cfg = jo.TINY
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, 6)
    pred_p = jo.make_params(jo.predictor_shapes(cfg), cfg, 7)
    enc, pred, tgt = _modules(cfg, enc_p, pred_p, enc_p)
    imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, 4, 6, 8, 4)
    x, me, mp = imgs.to(dev), [m.to(dev) for m in m_enc], [m.to(dev) for m in m_pred]
    opt = bvc.optim.SGD([{"params": [p for p in enc.parameters() if p.requires_grad]}, {"params": list(p for p in pred.parameters() if p.requires_grad)}],
                        lr=0.05, momentum=0.9, nesterov=True, weight_decay=0.0)
    scaler = torch.amp.GradScaler("cuda")
    ref_t = {k: v.clone() for k, v in enc_p.items()}
    losses = []
    for it in range(4):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            with torch.no_grad():
                h = bvc.jepa.select_targets(tgt(x), mp)
            z = pred(enc(x, me), me, mp)
            loss = bvc.jepa.smooth_l1_loss(z, h)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        m = 0.9
        bvc.jepa.ema_update(enc, tgt, m)
        jo.ema(ref_t, {k: v.detach().cpu() for k, v in enc.state_dict().items()}, m)
        losses.append(float(loss))
    torch.cuda.synchronize()
    for k, v in tgt.state_dict().items():
        assert G.rel_err(v.cpu(), ref_t[k]) < 1e-5, k
    assert all(l == l for l in losses) and losses[-1] < losses[0]


def test_target_forward_on_a_second_stream_is_the_same_loop_bit_for_bit():
    """bvc.jepa.forward_target_async (forward_target of pretrain_jepa.py:384-392 enqueued on a second stream, joined at the loss): five
    steps with EMA feedback against the sequential loop on identically initialised modules - every loss and every final parameter of
    the encoder, the predictor and the target encoder bit-equal, i.e. the two streams are ordered where the data says they must be
    (EMA -> next target forward, target forward -> loss -> EMA)."""
    cfg = jo.TINY
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, 16)
    pred_p = jo.make_params(jo.predictor_shapes(cfg), cfg, 17)
    imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, 4, 6, 8, 4)
    x, me, mp = imgs.to(dev), [m.to(dev) for m in m_enc], [m.to(dev) for m in m_pred]

    def run(overlap):
        enc, pred, tgt = _modules(cfg, enc_p, pred_p, enc_p)
        opt = bvc.optim.SGD([{"params": [p for p in enc.parameters() if p.requires_grad]},
                             {"params": [p for p in pred.parameters() if p.requires_grad]}], lr=0.05, momentum=0.9, nesterov=True)
        scaler = bvc.amp.GradScaler("cuda")
        losses = []
        for _ in range(5):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if overlap:
                    join = bvc.jepa.forward_target_async(tgt, x, mp)
                    z = pred(enc(x, me), me, mp)
                    h = join()
                else:
                    with torch.no_grad():
                        h = bvc.jepa.select_targets(tgt(x), mp)
                    z = pred(enc(x, me), me, mp)
                loss = bvc.jepa.smooth_l1_loss(z, h)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            opt.zero_grad()
            bvc.jepa.ema_update(enc, tgt, 0.9)
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return losses, [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (enc, pred, tgt)]

    la, sa = run(False)
    lb, sb = run(True)
    for a, b in zip(la, lb):
        assert torch.equal(a, b), (float(a), float(b))
    for da, db in zip(sa, sb):
        for k in da:
            assert torch.equal(da[k], db[k]), k
    assert float(la[-1]) < float(la[0])


def test_uint8_frames_equal_normalised_f32_bitwise():
    from oracle import jepa_oracle as jo
    cfg = jo.TINY
    enc = bvc.jepa.VisionTransformer(img_size=[cfg.image_size], patch_size=cfg.patch_size, num_frames=cfg.num_frames,
                                     tubelet_size=cfg.tubelet_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
                                     num_heads=cfg.num_heads).to(dev)
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (2, cfg.num_frames, cfg.in_chans, cfg.image_size, cfg.image_size), generator=g, dtype=torch.uint8)
    with torch.no_grad():
        a = enc(((u8.float() / 255.0 - 0.5) / 0.25).to(dev))
        b = enc(u8.to(dev))
    assert torch.equal(a, b)


def test_reference_param_groups_match_torch_sgd_in_two_launches():
    """The reference's JEPA optimiser (pretraining/predictive/helper.py:108-165, called at pretrain_jepa.py:274): four parameter
    groups - encoder weights, predictor weights, encoder biases / 1-D tensors and the predictor's with weight_decay 0 - under
    SGD-Nesterov.  bvc.jepa.init_opt builds exactly those groups; the fused step must (1) equal torch.optim.SGD on the same groups
    over 4 steps to 2e-6, weight decay large enough to tell a decayed tensor from an excluded one, and (2) cost ONE launch per
    flat buffer (encoder, predictor) - not one per run of memory-adjacent parameters of a group (~9 per layer)."""
    cfg = jo.TINY
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, 16)
    pred_p = jo.make_params(jo.predictor_shapes(cfg), cfg, 17)
    enc, pred, tgt = _modules(cfg, enc_p, pred_p, enc_p)
    imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, 4, 6, 8, 4)
    x, me, mp = imgs.to(dev), [m.to(dev) for m in m_enc], [m.to(dev) for m in m_pred]
    wd = 0.05
    opt, scaler, sched, wd_sched = bvc.jepa.init_opt(enc, pred, iterations_per_epoch=10, start_lr=0.02, ref_lr=0.02, momentum=0.9, warmup=0,
                                                     num_epochs=1, wd=wd, use_bfloat16=True)
    assert sched is None and wd_sched is None and isinstance(scaler, bvc.amp.GradScaler)
    assert len(opt.param_groups) == 4
    assert [g["weight_decay"] for g in opt.param_groups] == [wd, wd, 0, 0] and all(g.get("WD_exclude") for g in opt.param_groups[2:])
    # group membership by the reference's rule: 'bias' in the name or a 1-D tensor -> the excluded group
    for mod, gw, gb in ((enc, 0, 2), (pred, 1, 3)):
        for n, p in mod.named_parameters():
            want = gb if ("bias" in n or p.dim() == 1) else gw
            assert any(p is q for q in opt.param_groups[want]["params"]), n
    # one real step to get .grad views into the flat gradient buffers
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.no_grad():
            h = bvc.jepa.select_targets(tgt(x), mp)
        loss = bvc.jepa.smooth_l1_loss(pred(enc(x, me), me, mp), h)
    loss.backward()
    # the torch reference on clones, same four groups in the same order
    mine = [list(g["params"]) for g in opt.param_groups]
    ref = [[torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad) for p in ps] for ps in mine]
    ropt = torch.optim.SGD([{"params": ref[0]}, {"params": ref[1]}, {"params": ref[2], "weight_decay": 0}, {"params": ref[3], "weight_decay": 0}],
                           lr=0.02, weight_decay=wd, momentum=0.9, nesterov=True, foreach=False)
    calls = {"seg": 0, "run": 0}
    L = bvc._lib.lib()
    seg_fn, run_fn = L.bvc_op_sgd_step_segments, L.bvc_op_sgd_step

    class _Count:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)
    L.bvc_op_sgd_step_segments, L.bvc_op_sgd_step = _Count(seg_fn, "seg"), _Count(run_fn, "run")
    try:
        gen = torch.Generator(device=dev).manual_seed(3)
        for it in range(4):
            for m in (enc, pred):
                m.flat_grads().copy_(torch.randn(m.flat_grads().shape, device=dev, generator=gen) * 0.1)
            for ps, rs in zip(mine, ref):
                for p, r in zip(ps, rs):
                    r.grad = None if p.grad is None else p.grad.detach().clone()
            opt.param_groups[0]["lr"] = opt.param_groups[2]["lr"] = 0.02 * (1 + it)      # a schedule moves the hyper-parameters per step
            ropt.param_groups[0]["lr"] = ropt.param_groups[2]["lr"] = 0.02 * (1 + it)
            opt.step()
            ropt.step()
            torch.cuda.synchronize()
            for ps, rs in zip(mine, ref):
                for p, r in zip(ps, rs):
                    torch.testing.assert_close(p.data, r.data, rtol=2e-6, atol=2e-7)
    finally:
        L.bvc_op_sgd_step_segments, L.bvc_op_sgd_step = seg_fn, run_fn
    assert calls == {"seg": 8, "run": 0}, calls        # 4 steps x (encoder buffer + predictor buffer)
    # the frozen positional tables sit inside the flat buffers and belong to no group: untouched
    assert torch.equal(enc.state_dict()["pos_embed"].cpu(), enc_p["pos_embed"])
    sd = opt.state_dict()
    assert all(set(st) == {"momentum_buffer"} for st in sd["state"].values())
    # and the whole reference loop body with the scaler: steps are taken (not skipped), the loss falls
    opt.zero_grad()
    losses = []
    for it in range(4):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            with torch.no_grad():
                h = bvc.jepa.select_targets(tgt(x), mp)
            loss = bvc.jepa.smooth_l1_loss(pred(enc(x, me), me, mp), h)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        losses.append(float(loss))
    assert scaler.get_scale() == 65536.0 and losses[-1] < losses[0], (scaler.get_scale(), losses)


def test_fused_step_skips_a_parameter_frozen_after_the_first_step():
    """torch.optim skips a parameter whose .grad is None.  The fused optimiser steps a flat module through a segment table built on the
    first step; a parameter that loses its gradient LATER (frozen for a curriculum stage: requires_grad_(False), .grad = None) must
    drop out of its segment - the flat gradient buffer still holds its last gradient."""
    cfg = jo.TINY
    enc, pred, _tgt = _modules(cfg, jo.make_params(jo.encoder_shapes(cfg), cfg, 21), jo.make_params(jo.predictor_shapes(cfg), cfg, 22),
                               jo.make_params(jo.encoder_shapes(cfg), cfg, 21))
    imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, 2, 6, 8, 4)
    x, me, mp = imgs.to(dev), [m.to(dev) for m in m_enc], [m.to(dev) for m in m_pred]
    opt = bvc.optim.SGD([{"params": list(enc.parameters())}, {"params": list(pred.parameters())}], lr=0.1, momentum=0.9, nesterov=True)

    def backward():
        z = pred(enc(x, me), me, mp)
        (z.float() ** 2).mean().backward()
    backward()
    opt.step()
    frozen = dict(enc.named_parameters())["blocks.0.mlp.fc1.weight"]
    other = dict(enc.named_parameters())["blocks.0.mlp.fc2.weight"]
    opt.zero_grad()
    backward()
    frozen.grad = None
    before_f, before_o = frozen.detach().clone(), other.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(frozen.detach(), before_f)
    assert not torch.equal(other.detach(), before_o)
