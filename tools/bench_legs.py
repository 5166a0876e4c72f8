"""The two other BASELINE configurations as functions, for bench.py's `extra` block (one GPU, after the headline loop) and for
tools/bench_jepa.py / bench_simclr.py:
  jepa_leg   - config 4: I-JEPA training step (target encoder on all 392 tokens, context encoder, predictor, smooth-L1, backward,
               fused SGD-Nesterov, EMA; pretraining/predictive/pretrain_jepa.py:383-433), synthetic 2-frame 224^2 inputs;
  simclr_leg - config 5 on one GPU: ViT-B trunk + token mean + projection head + InfoNCE over the local rows (N > 1 gathers them
               first; pretraining/contrastive/pretrain_simclr.py:320-329), backward, fused SGD-Nesterov.
Each returns a dict with throughput, ms per step and TFLOP/s against the algorithmic FLOPs of SURVEY.md section 8d."""
import copy
import time

import torch

JEPA_GFLOP = {"vit_base": 160.6, "vit_large": 473.2}     # per sample at N_ctx = 100, N_pred = 25
SIMCLR_GFLOP = 104.8                                      # per image, ViT-B/16 at 224^2


def _timed(step, warmup, steps):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, float(loss.detach())


def jepa_leg(bvc, dev, model="vit_large", batch=16, nctx=100, npred=25, warmup=6, steps=10, overlap_target=True):
    torch.manual_seed(0)
    enc, pred = bvc.jepa.get_model(dev, patch_size=16, tubelet_size=1, num_frames=2, model_name=model, image_size=224)
    tgt = copy.deepcopy(enc).to(dev)
    for p in tgt.parameters():
        p.requires_grad = False
    for m in (enc, pred, tgt):
        m._ensure_flat(dev)
    # the reference's optimiser (pretraining/predictive/helper.py:108-165 via pretrain_jepa.py:274): four groups, weight decay 1e-6 on the
    # weights only, SGD-Nesterov, GradScaler
    opt, scaler, _sched, _wd_sched = bvc.jepa.init_opt(enc, pred, iterations_per_epoch=1000, start_lr=0.1, ref_lr=0.1, momentum=0.9, warmup=0,
                                                       num_epochs=1, wd=1e-6, use_bfloat16=True)
    B = batch
    g = torch.Generator().manual_seed(1)
    imgs = ((torch.randint(0, 256, (B, 2, 3, 224, 224), generator=g, dtype=torch.uint8).float() / 255 - 0.5) / 0.25).to(dev)
    me = [torch.stack([torch.sort(torch.randperm(196, generator=g)[:nctx]).values for _ in range(B)]).to(dev)]
    mp = [(torch.stack([torch.sort(torch.randperm(196, generator=g)[:npred]).values for _ in range(B)]) + 196).to(dev) for _ in range(4)]

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if not overlap_target:
                with torch.no_grad():
                    h = bvc.jepa.select_targets(tgt(imgs), mp)
                z = pred(enc(imgs, me), me, mp)
            else:
                # the target encoder's forward has no consumer before the loss: on a second stream it runs beside the context encoder
                # and the predictor (bvc.jepa.forward_target_async: same kernels, same arithmetic)
                join = bvc.jepa.forward_target_async(tgt, imgs, mp)
                z = pred(enc(imgs, me), me, mp)
                h = join()
            loss = bvc.AllReduce.apply(bvc.jepa.smooth_l1_loss(z, h))
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        bvc.jepa.ema_update(enc, tgt, 0.996)
        return loss

    dt, loss = _timed(step, warmup, steps)
    gf = JEPA_GFLOP.get(model, 0.0)
    return {"workload": f"I-JEPA {model}/16, 2x224^2 inputs, {B} samples/GPU, N_ctx {nctx}, N_pred {npred} x 4, full step (target + context "
                        "encoders, predictor, smooth-L1, bwd, SGD-Nesterov, EMA)" + ("; target forward on a second stream" if overlap_target else ""),
            "value": round(B / dt, 1), "unit": "samples/s", "ms_per_step": round(1e3 * dt, 3), "steps": steps,
            "tflops": round(gf * B / dt / 1e3, 1), "frac_of_mfma_peak": round(gf * B / dt / 1e3 / 2500.0, 4), "final_loss": round(loss, 5)}


def simclr_leg(bvc, dev, images=512, warmup=3, steps=10):
    torch.manual_seed(0)
    model = bvc.simclr.SimCLRViT("vit_base", image_size=224).to(dev).train()
    model.trunk._ensure_flat(dev)
    opt = bvc.optim.SGD([{"params": [p for p in model.trunk.parameters() if p.requires_grad]},
                         {"params": list(model.fc.parameters())}], lr=0.1, momentum=0.9, nesterov=True)
    scaler = torch.amp.GradScaler("cuda")
    n = images
    masks = bvc.simclr.make_masks(n // 2, dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8).to(dev)     # uint8 frames, normalised on the GPU

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = bvc.AllReduce.apply(bvc.simclr.global_info_nce_loss(0.1, masks, model(x)))
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        return loss

    dt, loss = _timed(step, warmup, steps)
    return {"workload": f"SimCLR ViT-B/16 224^2, {n} images/GPU (global batch 4096 on 8 GPUs), full step (trunk, token mean, head, InfoNCE, "
                        "bwd, SGD-Nesterov)",
            "value": round(n / dt, 1), "unit": "images/s", "ms_per_step": round(1e3 * dt, 3), "steps": steps,
            "tflops": round(SIMCLR_GFLOP * n / dt / 1e3, 1), "frac_of_mfma_peak": round(SIMCLR_GFLOP * n / dt / 1e3 / 2500.0, 4),
            "final_loss": round(loss, 4)}
