"""Golden-fixture generator  --  TEST INFRASTRUCTURE, runs in the BUILD CONTAINER only.

Imports the real implementation of the path (transformers 5.15.0
VideoMAEForPreTraining, built from a config object exactly as
pretraining/generative/pretrain_videomae.py:51-64 does, and the reference's own
pretraining/generative/mask.py), loads the oracle's deterministic weights into
it, runs forward+backward on the oracle's synthetic batches and writes small
JSON fixtures under tests/golden/.  Nothing of the reference or of transformers
is copied: the fixtures hold inputs' checksums and output numbers only.

    python oracle/make_golden.py            # writes tests/golden/*.json

It also asserts, while generating, that oracle/videomae_oracle.py agrees with
the real implementation to fp32 round-off (this is what "pins" the oracle).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import videomae_oracle as vo  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def hf_model(cfg: vo.OracleConfig, params):
    import transformers
    hc = transformers.VideoMAEConfig(
        image_size=cfg.image_size, patch_size=cfg.patch_size, num_channels=cfg.num_channels,
        num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, hidden_size=cfg.hidden_size,
        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        intermediate_size=cfg.intermediate_size, initializer_range=0.02, use_mean_pooling=True,
        decoder_num_attention_heads=cfg.decoder_num_attention_heads,
        decoder_hidden_size=cfg.decoder_hidden_size,
        decoder_num_hidden_layers=cfg.decoder_num_hidden_layers,
        decoder_intermediate_size=cfg.decoder_intermediate_size, norm_pix_loss=cfg.norm_pix_loss)
    m = transformers.VideoMAEForPreTraining(hc)
    assert list(m.state_dict().keys()) == list(params.keys()), "state-dict key order drifted"
    m.load_state_dict(params)
    m.train()
    return m, transformers.__version__


def summarize(t: torch.Tensor, n=8):
    t = t.detach().double().flatten()
    return {"l2": float(t.norm()), "mean": float(t.mean()), "head": [float(x) for x in t[:n]],
            "numel": int(t.numel())}


def hf_taps(model, pixels, mask):
    """Run HF with forward hooks that record the same activations the oracle taps."""
    taps = {}
    hooks = []

    def add(mod, name, pick=lambda o: o):
        hooks.append(mod.register_forward_hook(lambda m, i, o: taps.__setitem__(name, pick(o))))

    add(model.videomae.embeddings, "embed")
    for i, l in enumerate(model.videomae.encoder.layer):
        add(l, f"enc{i}")
    for i, l in enumerate(model.decoder.decoder_layers):
        add(l, f"dec{i}")
    hooks.append(model.decoder.register_forward_pre_hook(lambda m, a: taps.__setitem__("x_full", a[0])))
    out = model(pixels, bool_masked_pos=mask)
    for h in hooks:
        h.remove()
    taps["logits"] = out.logits
    return out, taps


def one_case(name, cfg, batch, seed, mask_ratio, wseed=0):
    params = vo.make_params(cfg, seed=wseed)
    pixels, mask = vo.synthetic_batch(cfg, batch, seed, mask_ratio)
    model, ver = hf_model(cfg, params)
    out, htaps = hf_taps(model, pixels, mask)
    out.loss.backward()
    hgrads = {k: v.grad for k, v in model.named_parameters()}

    otaps = {}
    oloss, ograds = vo.step(cfg, params, pixels, mask, taps=otaps)

    # ---- pin the oracle against the real implementation
    rel = abs(float(oloss) - float(out.loss)) / abs(float(out.loss))
    assert rel < 2e-6, (name, "loss", rel)
    for k, v in htaps.items():
        e = float((otaps[k] - v).norm() / v.norm())
        assert e < 2e-5, (name, "tap", k, e)
    worst = 0.0
    gmax = max(float(g.norm()) for g in hgrads.values())
    for k, g in hgrads.items():
        # key.bias has a mathematically zero gradient (softmax is shift invariant along
        # keys), so the error is measured against ||g|| plus a floor tied to the largest grad.
        e = float((ograds[k] - g).norm() / (g.norm() + 1e-4 * gmax))
        worst = max(worst, e)
        assert e < 5e-5, (name, "grad", k, e)
    print(f"[{name}] oracle vs transformers {ver}: loss rel {rel:.2e}, worst grad rel {worst:.2e}")

    fx = {
        "case": name, "transformers": ver, "torch": torch.__version__,
        "config": cfg.__dict__, "batch": batch, "seed": seed, "weight_seed": wseed, "mask_ratio": mask_ratio,
        "input": {"pixels": summarize(pixels), "mask_true": int(mask.sum()),
                  "visible_idx_row0": [int(i) for i in torch.nonzero(~mask[0]).flatten()[:32]]},
        "loss": float(out.loss),
        "taps": {k: summarize(v) for k, v in htaps.items()},
        "grad_l2": {k: float(g.double().norm()) for k, g in hgrads.items()},
        "grad_head": {k: [float(x) for x in hgrads[k].flatten()[:4]] for k in vo.GRAD_PROBES},
        "grad_probes": {k: float(hgrads[k].double().norm()) for k in vo.GRAD_PROBES},
    }
    with open(os.path.join(GOLD, f"videomae_{name}.json"), "w") as f:
        json.dump(fx, f, indent=1)


def mask_fixture():
    """pretraining/generative/mask.py under a fixed numpy seed (the reference never seeds numpy)."""
    sys.path.insert(0, os.path.join(REF, "pretraining", "generative"))
    import mask as refmask  # reference module: numpy only
    cases = []
    for seed, grid, ratio in [(0, (8, 14, 14), 0.9), (7, (8, 14, 14), 0.9), (3, (2, 4, 4), 0.75)]:
        np.random.seed(seed)
        gen = refmask.TubeMaskingGenerator(grid, ratio)
        rows = [gen() for _ in range(3)]
        mine_rng = np.random.RandomState(seed)
        for r in rows:
            assert np.array_equal(r, vo.tube_mask(grid, ratio, mine_rng))
        cases.append({"seed": seed, "grid": list(grid), "ratio": ratio,
                      "num_masks_per_frame": int(gen.num_masks_per_frame),
                      "total_masks": int(gen.total_masks),
                      "visible_frame0": [[int(i) for i in np.nonzero(r[:grid[1] * grid[2]] == 0)[0]] for r in rows]})
    with open(os.path.join(GOLD, "tube_mask.json"), "w") as f:
        json.dump({"source": "pretraining/generative/mask.py:3-24", "cases": cases}, f, indent=1)
    print("[mask] reference TubeMaskingGenerator == oracle.tube_mask for", len(cases), "cases")


def simclr_fixture():
    """info_nce_loss / get_special_matrix of pretraining/contrastive/pretrain_simclr.py.  The module imports torchvision
    (absent offline) at its top for the trunk only; empty in-memory stand-in modules let the import proceed so that the
    two loss functions themselves - which use only torch/numpy - are the reference's own code."""
    import types
    from oracle import simclr_oracle as so
    stubs = []
    for name in ("torchvision", "torchvision.transforms", "torchvision.models", "torchvision.io", "torchvision.datasets",
                 "torchvision.transforms.functional"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            stubs.append(name)
    sys.path.insert(0, os.path.join(REF, "pretraining", "contrastive"))
    import pretrain_simclr as ref
    for name in stubs:      # transformers probes for torchvision later; the stand-ins must not outlive this import
        del sys.modules[name]
    cases = []
    for B, p, seed in [(8, 128, 0), (32, 512, 1), (4, 64, 2)]:
        feats = so.synthetic_features(2 * B, p, seed).requires_grad_(True)
        masks = so.make_masks(B)
        assert np.array_equal(ref.get_special_matrix(2 * B), so.get_special_matrix(2 * B))
        loss = ref.info_nce_loss(0.1, masks, feats)
        loss.backward()
        mine = so.info_nce_loss(0.1, masks, feats.detach())
        assert abs(float(mine) - float(loss)) < 1e-6
        cases.append({"B": B, "p": p, "seed": seed, "temperature": 0.1, "n_pos": int(masks[0].sum()), "n_neg": int(masks[1].sum()),
                      "loss": float(loss), "grad_l2": float(feats.grad.double().norm()),
                      "grad_head": [float(x) for x in feats.grad.flatten()[:6]]})
    # BASELINE config sizes.  B = 256 / p = 2048 (512 rows) still fits the reference function's (n, n, p) broadcast product
    # (2.1 GB); 4096 pairs (8192 rows) would need 550 GB, so that case is produced by the oracle's low-memory form, which is
    # first checked here against the reference function at 512 rows.
    for B, p, seed, use_ref in [(256, 2048, 5, True), (4096, 2048, 6, False)]:
        feats = so.synthetic_features(2 * B, p, seed).requires_grad_(True)
        if use_ref:
            loss = ref.info_nce_loss(0.1, so.make_masks(B), feats)
            low = so.info_nce_loss_lowmem(0.1, B, feats.detach())
            assert abs(float(low) - float(loss)) < 2e-6 * abs(float(loss)), (float(low), float(loss))
        else:
            loss = so.info_nce_loss_lowmem(0.1, B, feats)
        loss.backward()
        cases.append({"B": B, "p": p, "seed": seed, "temperature": 0.1, "n_pos": 2 * (2 * B - 1), "n_neg": 4 * B * B - 2 * B - 2 * (2 * B - 1),
                      "loss": float(loss), "grad_l2": float(feats.grad.double().norm()),
                      "grad_head": [float(x) for x in feats.grad.flatten()[:6]],
                      "produced_by": "reference info_nce_loss" if use_ref else
                                     "oracle info_nce_loss_lowmem (the reference function's broadcast product needs 550 GB at this size)"})
        print(f"[simclr] B={B} p={p}: loss {float(loss):.6f} ({cases[-1]['produced_by']})")
    with open(os.path.join(GOLD, "simclr_info_nce.json"), "w") as f:
        json.dump({"source": "pretraining/contrastive/pretrain_simclr.py:86-91,114-128,284-292", "cases": cases}, f, indent=1)
    print("[simclr] reference info_nce_loss == oracle for", len(cases), "cases; B=8:", cases[0]["n_pos"], "pos /", cases[0]["n_neg"], "neg")


def jepa_fixture(large=False):
    """pretraining/predictive/{vision_transformer,tensors,mask}.py (need only torch/numpy): the reference's own encoder,
    predictor and train-step arithmetic with the oracle's deterministic weights, and its MaskCollator under fixed seeds."""
    import torch.nn.functional as F
    from oracle import jepa_oracle as jo
    sys.path.insert(0, os.path.join(REF, "pretraining", "predictive"))
    for m in ("mask", "tensors", "vision_transformer"):
        sys.modules.pop(m, None)
    import vision_transformer as rvit
    import tensors as rten
    import mask as rmask
    from functools import partial
    cases = []
    specs = [("tiny", jo.TINY, 3, 6, 4, 0), ("tiny_b", jo.TINY, 2, 9, 5, 1), ("vit_b", jo.VIT_B, 2, 83, 25, 0),
             ("tiny_hd24", jo.TINY_HD24, 2, 7, 5, 2)]
    if large:      # BASELINE config 4: ViT-L/16 (1024 wide, 24 layers, 16 heads; predictor heads of 24 dims), N_ctx 100, N_pred 25
        specs = [("vit_l", jo.VIT_L, 2, 100, 25, 0)]
    for name, cfg, B, n_ctx, n_pred, seed in specs:
        enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, seed)
        pred_p = jo.make_params(jo.predictor_shapes(cfg), cfg, seed + 50)
        tgt_p = jo.make_params(jo.encoder_shapes(cfg), cfg, seed + 100)
        imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, B, seed, n_ctx, n_pred)
        kw = dict(img_size=[cfg.image_size], patch_size=cfg.patch_size, num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size,
                  embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
                  norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
        enc, tgt = rvit.VisionTransformer(**kw), rvit.VisionTransformer(**kw)
        pred = rvit.vit_predictor(sequence_shape=enc.sequence_shape, embed_dim=cfg.embed_dim, predictor_embed_dim=cfg.pred_dim,
                                  depth=cfg.pred_depth, num_heads=enc.num_heads)
        assert set(enc.state_dict()) == set(enc_p) and set(pred.state_dict()) == set(pred_p)
        # the reference's PositionalEncoding3D vs the oracle's restatement
        assert torch.allclose(enc.pos_embed.data, enc_p["pos_embed"], atol=1e-6) and torch.allclose(pred.predictor_pos_embed.data, pred_p["predictor_pos_embed"], atol=1e-6)
        enc.load_state_dict(enc_p); tgt.load_state_dict(tgt_p); pred.load_state_dict(pred_p)
        with torch.no_grad():
            h = tgt(imgs)
            h = F.layer_norm(h, (h.size(-1),))
            h = rten.apply_masks(h, m_pred)
            h = rten.repeat_interleave_batch(h, B, repeat=len(m_enc))
        zc = enc(imgs, m_enc)
        z = pred(zc, m_enc, m_pred)
        loss = F.smooth_l1_loss(z, h)
        loss.backward()
        oloss, oge, ogp, oz, oh = jo.step(cfg, enc_p, pred_p, tgt_p, imgs, m_enc, m_pred)
        assert abs(float(oloss) - float(loss)) / float(loss) < 2e-6, (name, float(oloss), float(loss))
        assert float((oz - z.detach()).norm() / z.detach().norm()) < 2e-5 and float((oh - h).norm() / h.norm()) < 2e-5
        worst = 0.0
        gmax = max(float(p.grad.norm()) for p in list(enc.parameters()) + list(pred.parameters()) if p.grad is not None)
        for mod, og in ((enc, oge), (pred, ogp)):
            for k, p in mod.named_parameters():
                if p.grad is None:
                    continue
                e = float((og[k] - p.grad).norm() / (p.grad.norm() + 1e-4 * gmax))
                worst = max(worst, e)
                assert e < 5e-5, (name, k, e)
        print(f"[jepa {name}] oracle vs reference modules: loss rel {abs(float(oloss)-float(loss))/float(loss):.1e}, worst grad rel {worst:.1e}")
        first_qkv = "blocks.0.attn.qkv.weight"
        last_qkv = f"blocks.{cfg.depth - 1}.attn.qkv.weight"
        gn = dict(enc.named_parameters())
        cases.append({"case": name, "config": cfg.__dict__, "B": B, "n_ctx": n_ctx, "n_pred": n_pred, "seed": seed,
                      "loss": float(loss), "z": summarize(z), "h": summarize(h), "zc": summarize(zc),
                      "grad_first_qkv": float(gn[first_qkv].grad.double().norm()), "grad_last_qkv": float(gn[last_qkv].grad.double().norm()),
                      "enc_grad_l2": {k: float(p.grad.double().norm()) for k, p in enc.named_parameters() if p.grad is not None},
                      "pred_grad_l2": {k: float(p.grad.double().norm()) for k, p in pred.named_parameters() if p.grad is not None}})
    if large:
        with open(os.path.join(GOLD, "jepa_vit_l.json"), "w") as f:
            json.dump({"source": "pretraining/predictive/vision_transformer.py:572-576 (vit_large), tensors.py, pretrain_jepa.py:383-402",
                       "cases": cases}, f, indent=1)
        return
    # MaskCollator: block sizes are seeded by the step counter, positions by the global torch RNG
    mc_cases = []
    for gseed in (0, 5):
        torch.manual_seed(gseed)
        mc = rmask.MaskCollator(input_size=224, patch_size=16, pred_mask_scale=(0.15, 0.2), enc_mask_scale=(0.85, 1.0),
                                aspect_ratio=(0.75, 1.5), nenc=1, npred=4, allow_overlap=False, min_keep=10)
        batch = [torch.zeros(1) for _ in range(4)]
        outs = []
        for _ in range(2):
            _, me, mp = mc(batch)
            outs.append({"enc_shape": list(me[0].shape), "pred_shape": list(mp[0].shape), "enc_row0": [int(v) for v in me[0][0]],
                         "pred0_row0": [int(v) for v in mp[0][0]], "pred3_row3": [int(v) for v in mp[3][3]]})
        upd = rmask.update_masks([torch.arange(5).view(1, 5)], 224, 16, 2, 1, isencoder=False)
        mc_cases.append({"torch_seed": gseed, "steps": outs, "update_masks_offset": int(upd[0][0, 0])})
    with open(os.path.join(GOLD, "jepa.json"), "w") as f:
        json.dump({"source": "pretraining/predictive/vision_transformer.py, tensors.py, mask.py, pretrain_jepa.py:383-402",
                   "cases": cases, "mask_collator": mc_cases}, f, indent=1)


def embedding_fixture():
    """Encoder-only inference (benchmarks/compute_embeddings_videomae.py:78-96,253-264): the reference assembles
    VideoMAEForVideoClassification(num_labels=0) from a pre-training model's embeddings + encoder and reads `.logits`."""
    import transformers
    cases = {}
    for name, cfg, batch, seed in (("tiny", vo.TINY, 3, 5), ("base", vo.BASE, 1, 6)):
        params = vo.make_params(cfg, seed=2)
        source, ver = hf_model(cfg, params)
        tc = transformers.VideoMAEConfig(
            image_size=cfg.image_size, patch_size=cfg.patch_size, num_channels=cfg.num_channels, num_frames=cfg.num_frames,
            tubelet_size=cfg.tubelet_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
            num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size, num_labels=0)
        target = transformers.VideoMAEForVideoClassification(config=tc)
        target.videomae.embeddings.load_state_dict(source.videomae.embeddings.state_dict())     # adapt_videomae, :59-66
        target.videomae.encoder.load_state_dict(source.videomae.encoder.state_dict())
        g = torch.Generator().manual_seed(77)
        fw = 1 + 0.1 * torch.randn(cfg.hidden_size, generator=g)
        fb = 0.05 * torch.randn(cfg.hidden_size, generator=g)
        with torch.no_grad():
            target.fc_norm.weight.copy_(fw)
            target.fc_norm.bias.copy_(fb)
        target.eval()
        pixels, _ = vo.synthetic_batch(cfg, batch, seed, 0.9)
        with torch.no_grad():
            out = target(pixel_values=pixels, output_hidden_states=False)
            ref = out.logits
            pooled, tokens = vo.encode(cfg, params, pixels, fw, fb, float(target.fc_norm.eps))
        e = float((pooled - ref).norm() / ref.norm())
        assert e < 2e-5, (name, "embedding", e)
        print(f"[embed {name}] oracle vs transformers {ver}: rel {e:.2e}")
        cases[name] = {"batch": batch, "seed": seed, "weight_seed": 2, "fc_norm_seed": 77, "fc_norm_eps": float(target.fc_norm.eps),
                       "pixels": summarize(pixels), "embedding": summarize(ref, n=16), "tokens": summarize(tokens),
                       "embedding_row0": [float(x) for x in ref[0, :64]]}
    with open(os.path.join(GOLD, "videomae_embedding.json"), "w") as f:
        json.dump({"transformers": ver, "torch": torch.__version__, "cases": cases}, f, indent=1)


def bf16_policy_fixture():
    """Numbers of the bf16-OPERAND oracle (oracle/videomae_oracle_bf16.py) at the cases the GPU tests run, and its pin.

    (1) per case: loss and the three grad_logger norms of the fp32 step, of the step under the build's operand policy (BUILD) and
        under autocast proper (BUILD + linear outputs and weight gradients rounded);
    (2) the pin: transformers' own VideoMAEForPreTraining run under `torch.autocast("cpu", dtype=torch.bfloat16)` on the same
        weights and clips.  CPU autocast casts the linear layers and SDPA to bf16 (operands AND outputs; GELU then runs on bf16
        values) and keeps LayerNorm / the loss in f32 - the closest executable relative of the derived CUDA policy (SURVEY.md
        section 8a).  It is not the same policy as BUILD (outputs rounded, bf16 residual updates), so the check is on the SIGNED
        deviations of the probe norms from the fp32 step: same sign and same magnitude class as the oracle's autocast mode."""
    from oracle import videomae_oracle_bf16 as vb
    cases = {}
    for name, cfg, batch, seed, wseed, ratio in (("tiny_s0", vo.TINY, 2, 0, 0, 0.75), ("base_b2_s0", vo.BASE, 2, 0, 0, 0.9),
                                                 ("base_b2_s1", vo.BASE, 2, 1, 1, 0.9), ("base_b16_s0", vo.BASE, 16, 0, 0, 0.9)):
        params = vo.make_params(cfg, seed=wseed)
        pixels, mask = vo.synthetic_batch(cfg, batch, seed, ratio)
        l32, g32 = vo.step(cfg, params, pixels, mask)
        entry = {"batch": batch, "seed": seed, "weight_seed": wseed, "mask_ratio": ratio,
                 "fp32": {"loss": float(l32), "probes": vb.probe_norms(g32)}}
        off_l, off_g = vb.step(cfg, params, pixels, mask, vb.F32)
        assert abs(float(off_l) - float(l32)) < 1e-6 * float(l32)
        assert max(abs(a - b) / b for a, b in zip(vb.probe_norms(off_g), vb.probe_norms(g32))) < 1e-6, "all switches off must be the fp32 step"
        for tag, pol in (("build", vb.BUILD), ("autocast", vb.Policy(True, True, True, True, True))):
            l, g = vb.step(cfg, params, pixels, mask, pol)
            entry[tag] = {"loss": float(l), "probes": vb.probe_norms(g)}
        if batch <= 2:
            model, ver = hf_model(cfg, params)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = model(pixels, bool_masked_pos=mask)
            out.loss.float().backward()
            hg = {k: v.grad.float() for k, v in model.named_parameters()}
            entry["transformers_cpu_autocast"] = {"loss": float(out.loss), "probes": vb.probe_norms(hg), "transformers": ver}
            ref = entry["fp32"]["probes"]
            dev_hf = [(a - b) / b for a, b in zip(entry["transformers_cpu_autocast"]["probes"], ref)]
            dev_or = [(a - b) / b for a, b in zip(entry["autocast"]["probes"], ref)]
            print(f"[bf16 {name}] probe deviations from fp32: transformers under CPU autocast {['%+.2e' % d for d in dev_hf]}, "
                  f"oracle autocast mode {['%+.2e' % d for d in dev_or]}, oracle BUILD mode "
                  f"{['%+.2e' % ((a - b) / b) for a, b in zip(entry['build']['probes'], ref)]}")
            assert abs(float(out.loss) - float(l32)) < 2e-3 * float(l32)
            if cfg.hidden_size >= 768:
                for dh, do in zip(dev_hf, dev_or):     # same sign, same size class (the policies differ in the residual adds)
                    assert abs(dh - do) < 6e-4, (name, dev_hf, dev_or)
        cases[name] = entry
    with open(os.path.join(GOLD, "videomae_bf16_policy.json"), "w") as f:
        json.dump({"source": "oracle/videomae_oracle_bf16.py; pretrain_videomae.py:306-308 (autocast), loggingtools.py:98-119 (probes)",
                   "probe_keys": list(vo.GRAD_PROBES), "torch": torch.__version__, "cases": cases}, f, indent=1)


def b64_fixture():
    """VideoMAE-base at 64 clips - the smallest batch at which the launcher picks the 256-clip benchmark's kernels by itself (every
    encoder / decoder product passes gemm8's 45-GFLOP gate, the weight-gradient groups run split-K on 256 x 256 / 128 x 384 tiles):
    transformers' own fp32 step (loss, the three grad_logger norms, the L2 norm of each of the 264 gradient tensors) and, next to it,
    the probe norms of the oracle's bf16-OPERAND step under the build's policy (the reference for the build's OWN share of a
    deviation).  One model at a time: each run holds 20-35 GB of activations in this 62 GB container."""
    import gc
    from oracle import videomae_oracle_bf16 as vb
    name, cfg, batch, seed, wseed, ratio = "base_b64_s0", vo.BASE, 64, 0, 0, 0.9
    params = vo.make_params(cfg, seed=wseed)
    pixels, mask = vo.synthetic_batch(cfg, batch, seed, ratio)
    model, ver = hf_model(cfg, params)
    out = model(pixels, bool_masked_pos=mask)
    out.loss.backward()
    hgrads = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
    loss = float(out.loss)
    del out, model
    gc.collect()
    fx = {
        "case": name, "transformers": ver, "torch": torch.__version__,
        "config": cfg.__dict__, "batch": batch, "seed": seed, "weight_seed": wseed, "mask_ratio": ratio,
        "input": {"pixels": summarize(pixels), "mask_true": int(mask.sum())},
        "loss": loss,
        "grad_l2": {k: float(g.double().norm()) for k, g in hgrads.items()},
        "grad_probes": {k: float(hgrads[k].double().norm()) for k in vo.GRAD_PROBES},
    }
    with open(os.path.join(GOLD, f"videomae_{name}.json"), "w") as f:
        json.dump(fx, f, indent=1)
    print(f"[{name}] transformers {ver}: loss {loss:.7f}, probes {[fx['grad_probes'][k] for k in vo.GRAD_PROBES]}")
    fp32_probes = vb.probe_norms(hgrads)
    del hgrads
    gc.collect()
    l, g = vb.step(cfg, params, pixels, mask, vb.BUILD)
    entry = {"batch": batch, "seed": seed, "weight_seed": wseed, "mask_ratio": ratio,
             "fp32": {"loss": loss, "probes": fp32_probes, "source": "transformers fp32 step (the oracle's equals it to 7e-7 at 2 / 16 clips)"},
             "build": {"loss": float(l), "probes": vb.probe_norms(g)}}
    print(f"[bf16 {name}] BUILD-policy probe deviations from fp32: "
          f"{['%+.2e' % ((a - b) / b) for a, b in zip(entry['build']['probes'], fp32_probes)]}")
    path = os.path.join(GOLD, "videomae_bf16_policy.json")
    with open(path) as f:
        doc = json.load(f)
    doc["cases"][name] = entry
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if "--bf16-policy" in sys.argv:
        bf16_policy_fixture()
        return
    if "--b64" in sys.argv:            # the 64-clip whole-step pin (a few minutes of CPU, ~35 GB)
        b64_fixture()
        return
    if "--full-size" in sys.argv:      # the BASELINE-size pins only (minutes of CPU): VideoMAE-base B=16, JEPA ViT-L, SimCLR 512 / 8192 rows
        one_case("base_b16_s0", vo.BASE, batch=16, seed=0, mask_ratio=0.9)
        jepa_fixture(large=True)
        simclr_fixture()
        return
    mask_fixture()
    simclr_fixture()
    jepa_fixture()
    embedding_fixture()
    if "--only-new" in sys.argv:
        return
    one_case("tiny_s0", vo.TINY, batch=2, seed=0, mask_ratio=0.75)
    one_case("tiny_s1", vo.TINY, batch=3, seed=1, mask_ratio=0.75, wseed=1)
    one_case("base_b2_s0", vo.BASE, batch=2, seed=0, mask_ratio=0.9)
    one_case("base_b2_s1", vo.BASE, batch=2, seed=1, mask_ratio=0.9, wseed=1)
    one_case("base_b16_s0", vo.BASE, batch=16, seed=0, mask_ratio=0.9)
    jepa_fixture(large=True)


if __name__ == "__main__":
    main()
