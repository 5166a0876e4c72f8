"""JEPA oracle + host-side mirrors against the fixture written from the reference's own predictive/ modules (CPU)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle import jepa_oracle as jo


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.load(open(os.path.join(golden_dir, "jepa.json")))


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_oracle_matches_reference_fixture(fx, idx):
    c = fx["cases"][idx]
    cfg = jo.JepaConfig(**c["config"])
    enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, c["seed"])
    pred_p = jo.make_params(jo.predictor_shapes(cfg), cfg, c["seed"] + 50)
    tgt_p = jo.make_params(jo.encoder_shapes(cfg), cfg, c["seed"] + 100)
    imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, c["B"], c["seed"], c["n_ctx"], c["n_pred"])
    loss, ge_, gp, z, h = jo.step(cfg, enc_p, pred_p, tgt_p, imgs, m_enc, m_pred)
    assert abs(float(loss) - c["loss"]) / c["loss"] < 5e-6
    for t, s in ((z, c["z"]), (h, c["h"])):
        assert abs(float(t.double().norm()) - s["l2"]) / s["l2"] < 2e-5
        np.testing.assert_allclose(t.flatten()[:8].double().numpy(), np.array(s["head"]), rtol=2e-4, atol=2e-5)
    gmax = max(list(c["enc_grad_l2"].values()) + list(c["pred_grad_l2"].values()))
    for k, n in c["enc_grad_l2"].items():
        assert abs(float(ge_[k].double().norm()) - n) <= 1e-4 * n + 1e-6 * gmax, k
    for k, n in c["pred_grad_l2"].items():
        assert abs(float(gp[k].double().norm()) - n) <= 1e-4 * n + 1e-6 * gmax, k
    # the grad_logger probes of the predictive entry point: first / last qkv weight (loggingtools.py:98-112)
    assert abs(float(ge_["blocks.0.attn.qkv.weight"].double().norm()) - c["grad_first_qkv"]) / c["grad_first_qkv"] < 5e-5


def test_known_parameter_counts():
    # ViT-B (2 frames) 85,947,648 parameters of which pos_embed (301,056) is frozen; predictor 11,389,440
    e = jo.encoder_shapes(jo.VIT_B)
    assert sum(int(np.prod(s)) for s in e.values()) == 85_947_648
    assert sum(int(np.prod(s)) for k, s in e.items() if k != "pos_embed") == 85_646_592
    assert sum(int(np.prod(s)) for s in jo.predictor_shapes(jo.VIT_B).values()) == 11_389_440


def test_mask_collator_and_module_mirrors(fx):
    bvc = ge.load_package()
    for mc in fx["mask_collator"]:
        torch.manual_seed(mc["torch_seed"])
        col = bvc.jepa_mask.MaskCollator(input_size=224, patch_size=16, pred_mask_scale=(0.15, 0.2), enc_mask_scale=(0.85, 1.0),
                                         aspect_ratio=(0.75, 1.5), nenc=1, npred=4, allow_overlap=False, min_keep=10)
        batch = [torch.zeros(1) for _ in range(4)]
        for want in mc["steps"]:
            _, me, mp = col(batch)
            assert list(me[0].shape) == want["enc_shape"] and list(mp[0].shape) == want["pred_shape"]
            assert [int(v) for v in me[0][0]] == want["enc_row0"]
            assert [int(v) for v in mp[0][0]] == want["pred0_row0"] and [int(v) for v in mp[3][3]] == want["pred3_row3"]
        upd = bvc.jepa_mask.update_masks([torch.arange(5).view(1, 5)], 224, 16, 2, 1, isencoder=False)
        assert int(upd[0][0, 0]) == mc["update_masks_offset"] == 196
    # module objects: reference state-dict keys / shapes, baked positional tables, deepcopy for the target encoder
    enc = bvc.jepa.vit_base(img_size=[224], num_frames=2, tubelet_size=1)
    sd = enc.state_dict()
    shapes = jo.encoder_shapes(jo.VIT_B)
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == s for k, s in shapes.items())
    assert torch.allclose(sd["pos_embed"], jo.positional_encoding_3d((2, 14, 14), 768), atol=1e-6)
    assert not enc._param("pos_embed").requires_grad
    assert (enc.sequence_shape, enc.embed_dim, enc.num_heads) == ((2, 14, 14), 768, 12)
    pred = bvc.jepa.vit_predictor(sequence_shape=enc.sequence_shape, embed_dim=enc.embed_dim, predictor_embed_dim=384, depth=6,
                                  num_heads=enc.num_heads)
    ps = jo.predictor_shapes(jo.VIT_B)
    assert set(pred.state_dict()) == set(ps) and all(tuple(pred.state_dict()[k].shape) == s for k, s in ps.items())
    tgt = copy.deepcopy(enc)
    assert all(torch.equal(a, b) for a, b in zip(tgt.state_dict().values(), sd.values()))
    tgt._param("norm.weight").data.add_(1.0)
    assert not torch.equal(tgt.state_dict()["norm.weight"], enc.state_dict()["norm.weight"])
    x = torch.arange(24.).view(2, 3, 4)
    m = [torch.tensor([[0, 2], [1, 2]])]
    assert torch.equal(bvc.jepa.apply_masks(x, m), jo.apply_masks(x, m))
    with pytest.raises(bvc._lib.BvcError):
        enc(torch.zeros(1, 2, 3, 224, 224))
