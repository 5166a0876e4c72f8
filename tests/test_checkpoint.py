"""Checkpoint wire format (host logic, CPU): reference dict layout, legacy transformers-4.x key mapping, safe loading."""
import os

import pytest
import torch

import __graft_entry__ as ge


@pytest.fixture(scope="module")
def ckpt():
    try:
        bvc = ge.load_package()
    except Exception as e:      # the library is built by __graft_entry__.build(); without it there is nothing to test
        pytest.skip(f"package not loadable: {e}")
    return bvc.checkpoint


def _legacy_sd(d=8, layers=2):
    g = torch.Generator().manual_seed(0)
    sd = {}
    for i in range(layers):
        a = f"videomae.encoder.layer.{i}.attention.attention."
        for n in ("query", "key", "value"):
            sd[a + n + ".weight"] = torch.randn(d, d, generator=g)
        sd[a + "q_bias"] = torch.randn(d, generator=g)
        sd[a + "v_bias"] = torch.randn(d, generator=g)
        sd[f"videomae.encoder.layer.{i}.output.dense.bias"] = torch.randn(d, generator=g)
    return sd


def test_legacy_videomae_keys_are_mapped(ckpt):
    old = _legacy_sd()
    new = ckpt.convert_legacy_videomae_state_dict(old)
    a = "videomae.encoder.layer.1.attention.attention."
    assert torch.equal(new[a + "query.bias"], old[a + "q_bias"])
    assert torch.equal(new[a + "value.bias"], old[a + "v_bias"])
    assert torch.count_nonzero(new[a + "key.bias"]) == 0 and new[a + "key.bias"].shape == old[a + "q_bias"].shape
    assert not any(k.endswith(("q_bias", "v_bias")) for k in new)
    assert len(new) == len(old) + 2                     # one key.bias added per layer
    again = ckpt.convert_legacy_videomae_state_dict(new)
    assert again.keys() == new.keys() and all(again[k] is new[k] for k in new)   # already 5.x: untouched


def test_save_and_init_round_trip_in_the_reference_layout(ckpt, tmp_path):
    class Meter:
        def __init__(self, v):
            self.avg = v
    model = torch.nn.Linear(4, 3)
    wrapper = torch.nn.Module()
    wrapper.module = model                              # what DDP exposes; save_checkpoint stores .module's keys
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    path = os.path.join(tmp_path, "model_x.pth.tar")
    ckpt.save_checkpoint(path, wrapper, 3, {"train": Meter(0.5), "val": Meter(0.75)}, 16, 8, 0.1, opt)
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"model_state_dict", "opt", "epoch", "train_loss", "val_loss", "batch_size", "world_size", "lr"}
    assert set(raw["model_state_dict"]) == {"weight", "bias"} and raw["epoch"] == 3 and raw["val_loss"] == 0.75
    fresh = torch.nn.Linear(4, 3)
    ckpt.init_model_from_checkpoint(fresh, path)
    assert torch.equal(fresh.weight, model.weight) and torch.equal(fresh.bias, model.bias)


def test_jepa_load_checkpoint_convention(ckpt, tmp_path):
    enc, pred, tgt = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(list(enc.parameters()) + list(pred.parameters()), lr=0.1)
    path = os.path.join(tmp_path, "jepa.pth.tar")
    torch.save({"encoder": enc.state_dict(), "predictor": pred.state_dict(), "target_encoder": tgt.state_dict(),
                "opt": opt.state_dict(), "scaler": None, "epoch": 7}, path)
    e2, p2, t2 = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    out = ckpt.load_checkpoint(path, e2, p2, t2, None, None)
    assert out[-1] == 7 and torch.equal(e2.weight, enc.weight) and torch.equal(t2.bias, tgt.bias)
    out = ckpt.load_checkpoint(os.path.join(tmp_path, "missing.tar"), e2, None, None, None, None)
    assert out[-1] == 0                                  # the reference's fallback: log, start from epoch 0
