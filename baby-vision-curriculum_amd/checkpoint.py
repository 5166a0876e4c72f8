"""Checkpoint wire format of the curriculum stages, kept byte-compatible with the reference so that stage
k+1 can start from stage k whichever side wrote the file.

  pretraining/generative/pretrain_videomae.py:66-85  init_model_from_checkpoint / save_checkpoint
      {'model_state_dict', 'opt', 'epoch', 'train_loss', 'val_loss', 'batch_size', 'world_size', 'lr'}
  pretraining/predictive/helper.py:23-66             load_checkpoint
      {'encoder', 'predictor', 'target_encoder', 'opt', 'scaler', 'epoch', ...}

Files are read with ``torch.load(..., weights_only=True)``: a checkpoint is data, nothing in it is executed.
``convert_legacy_videomae_state_dict`` maps checkpoints written with 2023-era transformers 4.x (attention with
``q_bias`` / ``v_bias`` parameters and a bias-free key projection) onto the 5.x key set this package and the container's
transformers use (``query.bias`` / ``key.bias`` / ``value.bias``); the key bias of such a model is identically zero.
"""
import re

import torch

_LEGACY = re.compile(r"^(.*attention\.attention\.)(q_bias|v_bias)$")


def convert_legacy_videomae_state_dict(state_dict):
    """Returns a new dict with 5.x keys; a dict that already has them is returned unchanged (same tensors)."""
    if not any(_LEGACY.match(k) for k in state_dict):
        return dict(state_dict)
    out = {}
    for k, v in state_dict.items():
        m = _LEGACY.match(k)
        if m is None:
            out[k] = v
            continue
        prefix, which = m.group(1), m.group(2)
        if which == "q_bias":
            out[prefix + "query.bias"] = v
            out[prefix + "key.bias"] = torch.zeros_like(v)      # 4.x: torch.cat((q_bias, zeros_like(v_bias), v_bias))
        else:
            out[prefix + "value.bias"] = v
    return out


def _load(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def init_model_from_checkpoint(model, checkpoint_path, strict=True):
    """pretrain_videomae.py:66-70 (also benchmarks/compute_embeddings_videomae.py:55-59)."""
    checkpoint = _load(checkpoint_path)
    model.load_state_dict(convert_legacy_videomae_state_dict(checkpoint["model_state_dict"]), strict=strict)
    return model


def save_checkpoint(chpt_path, model, epoch, loss_meter, batch_size, world_size, lr, optimizer):
    """pretrain_videomae.py:72-85; `model` may be the DistributedDataParallel wrapper (its .module is saved) or the bare module."""
    inner = getattr(model, "module", model)
    avg = lambda m: float(getattr(m, "avg", m))
    torch.save({
        "model_state_dict": {k: v.detach().cpu().clone() for k, v in inner.state_dict().items()},
        "opt": optimizer.state_dict(),
        "epoch": epoch,
        "train_loss": avg(loss_meter["train"]),
        "val_loss": avg(loss_meter["val"]),
        "batch_size": batch_size,
        "world_size": world_size,
        "lr": lr,
    }, chpt_path)


def load_checkpoint(r_path, encoder, predictor, target_encoder, opt, scaler):
    """helper.py:23-66 with its return convention (epoch 0 and untouched objects when the file cannot be read)."""
    try:
        checkpoint = _load(r_path)
        epoch = checkpoint["epoch"]
        encoder.load_state_dict(checkpoint["encoder"])
        if predictor is not None:
            predictor.load_state_dict(checkpoint["predictor"])
        if target_encoder is not None:
            target_encoder.load_state_dict(checkpoint["target_encoder"])
        if opt is not None:
            opt.load_state_dict(checkpoint["opt"])
            if scaler is not None:
                scaler.load_state_dict(checkpoint["scaler"])
    except Exception as e:      # the reference logs and restarts from epoch 0
        print(f"Encountered exception when loading checkpoint {e}")
        epoch = 0
    return encoder, predictor, target_encoder, opt, scaler, epoch
