"""grad_logger with the reference's interface (pretraining/generative/loggingtools.py:98-119):
L2 norms of three named gradients -> the CSV columns grad-EFL / grad-ELL / grad-DLL."""
import torch


class _Stats:
    enc_first_layer = 0.0
    enc_last_layer = 0.0
    dec_last_layer = 0.0


GRAD_PROBES = {
    "enc_first_layer": "videomae.embeddings.patch_embeddings.projection.weight",
    "enc_last_layer": "encoder_to_decoder.weight",
    "dec_last_layer": "decoder.head.weight",
}


def grad_logger(named_params):
    stats = _Stats()
    wanted = {v: k for k, v in GRAD_PROBES.items()}
    for n, p in named_params:
        if n in wanted and p.grad is not None:
            setattr(stats, wanted[n], float(torch.norm(p.grad.data)))
    return stats
