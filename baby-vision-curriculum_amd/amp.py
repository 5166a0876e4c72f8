"""``GradScaler`` with the inf check done by one read-only HIP pass per contiguous gradient range.

``torch.amp.GradScaler`` (what the reference builds at pretrain_videomae.py:197 and steps at :312-314) checks the gradients
of an optimiser that consumes the scale itself (``_step_supports_amp_scaling``, as ``bvc.optim.*`` do) with
``_amp_foreach_non_finite_check_and_unscale_(grads, found_inf, inv_scale=1)``: every gradient is read AND written back.
On a flat gradient buffer that is a single read (``bvc_op_nonfinite_check``).  Same constructor, same ``scale / step / update
/ state_dict``; optimisers whose gradients are not contiguous f32 CUDA ranges fall back to the stock check.
"""
import torch

from . import _lib
from .optim import SGD


class GradScaler(torch.amp.GradScaler):
    def _check_inf_per_device(self, optimizer):
        # This overrides a private hook of torch.amp.GradScaler (torch 2.10: called from step() for optimisers that consume the
        # scale themselves).  Anything unexpected - a torch release that renamed the internals, gradients that are not
        # contiguous f32 CUDA ranges - falls back to the stock implementation.
        try:
            _scale, _ = self._check_scale_growth_tracker("_check_inf_per_device")
            states = self._per_optimizer_states
            runs = []
            for group in optimizer.param_groups:
                runs += SGD._contiguous_runs(group["params"])
        except (_lib.BvcError, AttributeError, TypeError):
            return super()._check_inf_per_device(optimizer)
        per_device = {}
        L = _lib.lib()
        for run in runs:
            dev = run[0].device
            if dev not in per_device:
                per_device[dev] = torch.zeros((), dtype=torch.float32, device=dev)
            n = sum(p.numel() for p in run)
            with torch.cuda.device(dev):
                _lib.check(L.bvc_op_nonfinite_check(run[0].grad.data_ptr(), n, per_device[dev].data_ptr(),
                                                    _lib.current_stream_ptr()), "bvc_op_nonfinite_check")
        if not per_device:
            per_device[_scale.device] = torch.zeros((), dtype=torch.float32, device=_scale.device)
        states[id(optimizer)]["found_inf_per_device"] = per_device
        return per_device
