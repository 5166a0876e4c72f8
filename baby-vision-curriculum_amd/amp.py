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
            ranges = []       # (gradient address, elements, device)
            if hasattr(optimizer, "_get_plans"):
                # bvc.optim: a flat module's whole gradient buffer is ONE range whatever the parameter groups (gradients of frozen
                # parameters are the buffer's zeros); parameters outside flat buffers go by adjacent runs
                plans, loose = optimizer._get_plans()
                ranges += [(pl.gbase, pl.n, pl.module._flat.device) for pl in plans]
                for gi, ps in loose.items():
                    ranges += [(run[0].grad.data_ptr(), sum(p.numel() for p in run), run[0].device) for run in optimizer._group_runs(gi, ps)]
            else:
                for group in optimizer.param_groups:
                    ranges += [(run[0].grad.data_ptr(), sum(p.numel() for p in run), run[0].device)
                               for run in SGD._contiguous_runs(group["params"])]
        except (_lib.BvcError, AttributeError, TypeError):
            return super()._check_inf_per_device(optimizer)
        per_device = {}
        L = _lib.lib()
        for ptr, n, dev in ranges:
            if dev not in per_device:
                per_device[dev] = torch.zeros((), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.bvc_op_nonfinite_check(ptr, n, per_device[dev].data_ptr(), _lib.current_stream_ptr()), "bvc_op_nonfinite_check")
        if not per_device:
            per_device[_scale.device] = torch.zeros((), dtype=torch.float32, device=_scale.device)
        states[id(optimizer)]["found_inf_per_device"] = per_device
        return per_device
